#!/usr/bin/env python3
"""profiles/compute_latest.json: the vector-issue and LDS figures bench.py uses for the compute roofline of the 10 s-per-call regime.

    python scripts/make_compute_latest.py <name>=<prof_dir>:<frames per bench step> ... > profiles/compute_latest.json
    e.g. cfg2_10s_chunks=gpurun_out/prof_r03d_cfg2_T625:640000

Per config, from the SQ counter passes of scripts/profile_bench.sh (compute.json, mean per launch and kernel) and the static instruction mix of
the shipped kernels (scripts/kernel_mix.py on libdsenh.so: cycles_per_valu = issue cost of one vector instruction at that kernel's mix,
priced with scratch/micro/valu_rate.hip's measurements):
  valu_issue_cycles_per_frame = sum over the step's kernels of SQ_INSTS_VALU * cycles_per_valu * launches per step / frames per step / 1024 SIMDs
  lds_cycles_per_frame        = sum of SQ_LDS_IDX_ACTIVE * launches per step / frames per step / 256 CUs
bench.py: frac_valu = frames/s per GPU * valu_issue_cycles_per_frame / 2.4e9 (the share of every SIMD's cycles that issuing vector
instructions takes at the measured frame rate; 1.0 = the vector pipes are the limit)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mix = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "kernel_mix.py")]))["kernels"]
out = {"source": "scripts/make_compute_latest.py (SQ counters: scripts/profile_bench.sh PROFILE_SQ=1; instruction mix: scripts/kernel_mix.py; issue costs: "
                 "scratch/micro/valu_rate.hip, profiles/r03a/valu_rate.txt)", "simds": 1024, "cus": 256, "clock_ghz_peak": 2.4}
for spec in sys.argv[1:]:
    name, rest = spec.split("=", 1)
    pdir, frames = rest.rsplit(":", 1)
    # pdir: a profile directory of scripts/profile_bench.sh (compute.json, pmc_sq.log), or the prefix of its condensed copies
    # (<prefix>_compute.json, <prefix>_pmc_sq_line.json: what scripts/profile_all.sh keeps and profiles/<round>/ holds)
    condensed = not os.path.isdir(pdir)
    comp = json.load(open(pdir + "_compute.json" if condensed else os.path.join(pdir, "compute.json")))
    frames = float(frames)
    ks = comp["kernels"]
    # bench steps of the profiled run: warm-up + the untimed round + the probe round + R timed rounds of K steps (bench.py measure()); the
    # run's own JSON line (pmc_sq.log) says K, W and R.  A step may be several launches of a kernel (utterance groups, 62-block pieces)
    line = [x for x in open(pdir + "_pmc_sq_line.json" if condensed else os.path.join(pdir, "pmc_sq.log")).read().splitlines() if x.startswith("{")][-1]
    bl = json.loads(line)
    steps = bl["warmup"] + bl["steps"] * (2 + bl["rounds"])
    # (the tracked name of the source: a round's gpurun_out/<tag>/ is committed as profiles/<tag>/)
    ent = {"profile": os.path.relpath(pdir, ROOT).replace("gpurun_out/", "profiles/", 1), "frames_per_step": frames, "steps_profiled": steps, "kernels": {}}
    valu = lds = insts = 0.0
    for k, v in ks.items():
        if "SQ_INSTS_VALU" not in v:
            continue
        m = None
        for mk, mv in mix.items():
            if mk.replace(" ", "") == k.replace(" ", "") or mk.replace(" ", "").startswith(k.replace(" ", "")[:95]):
                m = mv
                break
        cpv = m["cycles_per_valu"] if m else 3.6
        per_step = v["launches"] / steps
        kv = v["SQ_INSTS_VALU"] * cpv * per_step
        kl = v.get("SQ_LDS_IDX_ACTIVE", 0.0) * per_step
        valu += kv; lds += kl; insts += v["SQ_INSTS_VALU"] * per_step
        ent["kernels"][k] = {"launches_per_step": per_step, "SQ_INSTS_VALU": v["SQ_INSTS_VALU"], "cycles_per_valu": cpv,
                             "static_valu": m["valu"] if m else None,     # vector instructions of the kernel in the library the profile was made with (tests/test_profiles_fresh.py)
                             "SQ_LDS_IDX_ACTIVE": v.get("SQ_LDS_IDX_ACTIVE"), "GRBM_GUI_ACTIVE_per_xcd": v.get("GRBM_GUI_ACTIVE", 0.0) / 8,
                             "SQ_WAIT_ANY_share_of_wave_cycles": (v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else None}
    ent["valu_instructions_per_frame"] = insts / frames
    ent["valu_issue_cycles_per_frame"] = valu / frames / out["simds"]
    ent["lds_cycles_per_frame"] = lds / frames / out["cus"]
    out[name] = ent
json.dump(out, sys.stdout, indent=1)
