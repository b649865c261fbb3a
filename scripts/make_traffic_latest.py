#!/usr/bin/env python3
"""Assemble profiles/traffic_latest.json (what bench.py quotes as `roofline.traffic`) from the traffic.json files of scripts/profile_bench.sh.

    python scripts/make_traffic_latest.py <round dir, e.g. profiles/r03e> > profiles/traffic_latest.json

Expects <dir>/<key>_traffic.json for the bench's own keys (cfg2, cfg2_hbm, mvdr_pf, cfg3, cfg4, cfg5, wpe_nb, cfg4_n10, nb_mvdr, nb_mvdr_m4) and,
for the 10 s-per-call entries, <dir>/<cfg>_T<hops>_traffic.json + <cfg>_T<hops>_pmc_fetch_line.json (scripts/profile_all.sh writes both).  The figure per key is the HBM bytes of
ONE bench step: the sum over the kernels of mean bytes per launch x launches per step (a step of cfg3 / cfg2_hbm / cfg4 launches every kernel once
per utterance group; the chains launch several kernels per step) = traffic.json's `hbm_bytes_per_step`."""
import json
import os
import sys

d = sys.argv[1]
# utterance groups per step: the summariser's step count is the launch count of the once-per-"step" kernels, which for a workload that runs
# as G groups is G x the bench's steps (every kernel is launched once per group) — its per-"step" bytes are then one group's
GROUPS = {"cfg2": 1, "cfg2_hbm": 2, "cfg3": 2, "cfg4": 2, "cfg5": 1, "wpe_nb": 1, "cfg4_n10": 1, "mvdr_pf": 1, "nb_mvdr": 1, "nb_mvdr_m4": 1, "tdgsc": 1, "fdgsc": 1}
CHUNKED = {"cfg2": 625, "mvdr_pf": 625, "cfg3": 625, "cfg4": 312, "cfg5": 625, "wpe_nb": 2500, "nb_mvdr": 625, "nb_mvdr_m4": 625}
SRC = ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, scripts/profile_bench.sh) of `%s`; HBM bytes = FETCH_SIZE*1024*2 "
       "(gfx950 wide-read correction) + WRITE_SIZE*1024, mean per launch, summed over the launches of one bench step")


def tracked(path):
    """the name a source is committed under: scripts/profile_all.sh writes a round to gpurun_out/<tag>/ (scratch on the GPU box), the same files are
    then copied to profiles/<tag>/ — the tables name the tracked copy (they named the scratch one)"""
    return path.replace("gpurun_out/", "profiles/", 1)


def command_of(key):
    cmd = os.path.join(d, key + "_summary.txt")
    if not os.path.exists(cmd):
        return ""
    first = open(cmd).readline().strip()
    return first.split("): ", 1)[1] if "): " in first else first


out = {}
for key in GROUPS:
    f = os.path.join(d, key + "_traffic.json")
    if not os.path.exists(f):
        continue
    t = json.load(open(f))
    if "hbm_bytes_per_step" not in t:
        continue
    out[key] = {"hbm_bytes_per_launch": t["hbm_bytes_per_step"] * GROUPS[key], "utterance_groups": GROUPS[key],
                "kernels": {k.replace("void ds::", "").split("(")[0]: round(v["hbm_bytes_per_launch"]) for k, v in t["kernels"].items() if v["hbm_bytes_per_launch"] > 0},
                "source": SRC % (tracked(f), command_of(key))}
# 10 s per call: a bench step is one call of T hops; the chains run it as pieces and utterance groups, so the step's bytes are the sum over
# ALL launches of the run divided by the bench steps the run made (warm-up + the untimed round + the probe round + R timed rounds of K steps:
# the profiled run's own JSON line says K, W and R)
for cfg, T in CHUNKED.items():
    key = "%s_T%d" % (cfg, T)
    f, lf = os.path.join(d, key + "_traffic.json"), os.path.join(d, key + "_pmc_fetch_line.json")
    if not (os.path.exists(f) and os.path.exists(lf)):
        continue
    t = json.load(open(f))
    bl = json.loads(open(lf).read().strip().splitlines()[-1])
    steps = bl["warmup"] + bl["steps"] * (2 + bl["rounds"])
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in t.get("kernels", {}).values())
    if not total:
        continue
    out[cfg + "_10s_chunks"] = {"hbm_bytes_per_launch": total / steps, "bench_steps_profiled": steps, "hops_per_call": T,
                                "kernels": {k.replace("void ds::", "").split("(")[0]: round(v["hbm_bytes_per_launch"] * v["launches"] / steps)
                                            for k, v in t["kernels"].items() if v["hbm_bytes_per_launch"] > 0},
                                "source": SRC % (tracked(f), command_of(key))}
json.dump(out, sys.stdout, indent=1)
