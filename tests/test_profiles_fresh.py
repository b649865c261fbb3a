"""Staleness guard of the profile-derived numbers bench.py quotes: profiles/compute_latest.json holds, per kernel of the
10 s-per-call configs, the SQ_INSTS_VALU count of a committed rocprofv3 pass AND the static vector-instruction count of that kernel in the
library the pass was made with.  A kernel change moves the static count: this test then fails until the profile is regenerated
(scripts/profile_bench.sh PROFILE_SQ=1 + scripts/make_compute_latest.py).  CPU-only: it disassembles the built libdsenh.so."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compute_profile_names_kernels_of_this_build_with_their_instruction_counts():
    prof = json.load(open(os.path.join(ROOT, "profiles", "compute_latest.json")))
    mix = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "kernel_mix.py")]))["kernels"]
    norm = {k.replace(" ", ""): v for k, v in mix.items()}
    checked = 0
    for cfg, ent in prof.items():
        if not isinstance(ent, dict) or "kernels" not in ent:
            continue
        for name, rec in ent["kernels"].items():
            key = name.replace(" ", "")
            hit = [v for k, v in norm.items() if k == key or k.startswith(key[:95])]
            assert hit, "%s: kernel %s of the committed profile is not in the built library" % (cfg, name)
            assert rec.get("static_valu"), "%s: %s has no static instruction count: regenerate profiles/compute_latest.json" % (cfg, name)
            assert abs(hit[0]["valu"] - rec["static_valu"]) <= 0.02 * rec["static_valu"], \
                "%s: %s has %d vector instructions in this build, %d when it was profiled: the committed SQ counters are stale" % (
                    cfg, name, hit[0]["valu"], rec["static_valu"])
            checked += 1
    assert checked >= 4


def test_traffic_profile_names_kernels_of_this_build():
    prof = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    mix = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "kernel_mix.py")]))["kernels"]
    names = [k.replace(" ", "") for k in mix]
    for cfg, ent in prof.items():
        for kname in ent.get("kernels", {}):
            key = kname.replace(" ", "")
            assert any(key in n for n in names), "%s: kernel %s of the committed PMC passes is not in the built library" % (cfg, kname)
