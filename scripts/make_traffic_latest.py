#!/usr/bin/env python3
"""Assemble profiles/traffic_latest.json (what bench.py quotes as `roofline.traffic`) from the traffic.json files of scripts/profile_bench.sh.

    python scripts/make_traffic_latest.py <round dir, e.g. profiles/r03e> > profiles/traffic_latest.json

Expects <dir>/<key>_traffic.json for key in cfg2, cfg2_hbm, cfg3, cfg4, cfg5, wpe_nb, cfg4_n10 (the bench's own keys).  The figure per key is the HBM bytes of
ONE bench step: the sum over the kernels of mean bytes per launch x launches per step (a step of cfg3 / cfg2_hbm / cfg4 launches every kernel once
per utterance group; the chains launch several kernels per step) = traffic.json's `hbm_bytes_per_step`."""
import json
import os
import sys

d = sys.argv[1]
# utterance groups per step: the summariser's step count is the launch count of the once-per-"step" kernels, which for a workload that runs
# as G groups is G x the bench's steps (every kernel is launched once per group) — its per-"step" bytes are then one group's
GROUPS = {"cfg2": 1, "cfg2_hbm": 2, "cfg3": 2, "cfg4": 2, "cfg5": 1, "wpe_nb": 1, "cfg4_n10": 1}
out = {}
for key in ("cfg2", "cfg2_hbm", "cfg3", "cfg4", "cfg5", "wpe_nb", "cfg4_n10"):
    f = os.path.join(d, key + "_traffic.json")
    if not os.path.exists(f):
        continue
    t = json.load(open(f))
    cmd = os.path.join(d, key + "_summary.txt")
    line = ""
    if os.path.exists(cmd):
        first = open(cmd).readline().strip()
        line = first.split("): ", 1)[1] if "): " in first else first
    out[key] = {"hbm_bytes_per_launch": t["hbm_bytes_per_step"] * GROUPS[key], "utterance_groups": GROUPS[key],
                "kernels": {k.replace("void ds::", "").split("(")[0]: round(v["hbm_bytes_per_launch"]) for k, v in t["kernels"].items() if v["hbm_bytes_per_launch"] > 0},
                "source": "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, scripts/profile_bench.sh) of `%s`; HBM bytes = FETCH_SIZE*1024*2 "
                          "(gfx950 wide-read correction) + WRITE_SIZE*1024, mean per launch, summed over the launches of one bench step" % (f, line)}
json.dump(out, sys.stdout, indent=1)
