#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S --cuda-device-only listing, block by block.

    hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -S --cuda-device-only -Iinclude distantspeech_amd/csrc/ds_kernels_adaptive.hip -o /tmp/a.s
    python scripts/isa_stats.py /tmp/a.s ILi512ELi4ELi1ELb0 [--blocks]

Classes: valu (one-pass vector ALU), pk (v_pk_*_f32: two passes), trans (v_rcp / v_rsq / v_exp / v_log / v_sqrt: quarter rate), lds, vmem, salu,
wait (s_waitcnt), barrier.  `issue` = 2 * valu + 4 * pk + 8 * trans cycles per wave (MI355X_MICROARCH.md, cycle constants: a wave64 vector
instruction passes a SIMD-32 in 2 cycles; scratch/micro/valu_rate.hip measures the packed and transcendental forms)."""
import re
import sys


def classify(op):
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "pk"
    if re.match(r"v_(rcp|rsq|exp|log|sqrt|sin|cos)_", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l):
            start = i
            break
    if start is None:
        raise SystemExit("kernel matching %r not found" % pat)
    blocks, cur = [], ["entry", {}]
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            blocks.append(cur)
            cur = [m.group(1), {}]
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[1][c] = cur[1].get(c, 0) + 1
        if c in ("salu",) and op.startswith(("s_cbranch", "s_branch")):
            cur[1].setdefault("br", []).append(s.split()[-1])
    blocks.append(cur)
    keys = ["valu", "pk", "trans", "lds", "vmem", "salu", "wait", "barrier"]
    tot = {k: 0 for k in keys}
    if show_blocks:
        print("%-14s " % "block" + " ".join("%6s" % k for k in keys) + "  issue  branches")
    for name, d in blocks:
        for k in keys:
            tot[k] += d.get(k, 0)
        if show_blocks and sum(d.get(k, 0) for k in keys):
            issue = 2 * d.get("valu", 0) + 4 * d.get("pk", 0) + 8 * d.get("trans", 0)
            print("%-14s " % name + " ".join("%6d" % d.get(k, 0) for k in keys) + " %6d  %s" % (issue, ",".join(d.get("br", []))))
    issue = 2 * tot["valu"] + 4 * tot["pk"] + 8 * tot["trans"]
    print("total          " + " ".join("%6d" % tot[k] for k in keys) + " %6d" % issue)


if __name__ == "__main__":
    main()
