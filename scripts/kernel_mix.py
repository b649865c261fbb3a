#!/usr/bin/env python3
"""Static vector-instruction mix of every kernel in libdsenh.so and its issue cost per instruction on MI355X.

    python scripts/kernel_mix.py [distantspeech_amd/libdsenh.so] > profiles/<round>/kernel_mix.json

The fat binary's gfx950 code objects are unbundled (clang-offload-bundler) and disassembled (llvm-objdump); per kernel the vector
instructions are counted by issue class and priced with the costs measured by scratch/micro/valu_rate.hip at 4 waves per SIMD
(profiles/r03a/valu_rate.txt): packed f32 (v_pk_*) 4.33 cycles per wave-instruction per SIMD, three-source / carry / select forms (v_fma,
v_fmac, v_cndmask, v_mad, v_lshl_add, ...) 4.0, transcendentals 8.24, everything else 2.5.  `cycles_per_valu` = the mix-weighted mean: what
one vector instruction of that kernel costs the SIMD when issue is the only limit.  bench.py multiplies it with the PMC instruction count
per frame (SQ_INSTS_VALU, profiles/compute_latest.json) for the vector-issue roofline of the 10 s-per-call regime."""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
COST = {"pk": 4.33, "src3": 4.0, "trans": 8.24, "other": 2.5}
SRC3 = re.compile(r"v_(fma|fmac|fmamk|fmaak|mad|cndmask|lshl_add|add3|and_or|lshl_or|or3|xad|bfe|bfi|alignbit|perm|med3|max3|min3|div_fixup|div_fmas|add_lshl|"
                  r"addc|subb|add_co|sub_co|mad_u64|mul_lo|mul_hi|dot)")


def classify(op):
    if op.startswith("v_pk_"):
        return "pk"
    if re.match(r"v_(rcp|rsq|exp|log|sqrt|sin|cos)_", op):
        return "trans"
    if op.startswith(("v_mfma", "v_smfma")):
        return "mfma"
    if SRC3.match(op):
        return "src3"
    return "other"


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "distantspeech_amd", "libdsenh.so")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for n, a in enumerate(starts):
            b = starts[n + 1] if n + 1 < len(starts) else len(blob)
            piece, co = os.path.join(td, "b%d.bin" % n), os.path.join(td, "b%d.elf" % n)
            open(piece, "wb").write(blob[a:b])
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + piece, "--output=" + co], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) < 4096:
                continue
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    name = m.group(1)
                    cur = out.setdefault(name, {"pk": 0, "src3": 0, "trans": 0, "other": 0, "mfma": 0, "lds": 0, "salu": 0, "vmem": 0}) if "ds::" in name else None
                    continue
                if cur is None:
                    continue
                parts = line.split()
                if not parts:
                    continue
                op = parts[0]
                if op.startswith("v_"):
                    cur[classify(op)] += 1
                elif op.startswith("ds_"):
                    cur["lds"] += 1
                elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                    cur["vmem"] += 1
                elif op.startswith("s_"):
                    cur["salu"] += 1
    res = {}
    for k, c in sorted(out.items()):
        n = c["pk"] + c["src3"] + c["trans"] + c["other"]
        if n == 0:
            continue
        cyc = sum(COST[t] * c[t] for t in COST)
        res[k] = dict(c, valu=n, cycles_per_valu=round(cyc / n, 3))
    json.dump({"costs_cycles_per_wave_instruction_per_simd": COST, "source": "scratch/micro/valu_rate.hip at 4 waves per SIMD (profiles/r03a/valu_rate.txt)",
               "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
