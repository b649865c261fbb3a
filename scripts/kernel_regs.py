#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in libdsenh.so (from the code objects' metadata notes).

    python scripts/kernel_regs.py [distantspeech_amd/libdsenh.so] > regs.tsv        # name <tab> vgprs <tab> sgprs <tab> scratch B <tab> LDS B
    python scripts/kernel_regs.py a.so --diff b.so                                  # kernels whose numbers differ between two builds"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def table(so):
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
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" and cur.get("name"):
                    cur = {}
                cur[k] = v
                if k == "vgpr_count" and "name" in cur:
                    name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                    out[name] = (int(cur["vgpr_count"]), int(cur.get("sgpr_count", 0)), int(cur.get("private_segment_fixed_size", 0)),
                                 int(cur.get("group_segment_fixed_size", 0)))
                    cur = {}
    return out


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in sys.argv[1:] if a != "--diff"]
    a = table(args[0] if args else os.path.join(root, "distantspeech_amd", "libdsenh.so"))
    if "--diff" in sys.argv:
        b = table(args[1])
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                print("%s\t%s\t->\t%s" % (k, a.get(k), b.get(k)))
        return
    for k, v in sorted(a.items()):
        print("%s\t%d\t%d\t%d\t%d" % ((k,) + v))


if __name__ == "__main__":
    main()
