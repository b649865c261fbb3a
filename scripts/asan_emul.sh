#!/bin/bash
# AddressSanitizer + UBSan over the kernel block programs: the CPU-emulated build of ds_core.hpp / ds_ops.hpp / ds_fdaf.hpp /
# ds_wpe.hpp / ds_tdfilter.hpp (tests/emul) is compiled with -fsanitize=address,undefined and the emulator tests run against it.
# (GPU-side sanitizers are not available on the pool; this checks the same index arithmetic on the CPU.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -mfma -fsanitize=address,undefined -fno-omit-frame-pointer -fno-strict-aliasing \
    tests/emul/ds_emul.cpp -o /tmp/libds_emul_asan.so
cp tests/emul/libds_emul.so /tmp/libds_emul_plain.so 2>/dev/null || true
cp /tmp/libds_emul_asan.so tests/emul/libds_emul.so; touch tests/emul/libds_emul.so
trap 'if [ -f /tmp/libds_emul_plain.so ]; then cp /tmp/libds_emul_plain.so tests/emul/libds_emul.so; touch tests/emul/libds_emul.so; fi' EXIT
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
    python -m pytest tests/test_kernel_emul.py -x -q
