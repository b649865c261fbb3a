#!/bin/bash
# Device-only assembly of a FEW operator kernels of ds_kernels_ops.hip (seconds instead of the whole unit), with their register counts and
# static instruction mix:   bash scripts/asm_ops.sh "X(OP_MCSPP,6) X(OP_MCSPP,4)" [extra hipcc flags]   ->  /tmp/asm_ops.s
set -e
LIST=${1:-"X(OP_MCSPP,6)"}; shift || true
cd "$(dirname "$0")/../distantspeech_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -DDS_ARCH=gfx950 "-DDS_FOR_EACH_OP(X)=$LIST" "$@" -S --cuda-device-only ds_kernels_ops.hip -o /tmp/asm_ops.s
python3 - <<'PY'
import re
txt = open('/tmp/asm_ops.s').read()
for m in re.finditer(r'^(_ZN2ds15ds_binop_kernelILi(\d+)ELi(\d+)EEEvNS_8OpParamsE):[^\n]*\n(.*?)\.Lfunc_end\d+:', txt, re.S | re.M):
    name, op, M, body = m.group(1), m.group(2), m.group(3), m.group(4)
    ins = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    v = sum(1 for i in ins if i.startswith('v_')); f64 = sum(1 for i in ins if i.startswith('v_') and 'f64' in i)
    sc = sum(1 for i in ins if i.startswith(('scratch_', 'buffer_')) and False)
    meta = re.search(r'\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel' % re.escape(name), txt, re.S).group(1)
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', meta).group(1); ag = re.search(r'\.amdhsa_accum_offset (\d+)', meta)
    pss = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', meta).group(1)
    nv = re.search(r'\.set %s\.num_vgpr, (\d+)' % re.escape(name), txt); na = re.search(r'\.set %s\.num_agpr, (\d+)' % re.escape(name), txt)
    print("binop<%s,%s>: %d instr, %d vector (%d f64), vgpr %s + agpr %s (next_free %s) scratch %s B" % (op, M, len(ins), v, f64, nv and nv.group(1), na and na.group(1), vg, pss))
PY
