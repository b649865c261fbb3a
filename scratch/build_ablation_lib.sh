#!/bin/bash
# scratch/libdsenh_abl.so: the library with the chains' timing-experiment switches compiled in (-DDS_ABLATE_CHAIN: DS_ABL_SKIP, DS_ABL_AFTER,
# DS_ABL_PIECE in ds_api_chains.hip; DS_ABL_FIR_OPL4 in ds_kernels_ops.hip) — only those two units are rebuilt, the rest are the shipped
# library's objects (run `make -C distantspeech_amd/csrc` first).  Used by scratch/jobs_r06/jobs_r06_abl*.sh through DSENH_LIB.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/distantspeech_amd/csrc; T=/tmp/abl; mkdir -p $T
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DDS_ARCH=gfx950 -DDS_ABLATE_CHAIN"
(cd $C && /opt/rocm/bin/hipcc $F -c ds_api_chains.hip -o $T/ds_api_chains.o && /opt/rocm/bin/hipcc $F -c ds_kernels_ops.hip -o $T/ds_kernels_ops.o)
OBJS=$(cd $C && ls *.o | grep -v "ds_api_chains.o\|ds_kernels_ops.o\|ds_kernels_adaptive_q.o" | sed "s|^|$C/|")
/opt/rocm/bin/hipcc -shared -fPIC -Wl,-z,defs --offload-arch=gfx950 $OBJS $T/ds_api_chains.o $T/ds_kernels_ops.o -o $ROOT/scratch/libdsenh_abl.so
echo built scratch/libdsenh_abl.so
