#!/bin/bash
# scratch/libdsenh_abl.so: the library with the chains' timing-experiment switches compiled in (-DDS_ABLATE_CHAIN: DS_ABL_SKIP, DS_ABL_AFTER,
# DS_ABL_PIECE in ds_api_chains.hip; DS_ABL_FIR_OPL4 in ds_kernels_ops.hip; DS_ABL_CU_<stream bit> in ds_api.hip) — only those units are
# rebuilt, the rest are the shipped library's objects (run `make -C distantspeech_amd/csrc` first).  Used by scratch/jobs_r06/jobs_r06_abl*.sh
# through DSENH_LIB.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/distantspeech_amd/csrc; T=/tmp/abl; mkdir -p $T
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DDS_ARCH=gfx950 -DDS_ABLATE_CHAIN"
U="ds_api ds_api_chains ds_kernels_ops"
for u in $U; do (cd $C && /opt/rocm/bin/hipcc $F -c $u.hip -o $T/$u.o); done
OBJS=$(cd $C && ls *.o | grep -v -x "ds_api\.o\|ds_api_chains\.o\|ds_kernels_ops\.o\|ds_kernels_adaptive_q\.o" | sed "s|^|$C/|")
/opt/rocm/bin/hipcc -shared -fPIC -Wl,-z,defs --offload-arch=gfx950 $OBJS $T/ds_api.o $T/ds_api_chains.o $T/ds_kernels_ops.o -o $ROOT/scratch/libdsenh_abl.so
echo built scratch/libdsenh_abl.so
