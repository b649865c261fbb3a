#!/bin/bash
# build_variant.sh <git-rev|WORK> <name>: build libdsenh.so from a given revision into scratch/variants/ for A/B timing
set -e
REV=$1; NAME=$2; ROOT=$(cd "$(dirname "$0")/.." && pwd); TMP=/tmp/proto/var_$NAME
rm -rf $TMP; mkdir -p $TMP/distantspeech_amd/csrc $TMP/include
if [ "$REV" = "WORK" ]; then cp $ROOT/distantspeech_amd/csrc/*.h* $ROOT/distantspeech_amd/csrc/Makefile $TMP/distantspeech_amd/csrc/; cp $ROOT/include/dsenh.h $TMP/include/;
else (cd $ROOT && git archive $REV distantspeech_amd/csrc include | tar -x -C $TMP); fi
make -C $TMP/distantspeech_amd/csrc -j8 EXTRA="$3" > $TMP/build.log 2>&1 || { tail -20 $TMP/build.log; exit 1; }
cp $TMP/distantspeech_amd/libdsenh.so $ROOT/scratch/variants/libdsenh_$NAME.so
echo built scratch/variants/libdsenh_$NAME.so
