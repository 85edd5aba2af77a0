#!/bin/bash
# Builds a variant of libpion_gpu.so for A/B runs on one box (profiles/tools/ab_bench.sh label=ab/NAME/libpion_gpu.so):
#   profiles/tools/build_variant.sh NAME [extra hipcc flags, e.g. -DPION_DPP_SHIFT=0]
# The sources are the working tree's (or, with SRC=<dir>, another copy); objects and the library go to ab/NAME/
# (git-ignored, travels with gpurun).
set -e
NAME=$1; shift
HERE=$(cd "$(dirname "$0")/../.." && pwd)
SRC=${SRC:-$HERE/pion_amd/csrc}
OUT=$HERE/ab/$NAME
mkdir -p $OUT/src
cp $SRC/*.h $SRC/*.hip $SRC/Makefile $OUT/src/
mkdir -p $OUT/include && cp $HERE/include/pion_gpu.h $OUT/include/
# the Makefile refers to ../../include/pion_gpu.h
mkdir -p $OUT/src/../../include 2>/dev/null || true
cp $HERE/include/pion_gpu.h $OUT/../include/ 2>/dev/null || { mkdir -p $OUT/../include; cp $HERE/include/pion_gpu.h $OUT/../include/; }
make -j8 -C $OUT/src COMMON="--offload-arch=gfx950 -fno-slp-vectorize -fPIC -std=c++17 -w $*" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cp $OUT/src/libpion_gpu.so $OUT/libpion_gpu.so
rm -rf $OUT/src/build
echo "built $OUT/libpion_gpu.so"
