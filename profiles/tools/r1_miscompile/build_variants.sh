#!/bin/bash
# Rebuilds the round-1 kernels (git history: commit b38e8f6, the 500-register one-wavefront-per-SIMD stage
# kernels k_stage_rows / k_stage_march that round 2 removed) with the strict MHD object compiled in several ways,
# to reproduce and bisect the "register junk" output of strict k_stage_rows<MHD,1,HLL>:
#   r1_good.so   the round-1 Makefile as it was (-O1 -DPION_WAVE_UNIFORM for the strict objects)       passes
#   r1_bad.so    strict MHD object at -O2, vector wavefront index                                       FAILS
#   r1_v1..v8    -O2 + one -mllvm switch each:
#        v1 -enable-misched=0  FAILS      v2 -enable-post-misched=0  FAILS
#        v3 -disable-machine-licm  passes  v4 -disable-machine-sink  passes
#        v5 -vgpr-regalloc=basic  passes   v6 -sgpr-regalloc=basic  passes
#        v7 -amdgpu-spill-sgpr-to-vgpr=0  FAILS (other variables)   v8 -amdgpu-enable-rewrite-partial-reg-uses=0  FAILS
# (results of 2026-10 on MI355X, ROCm 7.2 hipcc; run through gpurun:
#    for v in good bad v1 ... v8; do PION_GPU_LIB=$PWD/variants/r1_$v.so python profiles/tools/r1_miscompile/repro.py; done)
# Everything is written under variants/ (git-ignored objects; remove it afterwards).
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
mkdir -p $ROOT/variants/r1 && cd $ROOT
git archive b38e8f6 pion_amd/csrc include | tar -x -C variants/r1
cd variants/r1/pion_amd/csrc
make -j8 > /dev/null 2>&1
cp libpion_gpu.so $ROOT/variants/r1_good.so
C="--offload-arch=gfx950 -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-variable -Wno-unused-but-set-variable -Wno-pass-failed -Wno-unused-value -ffp-contract=off -DPION_FPNS=fp_strict -DPION_EQSEL=2 -O2"
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/variants/r1_$1.so build/kernels_strict_0.o build/kernels_strict_1.o build/s2_$1.o build/kernels_strict_3.o build/kernels_fast_0.o build/kernels_fast_1.o build/kernels_fast_2.o build/kernels_fast_3.o build/pion_gpu.o; }
/opt/rocm/bin/hipcc $C -c kernels_fp.hip -o build/s2_bad.o && link bad
i=0
for f in -enable-misched=0 -enable-post-misched=0 -disable-machine-licm -disable-machine-sink -vgpr-regalloc=basic -sgpr-regalloc=basic -amdgpu-spill-sgpr-to-vgpr=0 -amdgpu-enable-rewrite-partial-reg-uses=0; do
  i=$((i+1)); ( /opt/rocm/bin/hipcc $C -mllvm $f -c kernels_fp.hip -o build/s2_v$i.o && link v$i ) &
done
wait
ls -la $ROOT/variants/*.so
