#!/bin/bash
# Collects the round-1 profile evidence on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command          -> gpurun_out/prof_r01/
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command and of the calibration
#      kernels (MI355X_MICROARCH.md, HBM section: calibrate 8 B/lane accesses)  -> gpurun_out/pmc_*/
# profiles/tools/summarize_r01.py then writes the files kept under profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r01 -- $BENCH > $OUT/prof_r01.log 2>&1
echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
echo "pmc write done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_calib_fetch -- $ROOT/profiles/tools/calib_traffic > $OUT/pmc_calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_calib_write -- $ROOT/profiles/tools/calib_traffic > $OUT/pmc_calib_write.log 2>&1
echo "calibration done"
