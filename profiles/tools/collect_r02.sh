#!/bin/bash
# Collects the round-2 profile evidence on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command (M1 default, M2, M3)       -> gpurun_out/prof_r02_*/
#   2. separate --pmc passes of the M1 command: FETCH_SIZE, WRITE_SIZE, and the SQ counters, each with
#      --kernel-trace only (MI355X_MICROARCH.md: never mix --pmc with other trace domains), plus the
#      calibration kernels for the 8 B/lane access width                                 -> gpurun_out/pmc_r02_*/
# profiles/tools/summarize_r02.py then writes the files kept under profiles/.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build"
run() { echo "== $1"; shift; "$@" || { echo "FAILED: $*"; exit 1; }; }
run m1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r02_m1 -- $BENCH > $OUT/prof_r02_m1.log 2>&1
run m2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r02_m2 -- $BENCH --workload m2 > $OUT/prof_r02_m2.log 2>&1
run m3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r02_m3 -- $BENCH --workload m3 --grid 256 > $OUT/prof_r02_m3.log 2>&1
run fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_r02_fetch -- $BENCH > $OUT/pmc_r02_fetch.log 2>&1
run write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_r02_write -- $BENCH > $OUT/pmc_r02_write.log 2>&1
run sq rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_r02_sq -- $BENCH > $OUT/pmc_r02_sq.log 2>&1
if [ ! -x $ROOT/profiles/tools/calib_traffic ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $ROOT/profiles/tools/calib_traffic $ROOT/profiles/tools/calib_traffic.hip
fi
run calib_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_r02_calib_fetch -- $ROOT/profiles/tools/calib_traffic > $OUT/pmc_r02_calib_fetch.log 2>&1
run calib_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_r02_calib_write -- $ROOT/profiles/tools/calib_traffic > $OUT/pmc_r02_calib_write.log 2>&1
echo "collection done"
python3 $ROOT/profiles/tools/summarize_r02.py
