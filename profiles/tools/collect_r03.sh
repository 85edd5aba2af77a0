#!/bin/bash
# Collects the round-3 profile evidence on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command: M1 (default), M1 ideal MHD, M2, M3, the two 2-D
#      workloads and the two axisymmetric ones                                                                      -> gpurun_out/prof_r03_*/
#   2. separate --pmc passes (each with --kernel-trace only; MI355X_MICROARCH.md: never mix --pmc with other trace
#      domains; TCC block: 4 slots, FETCH_SIZE 3, WRITE_SIZE 2): FETCH_SIZE, WRITE_SIZE, the raw fabric-side read
#      counters (requests, 32-B requests, DRAM-routed requests), L2 hit / miss, the SQ set, GRBM_GUI_ACTIVE
#      (effective clock) for M1; FETCH_SIZE / WRITE_SIZE / SQ for M2, M3 and the 2-D workloads; the calibration
#      kernels at the same 8 B/lane access width                                      -> gpurun_out/pmc_r03_*/
# profiles/tools/summarize_r03.py then writes the files kept under profiles/.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build"
prof() { local tag=$1; shift; echo "== stats $tag"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r03_$tag -- $B "$@" > $OUT/prof_r03_$tag.log 2>&1 || { echo "FAILED stats $tag"; tail -3 $OUT/prof_r03_$tag.log; }; }
pmc() { local tag=$1; local ctrs=$2; shift; shift; echo "== pmc $tag"; rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_r03_$tag -- $B "$@" > $OUT/pmc_r03_$tag.log 2>&1 || { echo "FAILED pmc $tag"; tail -3 $OUT/pmc_r03_$tag.log; }; }
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD"
prof m1
prof mhd8 --eqn mhd
prof m2 --workload m2
prof m3 --workload m3 --grid 256
prof dmr2d --workload dmr2d --grid 4096
prof mhd2d --workload mhd2d --grid 4096
prof axi2d --workload axi2d --grid 4096
prof mhdaxi2d --workload mhdaxi2d --grid 4096
for wl in "m1" "m2 --workload m2" "m3 --workload m3 --grid 256" "dmr2d --workload dmr2d --grid 4096" "mhd2d --workload mhd2d --grid 4096" "axi2d --workload axi2d --grid 4096" "mhdaxi2d --workload mhdaxi2d --grid 4096"; do
  set -- $wl; w=$1; shift
  pmc ${w}_fetch "FETCH_SIZE" "$@"
  pmc ${w}_write "WRITE_SIZE" "$@"
  pmc ${w}_sq "$SQ" "$@"
done
pmc m1_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum"
pmc m1_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
pmc m1_wrreq "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"
pmc m1_clk "GRBM_GUI_ACTIVE GRBM_COUNT"
# (always rebuilt: a binary of an earlier round may be lying in the tree)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $ROOT/profiles/tools/calib_traffic $ROOT/profiles/tools/calib_traffic.hip 2>/dev/null
for c in "fetch FETCH_SIZE" "write WRITE_SIZE" "rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum"; do
  set -- $c; ct=$1; shift
  echo "== calib $ct"; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_r03_calib_$ct -- $ROOT/profiles/tools/calib_traffic > $OUT/pmc_r03_calib_$ct.log 2>&1 || echo "FAILED calib $ct"
done
echo "collection done"
python3 $ROOT/profiles/tools/summarize_r03.py
