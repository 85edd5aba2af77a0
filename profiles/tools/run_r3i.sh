mkdir -p gpurun_out/r3i
O=gpurun_out/r3i/ab.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_xtile.py tests/test_gpu_split_stage.py tests/test_gpu_host_rccl.py tests/test_gpu_rccl_loopback.py -m gpu -x -q 2>&1 | tail -4
AB_ARGS="--nz 64" profiles/tools/ab_bench.sh slab=default slab_b=default | tee $O
AB_ARGS="--nz 64 --loopback" profiles/tools/ab_bench.sh loop=default loop_b=default | tee -a $O
profiles/tools/ab_bench.sh m1=default | tee -a $O
AB_ARGS="--workload dmr2d --grid 4096" profiles/tools/ab_bench.sh dmr2d=default | tee -a $O
AB_ARGS="--workload mhd2d --grid 4096" profiles/tools/ab_bench.sh mhd2d=default | tee -a $O
AB_ARGS="--workload mhd2d --grid 1024" profiles/tools/ab_bench.sh mhd2d_1024=default | tee -a $O
echo "--- loopback kernel breakdown" | tee -a $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3i/loop_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build --nz 64 --loopback > $GRAFT_REPO_ROOT/gpurun_out/r3i/loop_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee -a gpurun_out/r3i/ab.txt
import csv,glob
for f in glob.glob("gpurun_out/r3i/loop_prof/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %4s avg %8.3f ms %6s %%"%(r["Name"][:90],r["Calls"],float(r["AverageNs"])/1e6,r["Percentage"]))
PY
python3 profiles/tools/trace_gaps.py gpurun_out/r3i/loop_prof 2>&1 | tail -15 | tee -a gpurun_out/r3i/ab.txt
