mkdir -p gpurun_out/r3l
O=gpurun_out/r3l/ab.txt
AB_ARGS="--workload m3 --grid 256" profiles/tools/ab_bench.sh m3=default m3_nofusedt=default,PION_FUSE_DT=0 m3b=default m3_nofusedt_b=default,PION_FUSE_DT=0 | tee $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3l/m3_nofuse -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build --workload m3 --grid 256 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee -a gpurun_out/r3l/ab.txt
import csv,glob
for f in glob.glob("gpurun_out/r3l/m3_nofuse/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-80s calls %4s avg %8.3f ms %6s %%"%(r["Name"][:80],r["Calls"],float(r["AverageNs"])/1e6,r["Percentage"]))
PY
