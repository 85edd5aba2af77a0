# timing-only experiment (the variants compute WRONG results): response of the second-order stage kernel to fewer bytes
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/exp_traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in default expc expz expcz; do
  if [ $v = default ]; then unset PION_GPU_LIB; else export PION_GPU_LIB=$ROOT/ab/$v/libpion_gpu.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$v -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build > $OUT/st_$v.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fe_$v -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build > $OUT/fe_$v.log 2>&1
done
python3 - <<PY
import csv,glob
for v in ("default","expc","expz","expcz"):
    dur={}
    for f in glob.glob("$OUT/st_%s/*/*_kernel_stats.csv"%v):
        for r in csv.DictReader(open(f)):
            if "k_stage_rows2" in r["Name"]: dur[r["Name"].split("(")[0][-30:]]=float(r["AverageNs"])/1e6
    acc={}
    for f in glob.glob("$OUT/fe_%s/*/*_counter_collection.csv"%v):
        per={}
        for r in csv.DictReader(open(f)):
            if "k_stage_rows2" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
                k=(r["Dispatch_Id"],r["Kernel_Name"].split("(")[0][-30:]); per[k]=per.get(k,0)+float(r["Counter_Value"])
        for (d,n),x in per.items(): acc.setdefault(n,[]).append(x)
    print(v, {k:round(x,3) for k,x in dur.items()}, {k:round(sum(x)/len(x)*1024*2/1e9,2) for k,x in acc.items()})
PY
