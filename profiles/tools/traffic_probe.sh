#!/bin/bash
# L2-miss read traffic of the stage kernels for one build / environment (run through gpurun from the repo root):
#   profiles/tools/traffic_probe.sh TAG [bench args...]     (PION_GPU_LIB and the PION_* switches pass through)
# one rocprofv3 --pmc FETCH_SIZE pass (with --kernel-trace only); bytes = reported KB x 1024 x 2.0 (gfx950 read
# correction, profiles/r02_pmc_traffic.json "calibration")
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tp_$TAG
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/tp_$TAG -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build "$@" > $OUT/tp_$TAG.log 2>&1 || { echo "pmc run failed"; tail -5 $OUT/tp_$TAG.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/tp_$TAG/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stage_rows" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0].replace("void pion::", "")][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    print("$TAG %-55s read %.2f GB per launch (%d launches)" % (k[:55], sum(d.values()) / len(d) * 1024 * 2.0 / 1e9, len(d)))
PY
