#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq_$1 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --n ${2:-512} > $OUT/pmc_sq_$1.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_sq_$1/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stage_rows" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
import json
out = {}
for k, d in sorted(acc.items()):
    vals = [sum(v) for v in d.values()]
    out[k] = sum(vals) / len(vals)
    print("$1", k, "%.4g" % out[k], "n=%d" % len(vals))
json.dump({"what": "SQ counters per k_stage_rows launch (mean of the first- and second-order stage instances), "
                   "bench.py --steps 2 --warmup 1 --grid ${2:-512}, rocprofv3 --pmc (one pass, with --kernel-trace only)",
           "counters": out}, open("$OUT/pmc_sq_$1.json", "w"), indent=1)
PY
