#!/bin/bash
# quick per-instance profile of the stage kernels on the GPU box (run through gpurun from the repo root):
#   profiles/tools/quick_prof.sh TAG [bench args...]      (environment variables pass through)
# 1. rocprofv3 --kernel-trace --stats -> per-kernel average durations
# 2. one --pmc pass with the SQ counters, reported per stage-kernel instance
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-build $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/qp_$TAG -- $BENCH > $OUT/qp_$TAG.log 2>&1 || { echo "kernel-trace run failed"; tail -5 $OUT/qp_$TAG.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/qs_$TAG -- $BENCH > $OUT/qs_$TAG.log 2>&1 || { echo "pmc run failed"; tail -5 $OUT/qs_$TAG.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
out = {"tag": "$TAG", "kernels": {}}
for f in glob.glob("$OUT/qp_$TAG/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_stage", "k_prepass", "k_bc", "k_cooling", "k_dt", "k_halo", "k_wind")):
            short = n.split("(")[0].replace("void pion::", "")
            out["kernels"][short] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
            print("%-70s calls %4s avg %8.3f ms  %5s %%" % (short[:70], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob("$OUT/qs_$TAG/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stage_rows" in r["Kernel_Name"]:
            short = r["Kernel_Name"].split("(")[0].replace("void pion::", "")
            acc[short][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out["sq"] = {}
for k, d in acc.items():
    m = {c: sum(v.values()) / len(v) for c, v in d.items()}
    out["sq"][k] = m
    wc = m.get("SQ_WAVE_CYCLES", 1.0)
    print(k[:60], "VALU insts %.4g  valu_active/wave_cyc %.3f  wait_any %.3f  wait_inst %.3f  salu %.4g vmem_rd %.4g busy_cyc %.4g" % (
        m.get("SQ_INSTS_VALU", 0), m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc,
        m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_BUSY_CYCLES", 0)))
json.dump(out, open("$OUT/quick_$TAG.json", "w"), indent=1)
PY
