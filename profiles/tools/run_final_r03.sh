#!/bin/bash
# Round-3 final measurement set on one GPU box (through gpurun): plain bench runs of every workload, the slab /
# loop-back shapes of an N-rank run, the N = 2 rehearsal of bench.py over the shared-memory transport.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/final_r03
mkdir -p $O
cd $ROOT
python bench.py > $O/default_run.json 2> $O/default_run.err
python bench.py --workload m2 --no-cpu-baseline > $O/bench_m2.json 2>/dev/null
python bench.py --workload m3 --grid 256 --no-cpu-baseline > $O/bench_m3.json 2>/dev/null
python bench.py --eqn mhd --no-cpu-baseline > $O/bench_mhd8.json 2>/dev/null
python bench.py --workload dmr2d --grid 4096 --no-cpu-baseline > $O/bench_dmr2d.json 2>/dev/null
python bench.py --workload mhd2d --grid 4096 --no-cpu-baseline > $O/bench_mhd2d.json 2>/dev/null
python bench.py --workload axi2d --grid 4096 --no-cpu-baseline > $O/bench_axi2d.json 2>/dev/null
python bench.py --workload mhdaxi2d --grid 4096 --no-cpu-baseline > $O/bench_mhdaxi2d.json 2>/dev/null
for nz in 256 128 64; do
  python bench.py --nz $nz --no-cpu-baseline --no-parity-build > $O/slab_nz${nz}.json 2>/dev/null
  python bench.py --nz $nz --loopback --no-cpu-baseline --no-parity-build > $O/loop_nz${nz}.json 2>/dev/null
done
timeout -k 10 300 python bench.py --gpus 2 --transport shm --steps 5 --warmup 2 > $O/bench_gpus2_shm.json 2> $O/bench_gpus2_shm.err
for f in $O/*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],3),'ms/step frac', round(d['roofline']['frac'],3), 'step', round(d['roofline']['step_frac'],3), d['config'].get('transport','')[:40])
except Exception as e: print('FAILED', e)
")"; done
