mkdir -p gpurun_out/r3g
O=gpurun_out/r3g/ab.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_xtile.py tests/test_gpu_golden.py tests/test_gpu_split_stage.py tests/test_endstate.py -m gpu -x -q 2>&1 | tail -4
AB_STEPS=6 profiles/tools/ab_bench.sh uneven=default even=default,PION_UNEVEN_CHUNKS=0 nocp=ab/nocopies/libpion_gpu.so uneven2=default even2=default,PION_UNEVEN_CHUNKS=0 nocp2=ab/nocopies/libpion_gpu.so 2>&1 | tee $O
AB_ARGS="--workload m2" profiles/tools/ab_bench.sh m2=default m2even=default,PION_UNEVEN_CHUNKS=0 m2nocp=ab/nocopies/libpion_gpu.so | tee -a $O
AB_ARGS="--workload m3 --grid 256" profiles/tools/ab_bench.sh m3=default m3even=default,PION_UNEVEN_CHUNKS=0 m3nocp=ab/nocopies/libpion_gpu.so | tee -a $O
echo "--- slab 512x512x64" | tee -a $O
AB_ARGS="--nz 64" profiles/tools/ab_bench.sh slab_uneven=default slab_even=default,PION_UNEVEN_CHUNKS=0 slab_un16=default,PION_ZCHUNK=16 | tee -a $O
AB_ARGS="--nz 64 --loopback" profiles/tools/ab_bench.sh loop_uneven=default loop_even=default,PION_UNEVEN_CHUNKS=0 | tee -a $O
echo "--- 2-D" | tee -a $O
for spec in "dmr2d 4096" "mhd2d 4096"; do set -- $spec
 for env in "PION_ROWS_2D=0" "PION_ROWS=4" "PION_ROWS=8" "PION_ROWS=16" "PION_ROWS=32"; do
  env $env python bench.py --workload $1 --grid $2 --steps 10 --warmup 2 --no-cpu-baseline --no-parity-build 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1 $env', 'value %.1f'%d['value'],'ms/step %.3f'%d['ms_per_step'],'kfrac %.3f'%d['roofline']['frac'],'stepfrac %.3f'%d['roofline']['step_frac'],'kernel_ms %.3f'%d['roofline']['kernel_ms'])" | tee -a $O
 done
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r3g/counters_avail.txt 2>&1; grep -c . $GRAFT_REPO_ROOT/gpurun_out/r3g/counters_avail.txt
