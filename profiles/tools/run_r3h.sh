mkdir -p gpurun_out/r3h
O=gpurun_out/r3h/ab.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_xtile.py tests/test_gpu_golden.py tests/test_gpu_split_stage.py tests/test_endstate.py tests/test_cooling_reference.py tests/test_shock_tubes.py -m gpu -x -q 2>&1 | tail -6
AB_STEPS=6 profiles/tools/ab_bench.sh m1=default m1b=default 2>&1 | tee $O
AB_ARGS="--workload m2" profiles/tools/ab_bench.sh m2=default | tee -a $O
AB_ARGS="--workload m3 --grid 256" profiles/tools/ab_bench.sh m3=default m3even=default,PION_UNEVEN_CHUNKS=0 | tee -a $O
AB_ARGS="--eqn mhd" profiles/tools/ab_bench.sh mhd8=default | tee -a $O
echo "--- 2-D" | tee -a $O
for spec in "dmr2d 4096" "mhd2d 4096"; do set -- $spec
 for env in "PION_ROWS_2D=0" "PION_ROWS=4" "PION_ROWS=8" "PION_ROWS=16" "PION_ROWS=32"; do
  env $env python bench.py --workload $1 --grid $2 --steps 10 --warmup 2 --no-cpu-baseline --no-parity-build 2>gpurun_out/r3h/err_$1.txt | python -c "
import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1 $env', 'value %.1f'%d['value'],'ms/step %.3f'%d['ms_per_step'],'kfrac %.3f'%d['roofline']['frac'],'stepfrac %.3f'%d['roofline']['step_frac'],'kernel_ms %.3f'%d['roofline']['kernel_ms'])" | tee -a $O
 done
done
tail -3 gpurun_out/r3h/err_mhd2d.txt
echo "--- slab kernel breakdown" | tee -a $O
profiles/tools/quick_prof.sh r3h_slab --nz 64 2>&1 | tee -a $O | tail -12
profiles/tools/quick_prof.sh r3h_full 2>&1 | tee -a $O | tail -12
