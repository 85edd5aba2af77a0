mkdir -p gpurun_out/r3f
python -m pytest tests/test_gpu_parity.py tests/test_gpu_xtile.py tests/test_gpu_golden.py tests/test_gpu_split_stage.py -m gpu -x -q 2>&1 | tail -4
AB_STEPS=6 profiles/tools/ab_bench.sh cp=default nocp=ab/nocopies/libpion_gpu.so cp2=default nocp2=ab/nocopies/libpion_gpu.so 2>&1 | tee gpurun_out/r3f/ab.txt
AB_ARGS="--workload m2" profiles/tools/ab_bench.sh m2cp=default m2nocp=ab/nocopies/libpion_gpu.so | tee -a gpurun_out/r3f/ab.txt
AB_ARGS="--workload m3 --grid 256" profiles/tools/ab_bench.sh m3cp=default m3nocp=ab/nocopies/libpion_gpu.so | tee -a gpurun_out/r3f/ab.txt
echo "--- slab 512x512x64, uneven chunks on/off, zchunk variants" | tee -a gpurun_out/r3f/ab.txt
AB_ARGS="--nz 64" profiles/tools/ab_bench.sh slab_uneven=default slab_even=default,PION_UNEVEN_CHUNKS=0 slab_even8=default,PION_UNEVEN_CHUNKS=0,PION_ZCHUNK=8 slab_un16=default,PION_ZCHUNK=16 | tee -a gpurun_out/r3f/ab.txt
AB_ARGS="--nz 64 --loopback" profiles/tools/ab_bench.sh loop_uneven=default loop_even=default,PION_UNEVEN_CHUNKS=0 | tee -a gpurun_out/r3f/ab.txt
echo "--- full, uneven off" | tee -a gpurun_out/r3f/ab.txt
profiles/tools/ab_bench.sh full_even=default,PION_UNEVEN_CHUNKS=0 | tee -a gpurun_out/r3f/ab.txt
echo "--- 2-D" | tee -a gpurun_out/r3f/ab.txt
python bench.py --workload dmr2d --grid 4096 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/dmr2d.json 2> gpurun_out/r3f/dmr2d.err; tail -c 600 gpurun_out/r3f/dmr2d.err; python -c "
import json;d=json.load(open('gpurun_out/r3f/dmr2d.json'));print('dmr2d',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['step_frac'],d['roofline']['kernel_ms'],d.get('parity_build',{}).get('value'))"
python bench.py --workload mhd2d --grid 4096 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/mhd2d.json 2> gpurun_out/r3f/mhd2d.err; tail -c 600 gpurun_out/r3f/mhd2d.err; python -c "
import json;d=json.load(open('gpurun_out/r3f/mhd2d.json'));print('mhd2d',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['step_frac'],d['roofline']['kernel_ms'],d.get('parity_build',{}).get('value'))"
