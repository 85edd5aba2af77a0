#!/bin/bash
# A/B runs of bench.py on one box: profiles/tools/ab_bench.sh label=lib[,ENV=VAL...] ...   (lib "default" = the in-tree build)
# An alternative library must carry the soname libpion_gpu.so (csrc/Makefile links it so) to serve libpion_host.so too.
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}
  envs=""; if [ "$rest" != "$lib" ]; then envs=${rest#*,}; fi
  (
    if [ "$lib" = "default" ]; then unset PION_GPU_LIB; else export PION_GPU_LIB=$lib; fi
    IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    python bench.py --steps ${AB_STEPS:-4} --warmup 1 --no-cpu-baseline --no-parity-build ${AB_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'value %.1f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])
"
  )
done
