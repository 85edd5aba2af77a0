#!/bin/bash
# usage: ab.sh label lib [env...]
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "default" ]; then unset PION_GPU_LIB; else export PION_GPU_LIB=$lib; fi
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', 'value %.1f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'dt_ms %.2f' % d['roofline']['dt_ms'])
"
done
