#!/bin/bash
# Register / scratch / instruction-mix probe of the production stage-kernel instances (no GPU needed):
#   profiles/tools/probe_regs.sh [eqsel=3] [extra hipcc flags...]
# compiles pion_amd/csrc/kernels_fp.hip with -DPION_PROBE (production instances only) to ISA under /tmp/isa.
set -e
EQ=${1:-3}; shift || true
mkdir -p /tmp/isa
HERE=$(cd "$(dirname "$0")/../.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-slp-vectorize -fPIC -std=c++17 -O2 -ffp-contract=fast -fapprox-func \
  -freciprocal-math -DPION_FAST_MATH -DPION_FPNS=fp_fast -DPION_EQSEL=$EQ -DPION_PROBE "$@" --cuda-device-only -S \
  $HERE/pion_amd/csrc/kernels_fp.hip -o /tmp/isa/probe$EQ.s 2>&1 | grep -v hip-link || true
python3 - <<PY
import re
txt=open('/tmp/isa/probe$EQ.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name=m.group(1); body=m.group(2)
    if 'k_stage_rows2' not in name: continue
    g=lambda k: re.search(r'\.amdhsa_'+k+r'\s+(\S+)',body).group(1)
    # instruction count of the kernel body
    i0=txt.index(name+':'); i1=txt.index('s_endpgm',i0)
    ins=[l.split()[0] for l in txt[i0:i1].split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.',';'))]
    nsc=sum(1 for x in ins if x.startswith('scratch_'))
    print(name[28:70],'vgpr',g('next_free_vgpr'),'sgpr',g('next_free_sgpr'),'scratch_bytes',g('private_segment_fixed_size'),'instrs',len(ins),'scratch_instrs',nsc)
PY
