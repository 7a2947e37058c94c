#!/bin/bash
# registers / LDS / scratch / occupancy of the kernels of one source file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kres.sh scatter.hip [kernel-regex] [extra flags]
cd /root/repo/humanrf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -munsafe-fp-atomics $3 -x hip -c $1 -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
rx=re.compile(sys.argv[1] if len(sys.argv)>1 and sys.argv[1] else '.')
cur=None;d={}
for line in sys.stdin:
    m=re.search(r'remark: (.*)',line)
    if not m: continue
    t=m.group(1).strip()
    m2=re.match(r'Function Name: (\S+)',t)
    if m2:
        if cur and rx.search(cur): print(cur[:70],d)
        cur=m2.group(1);d={}
        continue
    m3=re.match(r'(\S[^:]*): (\S+)',t)
    if m3 and cur:
        k=m3.group(1)
        if k in('VGPRs','AGPRs','SGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','SGPRs Spill','VGPRs Spill','LDS Size [bytes/block]'):
            d[k.split(' ')[0]+('Spill' if 'Spill' in k else '')]=m3.group(2)
if cur and rx.search(cur): print(cur[:70],d)
" "$2"
