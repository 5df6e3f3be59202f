#!/bin/bash
# Here (no GPU needed): registers, spills, scratch and occupancy of every kernel of one device source, from hipcc -Rpass-analysis=kernel-resource-usage.
# usage: bash scripts/kernel_resource_usage.sh cpugpupathtracing_amd/csrc/device/<file>.hip [extra hipcc flags]
F=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fPIC -I$(cd "$(dirname "$0")/.." && pwd)/include -I$(cd "$(dirname "$0")/.." && pwd)/cpugpupathtracing_amd/csrc/host -I$(cd "$(dirname "$0")/.." && pwd)/cpugpupathtracing_amd/csrc/device --offload-arch=gfx950 -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage "$@" -c $F -o /tmp/kernel_resource_usage.o 2>&1 | python3 -c "
import sys,re
cur=None;d={}
for l in sys.stdin:
    m=re.search(r'remark: (.*)',l)
    if not m: continue
    s=m.group(1).strip()
    if s.startswith('Function Name:'): cur=s.split(':',1)[1].replace('[-Rpass-analysis=kernel-resource-usage]','').strip(); d[cur]={}
    elif cur and ':' in s:
        k,v=s.split(':',1); d[cur][k.strip()]=v.replace('[-Rpass-analysis=kernel-resource-usage]','').strip()
for k,v in d.items():
    print(k[:60].ljust(60), 'VGPR',v.get('VGPRs'),'AGPR',v.get('AGPRs'),'SGPR',v.get('SGPRs') or v.get('TotalSGPRs'),'sspill',v.get('SGPRs Spill'),'vspill',v.get('VGPRs Spill'),'scratch',v.get('ScratchSize [bytes/lane]'),'occ',v.get('Occupancy [waves/SIMD]'),'lds',v.get('LDS Size [bytes/block]'))
"
