#!/bin/bash
# usage: tools/regs.sh file.hip [grep-pattern]  — VGPR / scratch / occupancy per kernel
cd /root/repo/transport_analysis_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/regs_x.o 2>&1 | python3 -c "
import sys,re
name=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)', l)
    if m: name=m.group(1); d[name]={}
    for k in ('VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)', l)
        if m and name: d[name][k.split(' ')[0]]=m.group(1)
for n,v in d.items():
    short=re.sub(r'_ZN2ta\d+','',n); short=re.sub(r'INS_4Plan','',short); short=re.sub(r'EEv.*','',short)
    print(f'{short:60s} vgpr={v.get(\"VGPRs\")} scratch={v.get(\"ScratchSize\")} occ={v.get(\"Occupancy\")}')
" | grep -E "${2:-.}"
