#!/bin/bash
# tools/exp/res.sh <out> <src.hip> [extra flags]: compile for gfx950 and print registers / scratch / LDS per kernel
out=$1; src=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage "$@" -o $out $src 2>&1 | python3 -c "
import sys,re
cur=None;rows={}
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z][A-Za-z \[\]/]+): (\d+)',l)
    if m and cur: rows[cur][m.group(1).strip()]=int(m.group(2))
for k,v in rows.items():
    print('%-70s vgpr %3d agpr %3d sgpr %3d scratch %3d lds %6d occ %d'%(k[:70],v.get('VGPRs',-1),v.get('AGPRs',-1),v.get('TotalSGPRs',-1),v.get('ScratchSize [bytes/lane]',-1),v.get('LDS Size [bytes/block]',-1),v.get('Occupancy [waves/SIMD]',-1)))
"
