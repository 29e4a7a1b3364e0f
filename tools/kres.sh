#!/bin/bash
# usage: tools/kres.sh <file.hip> [grep pattern] -- one line per kernel: registers, spills, scratch, LDS (gfx950 cross-compile)
f=$1; pat=${2:-.}
extra=""
case "$f" in *bf16*) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-result $extra \
  -I"$(dirname "$f")" -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
cur={}
for l in sys.stdin:
    m=re.search(r'remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)',l)
    if not m: continue
    k,v=m.groups()
    if k=='Function Name':
        cur={'name':v}
    else:
        cur[k]=v
    if k.startswith('LDS'):
        print('%-70s V %3s A %3s spillV %3s scratch %4s occ %s lds %s'%(cur['name'][:70],cur.get('VGPRs'),cur.get('AGPRs'),cur.get('VGPRs Spill'),cur.get('ScratchSize [bytes/lane]'),cur.get('Occupancy [waves/SIMD]'),v))
" | grep -E "$pat"
