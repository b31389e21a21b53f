#!/bin/bash
# usage: scripts/kernel_resources.sh <file.hip> [kernel-name-pattern] — VGPRs / scratch bytes per lane / waves per SIMD / LDS of every kernel (hipcc's own remarks; no GPU needed)
F=$1; PAT=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$F" -o /tmp/_res_chk.o 2>&1 | python3 -c "
import re, sys
cur = None
for l in sys.stdin:
    if 'error' in l: print(l.rstrip())
    m = re.search(r'remark: +(.*?) \[-Rpass', l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(': ', 1)[1]}
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
        if k.strip().startswith('LDS Size'):
            if re.search(sys.argv[1], cur['name']):
                print(cur['name'][:70], 'vgpr', cur.get('VGPRs'), 'scratch', cur.get('ScratchSize [bytes/lane]'), 'occ', cur.get('Occupancy [waves/SIMD]'), 'lds', cur.get('LDS Size [bytes/block]'))
            cur = None
" "$PAT"
