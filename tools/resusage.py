#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/occupancy per kernel: python tools_resusage.py file.hip [filter]"""
import re, subprocess, sys
src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-munsafe-fp-atomics',
                      '-ffp-contract=fast', '-c', src, '-o', '/tmp/_ru.o', '-Rpass-analysis=kernel-resource-usage'],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass', line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        cur = subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip(); rows[cur] = {}
    elif cur: rows[cur][k.split(' ')[0]] = v
for k, v in rows.items():
    if filt in k: print(k[:90].ljust(92), v)
