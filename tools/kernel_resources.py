#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, AGPRs, SGPRs, spills, LDS, occupancy) of kernels.hip for gfx950.
    python tools/kernel_resources.py [filter-regex] [extra hipcc flags...]"""
import re, subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else '.'
flags = sys.argv[2:]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c',
       os.path.join(REPO, 'mycroft_precise_amd/csrc/kernels.hip'), '-o', '/tmp/kernels_res.o',
       '-Rpass-analysis=kernel-resource-usage'] + flags
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith('Function Name:'):
        cur = body.split(':', 1)[1].strip(); rows[cur] = {}
    elif cur and ':' in body:
        k, v = body.split(':', 1); rows[cur][k.strip()] = v.strip()
def demangle(n):
    return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
print('%-64s %5s %5s %5s %6s %7s %4s' % ('kernel', 'VGPR', 'AGPR', 'SGPR', 'spill', 'LDS', 'occ'))
for n, r in rows.items():
    d = re.sub(r'\(.*', '', demangle(n)).replace('void pe::', '')
    if not re.search(pat, d):
        continue
    print('%-64s %5s %5s %5s %6s %7s %4s' % (d[:64], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'),
          r.get('ScratchSize [bytes/lane]', r.get('VGPR Spill', '0')), r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))
