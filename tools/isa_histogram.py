#!/usr/bin/env python3
"""Opcode histogram of one kernel of kernels.hip (gfx950 ISA as the compiler emits it): how many VALU / packed / LDS /
vector-memory / MFMA instructions a code path is made of.     python tools/isa_histogram.py <regex on the demangled name> [top]"""
import collections, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = '/tmp/pe_kernels.s'
src = os.path.join(REPO, 'mycroft_precise_amd/csrc/kernels.hip')
if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(os.path.join(REPO, 'mycroft_precise_amd/csrc', f))
                                                          for f in os.listdir(os.path.join(REPO, 'mycroft_precise_amd/csrc'))):
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', os.path.join(REPO, 'include'),
                    '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-S', '--cuda-device-only', src, '-o', asm], check=True)
text = open(asm).read()
pat = re.compile(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
labels = list(re.finditer(r'^(_Z\w+):[^\n]*\n', text, re.M))
for i, m in enumerate(labels):
    body = text[m.end():labels[i + 1].start() if i + 1 < len(labels) else len(text)]
    body = body.split('.section')[0]
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void pe::', '')
    if not pat.search(name):
        continue
    ops = collections.Counter(re.findall(r'^\t([a-z][a-z_0-9]+)', body, re.M))
    cls = collections.Counter()
    for op, n in ops.items():
        k = ('mfma' if 'mfma' in op else 'valu pk' if op.startswith('v_pk_') else 'valu f64' if op.startswith('v_') and op.endswith('f64')
             else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_'))
             else 'salu/branch/wait')
        cls[k] += n
    print('%s: %d instructions (static) %s' % (name, sum(ops.values()), dict(cls)))
    print('   ' + ' '.join('%s:%d' % kv for kv in sorted(ops.items(), key=lambda x: -x[1])[:top]))
