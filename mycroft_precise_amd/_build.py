"""
In-tree build of libprecise_engine.so with hipcc for gfx950 (MI355X) only.

    python -m mycroft_precise_amd._build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
is git-ignored but travels to the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ_DIR = os.path.join(CSRC, 'build')
LIB_PATH = os.path.join(HERE, 'libprecise_engine.so')
SOURCES = ['engine.hip', 'kernels.hip']
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + \
    [os.path.join(os.path.dirname(HERE), 'include', 'precise_engine.h')]
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx950's register file is unified): the gate arithmetic
# reads them with VALU instructions every timestep, and from AGPRs each read is an extra v_accvgpr_read
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-mllvm', '-amdgpu-mfma-vgpr-form=1']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libprecise_engine.so')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library. Returns its path."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ_DIR, src.replace('.hip', '.o'))
        if force or _stale(op, [sp] + HEADERS):
            jobs.append((sp, op))

    def compile_one(job):
        sp, op = job
        cmd = [hipcc] + FLAGS + ['-c', sp, '-o', op]
        if verbose:
            print(' '.join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (sp, res.stderr))
        return res.stderr

    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        for warn in ex.map(compile_one, jobs):
            if warn.strip() and verbose:
                print(warn, file=sys.stderr)

    objs = [os.path.join(OBJ_DIR, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-o', LIB_PATH] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('link failed:\n' + res.stderr)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
