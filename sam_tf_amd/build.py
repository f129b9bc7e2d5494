"""Build libsamx.so (HIP, gfx950) in-tree:  python -m sam_tf_amd.build

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
tree; nothing is JIT-compiled at run time.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(HERE, 'libsamx.so')
SOURCES = ['core.hip', 'tree.hip', 'replay.hip', 'optim.hip', 'gemm.hip', 'nets.hip',
           'fused_small.hip', 'fused_ln.hip', 'fused_wide.hip', 'step.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-function'] + os.environ.get('SAMX_EXTRA_FLAGS', '').split()


def _sources():
    """What the build depends on: the HIP sources and headers only (not objects or compiler by-products)."""
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    deps.append(os.path.join(HERE, '..', 'include', 'samx.h'))
    return deps


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in _sources())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(obj)
        if (not force and os.path.exists(obj) and
                os.path.getmtime(obj) > max(os.path.getmtime(d) for d in _sources())):
            continue
        cmd = [hipcc] + FLAGS + ['-c', path, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', SO]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SO


if __name__ == '__main__':
    build(force='--force' in sys.argv)
