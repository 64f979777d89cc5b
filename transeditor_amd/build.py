"""Build libte_hip.so (hand-written gfx950 kernels behind the C ABI of include/te_hip.h) in-tree.

    python -m transeditor_amd.build          # hipcc cross-compiles without a GPU

The shared object lands next to this file (git-ignored, but shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libte_hip.so')
SOURCES = ['te_common.hip', 'bias_act.hip', 'upfirdn2d.hip', 'conv.hip', 'wino.hip', 'wino6.hip', 's2s6.hip', 't2s6.hip', 'p1s6.hip', 'wgrad.hip', 'wgrad6.hip', 'attention.hip', 'rgb.hip', 'linear.hip', 'style.hip', 'layernorm.hip', 'optim.hip', 'stddev.hip', 'chanscale.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-pass-failed']
# per-source flags.  wino6.hip: the staging arithmetic of the ping-pong kernel is written as scalar fp32 steps placed between MFMAs; the
# SLP vectoriser would re-pack them into v_pk_* forms that need register shuffles and wait states
EXTRA_FLAGS = {'wino6.hip': ['-fno-slp-vectorize'], 's2s6.hip': ['-fno-slp-vectorize'], 't2s6.hip': ['-fno-slp-vectorize'],
               'p1s6.hip': ['-fno-slp-vectorize'],
               'wgrad6.hip': ['-fno-slp-vectorize']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'te_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        procs.append((src, obj, subprocess.Popen([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), '-c', path, '-o', obj],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out.decode()}')
        objs.append(obj)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB])
    if verbose:
        print(f'built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB) from {len(objs)} sources', file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
