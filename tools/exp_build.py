"""Build an EXPERIMENTAL variant of libte_hip.so with extra -D flags (kernel tuning experiments on the GPU box).

    python tools/exp_build.py <name> [-DFLAG ...]      ->  tools/exp/libte_<name>.so

The product library (transeditor_amd/libte_hip.so) is never touched; tools/exp_time.py loads a variant by name.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import build as B      # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out = os.path.join(ROOT, 'tools', 'exp', f'libte_{name}.so')
    objdir = os.path.join(ROOT, 'tools', 'exp', f'obj_{name}')
    os.makedirs(objdir, exist_ok=True)
    hipcc = B._hipcc()
    procs = []
    # sources that never mention one of the macros are taken from the product build's objects (transeditor_amd/build/*.o)
    macros = [f[2:].split('=')[0] for f in flags if f.startswith('-D')]
    B.build(verbose=False)
    objs = []
    for src in B.SOURCES:
        prod_obj = os.path.join(B.HERE, 'build', src.replace('.hip', '.o'))
        text = open(os.path.join(B.CSRC, src)).read() + ''.join(
            open(os.path.join(B.CSRC, h)).read() for h in os.listdir(B.CSRC) if h.endswith('.h'))
        if macros and os.path.exists(prod_obj) and not any(m in text for m in macros) and \
                os.path.getmtime(prod_obj) >= os.path.getmtime(os.path.join(B.CSRC, src)):
            objs.append(prod_obj)
            continue
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        procs.append((src, obj, subprocess.Popen([hipcc, *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *flags, '-c', os.path.join(B.CSRC, src), '-o', obj],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, obj, p in procs:
        o, _ = p.communicate()
        if p.returncode:
            raise SystemExit(f'{src}:\n{o.decode()}')
        objs.append(obj)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out])
    subprocess.call(['rm', '-rf', objdir])
    print('built', out)


if __name__ == '__main__':
    main()
