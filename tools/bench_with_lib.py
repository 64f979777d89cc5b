"""bench.py against an experimental library variant:  python tools/bench_with_lib.py <variant|product> <bench args...>"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

if sys.argv[1] != 'product':
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f'libte_{sys.argv[1]}.so')
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
