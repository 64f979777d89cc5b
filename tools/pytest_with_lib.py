"""run the test suite against an experimental library variant:  python tools/pytest_with_lib.py <variant> <pytest args...>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from transeditor_amd import _lib      # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f'libte_{sys.argv[1]}.so')
import pytest                          # noqa: E402

sys.exit(pytest.main(sys.argv[2:]))
