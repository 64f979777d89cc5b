"""Drop-in for the reference's `utils` package.  The reference's `utils/` is a namespace package (no __init__.py); this
regular package takes its place at the front of `PYTHONPATH` and then EXTENDS its search path with every other `utils`
directory on sys.path, so `utils.op`, `utils.sample`, `utils.distributed` and `utils.dataset` resolve here (MI355X path)
while `utils.lpips`, `utils.editing_utils`, `utils.dataset_projector` keep resolving to the reference's own files."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
