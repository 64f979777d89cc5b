"""utils/op/__init__.py:1-2 of the reference -> the gfx950 kernels (no JIT compile at import)."""
from transeditor_amd.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d                          # noqa: F401
