"""Op layer with the reference's call surface (utils/op/__init__.py:1-2):
    from transeditor_amd.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d
backed by hand-written gfx950 kernels (libte_hip.so).  Both ops are differentiable twice.
"""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d

__all__ = ['FusedLeakyReLU', 'fused_leaky_relu', 'upfirdn2d']
