"""Import the reference's model_spatial_query.py in the BUILD CONTAINER (never on the GPU box).

TEST INFRASTRUCTURE.  The reference has no CPU path (utils/op JIT-compiles CUDA at import,
Generator.forward hard-codes .cuda()), so three import-time shims are installed;
no reference file is edited or copied:
  1. a stub `torchvision` (only the dead Vgg19 class touches it, model_spatial_query.py:6,17);
  2. a stub `utils.op` providing FusedLeakyReLU / fused_leaky_relu / upfirdn2d.  upfirdn2d runs
     the reference's OWN pure-PyTorch `upfirdn2d_native` (utils/op/upfirdn2d.py:151-185), pulled
     out of the file by AST at run time (the module itself cannot be imported: it compiles CUDA at
     import, and the function references an un-imported `F`); fused_leaky_relu has no native
     twin in the reference, so it is the oracle's restatement of fused_bias_act_kernel.cu:26-47 —
     the only arithmetic in the golden vectors that is not the reference's own code;
  3. torch.Tensor.cuda -> identity (model_spatial_query.py:630,642).
"""
import os
import sys
import types

import torch

REF_ROOT = '/root/reference'


def available():
    return os.path.isfile(os.path.join(REF_ROOT, 'model_spatial_query.py'))


def reference_upfirdn2d_native():
    """Return the reference's `upfirdn2d_native` function object (executed from its own source
    file in place; nothing is copied into this repo)."""
    import ast
    import torch.nn.functional as F
    path = os.path.join(REF_ROOT, 'utils', 'op', 'upfirdn2d.py')
    src = open(path).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'upfirdn2d_native'][0]
    ns = {'torch': torch, 'F': F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    return ns['upfirdn2d_native']


def reference_upfirdn2d():
    """upfirdn2d(input[B,C,H,W], kernel, up, down, pad) with the calling convention of
    utils/op/upfirdn2d.py:143-148 / :97 (input viewed [B*C,H,W,1]) on top of the native body."""
    native = reference_upfirdn2d_native()

    def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
        B, C, H, W = x.shape
        y = native(x.reshape(-1, H, W, 1), kernel.to(x.dtype), up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        return y.reshape(B, C, y.shape[1], y.shape[2])
    return upfirdn2d


def reference_train_functions():
    """The loss / bookkeeping functions of the reference's train_spatial_query.py (:49-105), executed from the
    reference's own file via AST (the module itself imports torchvision / tensorboard / lmdb at top level)."""
    import ast
    import math
    import torch.nn.functional as F
    from torch import autograd
    path = os.path.join(REF_ROOT, 'train_spatial_query.py')
    tree = ast.parse(open(path).read())
    want = {'requires_grad', 'accumulate', 'd_logistic_loss', 'd_r1_loss', 'g_nonsaturating_loss', 'g_path_regularize'}
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {'torch': torch, 'F': F, 'autograd': autograd, 'math': math}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), ns)
    return types.SimpleNamespace(**{k: ns[k] for k in want})


def import_reference():
    if 'model_spatial_query' in sys.modules and getattr(sys.modules['model_spatial_query'], '_te_ref', False):
        return sys.modules['model_spatial_query']
    from . import te_oracle as O

    tv, tvm = types.ModuleType('torchvision'), types.ModuleType('torchvision.models')
    tv.models = tvm
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tvm)

    class FusedLeakyReLU(torch.nn.Module):
        def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = torch.nn.Parameter(torch.zeros(channel)) if bias else None
            self.negative_slope, self.scale = negative_slope, scale

        def forward(self, x):
            return O.fused_leaky_relu(x, self.bias, self.negative_slope, self.scale)

    op = types.ModuleType('utils.op')
    op.FusedLeakyReLU = FusedLeakyReLU
    op.fused_leaky_relu = O.fused_leaky_relu
    op.upfirdn2d = reference_upfirdn2d()
    utils = types.ModuleType('utils')
    utils.__path__ = [os.path.join(REF_ROOT, 'utils')]
    utils.op = op
    saved = {k: sys.modules.get(k) for k in ('utils', 'utils.op')}
    sys.modules['utils'], sys.modules['utils.op'] = utils, op
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF_ROOT)
    try:
        import model_spatial_query as M
    finally:
        sys.path.remove(REF_ROOT)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    M._te_ref = True
    return M
