"""One TransEditor training iteration on the MI355X path (BASELINE configs 3 and 4).

Semantics follow the reference's `train()` loop, train_spatial_query.py:166-294 (losses :70-105, optimiser
setup :458-473, EMA :56-61), restated as a small class instead of one long function:

    D step      G frozen -> fake batch (no graph) ; softplus logistic loss on D(fake), D(real)           :173-194
    R1          every d_reg_every: ||dD(real)/dreal||^2, weighted r1/2 * d_reg_every (+ 0 * real_pred[0])   :196-206
    G step      D frozen -> non-saturating loss                                                            :210-224
    path reg    every g_reg_every on batch // path_batch_shrink: path-length penalty (double backward)     :226-250
    EMA         g_ema <- decay * g_ema + (1 - decay) * g,  decay = 0.5 ** (32 / 10000)                     :294

Data parallelism (config 4): one process per GPU, `GradSync` mean all-reduce of the gradients after every
backward (what the reference's two DDP wrappers do, :494-509) and the scalar collectives of utils/distributed.
`--spatial_regu` (the P-space path regulariser, :252-285; off in every BASELINE config) is `spatial_step`.
"""
import copy
import math
from types import SimpleNamespace

import torch
from torch import autograd
from torch.nn import functional as F

from .model_spatial_query import Discriminator, Generator
from .op.modconv import no_weight_grads, packed_weights_cache, refresh_packed_weights, second_order
from .optim import FusedAdam, MultiTensorEMA
from .utils import distributed as D
from .utils.sample import prepare_noise_new, prepare_param


def default_args(**kw):
    """The reference's CLI defaults that matter for one iteration (train_spatial_query.py:377-433)."""
    a = dict(size=256, batch=16, para_num=16, latent=512, r1=10.0, path_regularize=2.0, path_batch_shrink=2,
             d_reg_every=16, g_reg_every=4, lr=0.002, channel_multiplier=2, num_trans=8, pixel_norm_op_dim=1,
             spatial_regu=False, regu_sapce='p+', spatial_path_regularize=2.0,      # (`regu_sapce`: the reference's spelling, :406)
             d_joint=True)       # extension: D's fake and real passes as one batch of 2B (Discriminator.forward(chunks=2))
    a.update(kw)
    a['token'] = 2 * (int(math.log2(a['size'])) - 1)
    return SimpleNamespace(**a)


def d_logistic_loss(real_pred, fake_pred):                                   # :70-74
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def d_r1_loss(real_pred, real_img):                                          # :77-83
    with no_weight_grads():                                                  # only d/d(real_img) is requested here
        grad_real, = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def g_nonsaturating_loss(fake_pred):                                         # :86-89
    return F.softplus(-fake_pred).mean()


def g_path_regularize(fake_img, latents, mean_path_length, noise, decay=0.01):   # :92-105 (noise = randn_like(img))
    noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    with no_weight_grads():                                                  # only d/d(latents) is requested here
        grad, = autograd.grad(outputs=(fake_img * noise).sum(), inputs=latents, create_graph=True)
    path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    path_penalty = (path_lengths - path_mean).pow(2).mean()
    return path_penalty, path_mean.detach(), path_lengths


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def accumulate(model1, model2, decay=0.999):                                 # :56-61, as ONE multi-tensor launch
    """model1 <- decay * model1 + (1 - decay) * model2 over all parameters (te_mt_ema_f32)."""
    MultiTensorEMA(model1, model2).update(decay)


def load_checkpoint_into(ckpt, g_ema, generator=None, discriminator=None, g_optim=None, d_optim=None, device=None, g_ema_only_ok=False):
    """The reference's restore (train_spatial_query.py:475-492; test_spatial_query.py:285 for a 'g_ema'-only file) on any set of
    modules / optimisers: `ckpt` is a path or a loaded dictionary in the layout :361-371 writes.  -> the start iteration parsed
    from the file name ('790000.pt' -> 790000), or None.
    A TRAINING restore (all four training objects given) from a file that holds only 'g_ema' raises KeyError, as the reference's
    `ckpt['g']` does (:487): resuming at iteration N with freshly initialised G, D and optimisers is never what was meant.
    `g_ema_only_ok=True` accepts such a file knowingly: the EMA generator alone is loaded and the start iteration returned is None."""
    import os
    start = None
    if isinstance(ckpt, (str, bytes, os.PathLike)):
        try:
            start = int(os.path.splitext(os.path.basename(ckpt))[0])
        except ValueError:
            pass
        ckpt = torch.load(ckpt, map_location=device)
    # all or nothing, like the reference's restore (:475-492): every argument is checked BEFORE anything is loaded
    training = {'g': generator, 'd': discriminator, 'g_optim': g_optim, 'd_optim': d_optim}
    given = [k for k, v in training.items() if v is not None]
    if given:
        missing_obj = [k for k, v in training.items() if v is None]
        if missing_obj:
            raise ValueError(f'load_checkpoint_into: a training restore needs generator, discriminator, g_optim and d_optim together '
                             f'(got {given}, missing {missing_obj}); pass none of them to load g_ema only')
        missing_key = [k for k in training if k not in ckpt]
        if len(missing_key) == len(training):
            # a 'g_ema'-only file (the published inference checkpoints, test_spatial_query.py:285) handed to a training restore
            if not g_ema_only_ok:
                raise KeyError("load_checkpoint_into: training objects were passed but the checkpoint holds only 'g_ema' (no 'g', 'd', "
                               "'g_optim', 'd_optim'): pass g_ema alone, or g_ema_only_ok=True to load the EMA generator and start from 0")
            given, start = [], None
        elif missing_key:
            raise KeyError(f'load_checkpoint_into: the checkpoint holds {sorted(k for k in training if k in ckpt)} but not '
                           f'{missing_key}: neither a training checkpoint (train_spatial_query.py:361-371) nor a g_ema-only file')
    elif 'g' in ckpt:
        import warnings
        warnings.warn("load_checkpoint_into: full training checkpoint ('g', 'd', 'g_optim', 'd_optim' present) loaded into g_ema only",
                      stacklevel=2)
    g_ema.load_state_dict(ckpt['g_ema'])
    if given:
        generator.load_state_dict(ckpt['g'])
        discriminator.load_state_dict(ckpt['d'])
        g_optim.load_state_dict(ckpt['g_optim'])
        d_optim.load_state_dict(ckpt['d_optim'])
    return start


class RandomSampler:
    """Random draws of one iteration; tests substitute a deterministic one."""

    def __init__(self, args, device):
        self.args, self.device = args, device

    def latents(self, n):
        return (prepare_noise_new(n, self.args, self.device, method='query'),
                prepare_param(n, self.args, self.device, method='spatial'))

    def randn_like(self, t):
        return torch.randn_like(t)


class TrainStep:
    def __init__(self, args, device, generator=None, discriminator=None, sampler=None, force_sync=None):
        """`force_sync`: run the gradient exchange machinery on a process group of one rank too (GradSync(force=...))."""
        self.args, self.device = args, device
        mk = lambda: Generator(args.size, args.latent, args.latent, args.token, channel_multiplier=args.channel_multiplier,
                               n_trans=args.num_trans, pixel_norm_op_dim=args.pixel_norm_op_dim).to(device)
        self.generator = generator if generator is not None else mk()
        self.discriminator = (discriminator if discriminator is not None
                              else Discriminator(args.size, channel_multiplier=args.channel_multiplier).to(device))
        # :452-455 — the EMA copy has the generator's own structure (also when the caller supplied a generator built with
        # other constructor options than `args` describes)
        self.g_ema = copy.deepcopy(self.generator).eval() if generator is not None else mk().eval()
        accumulate(self.g_ema, self.generator, 0)                            # :455
        g_ratio = args.g_reg_every / (args.g_reg_every + 1)
        d_ratio = args.d_reg_every / (args.d_reg_every + 1)
        # :458-473 — Adam with the lazy-regularisation ratios; one kernel launch per step (optim.FusedAdam keeps
        # torch.optim.Adam's state layout, so g_optim / d_optim checkpoints interchange)
        self.g_optim = FusedAdam(self.generator.parameters(), lr=args.lr * g_ratio, betas=(0 ** g_ratio, 0.99 ** g_ratio))
        self.d_optim = FusedAdam(self.discriminator.parameters(), lr=args.lr * d_ratio,
                                 betas=(0 ** d_ratio, 0.99 ** d_ratio))
        self._ema = MultiTensorEMA(self.g_ema, self.generator)
        self.sampler = sampler if sampler is not None else RandomSampler(args, device)
        self._packs = {}
        self.mean_path_length = 0
        self._mpl_avg = 0
        self.mean_spatial_path_length = 0
        self._mspl_avg = 0
        self.accum = 0.5 ** (32 / (10 * 1000))
        D.broadcast_module(self.generator)
        D.broadcast_module(self.discriminator)
        self.g_sync = D.GradSync(self.generator, force=force_sync)
        self.d_sync = D.GradSync(self.discriminator, force=force_sync)
        zero = torch.tensor(0.0, device=device)
        self.loss = {'r1': zero, 'path': zero, 'path_length': zero, 'spatial_path': zero, 'spatial_path_length': zero}

    # ---- the four optimisation sub-steps
    def d_step(self, real_img):
        G, Dn, a = self.generator, self.discriminator, self.args
        requires_grad(G, False)
        requires_grad(Dn, True)
        noise, param = self.sampler.latents(a.batch)
        fake_img, _, _ = G(noise, param)                                     # G frozen: no graph is built
        if getattr(a, 'd_joint', True) and fake_img.shape == real_img.shape:
            # the two passes of train_spatial_query.py:190-191 as one batch of 2B; the minibatch-stddev statistic stays per pass
            fake_pred, real_pred = Dn(torch.cat([fake_img, real_img]), chunks=2).chunk(2)
        else:
            fake_pred, real_pred = Dn(fake_img), Dn(real_img)
        d_loss = d_logistic_loss(real_pred, fake_pred)
        self.loss.update(d=d_loss.detach(), real_score=real_pred.mean().detach(), fake_score=fake_pred.mean().detach())
        Dn.zero_grad()
        d_loss.backward()
        self.d_sync.all_reduce('d')
        self.d_optim.step()
        refresh_packed_weights(self._packs)          # every packed layout of the updated model: one launch

    def r1_step(self, real_img):
        Dn, a = self.discriminator, self.args
        real_img = real_img.detach().requires_grad_(True)
        with second_order():
            real_pred = Dn(real_img)
        r1_loss = d_r1_loss(real_pred, real_img)
        Dn.zero_grad()
        (a.r1 / 2 * r1_loss * a.d_reg_every + 0 * real_pred[0]).backward()
        self.d_sync.all_reduce('r1')
        self.d_optim.step()
        refresh_packed_weights(self._packs)          # every packed layout of the updated model: one launch
        self.loss['r1'] = r1_loss.detach()

    def g_step(self):
        G, Dn, a = self.generator, self.discriminator, self.args
        requires_grad(G, True)
        requires_grad(Dn, False)
        noise, param = self.sampler.latents(a.batch)
        fake_img, _, _ = G(noise, param)
        g_loss = g_nonsaturating_loss(Dn(fake_img))
        self.loss['g'] = g_loss.detach()
        G.zero_grad()
        g_loss.backward()
        self.g_sync.all_reduce('g')
        self.g_optim.step()
        refresh_packed_weights(self._packs)          # every packed layout of the updated model: one launch

    def path_step(self):
        G, a = self.generator, self.args
        n = max(1, a.batch // a.path_batch_shrink)
        noise, param = self.sampler.latents(n)
        with second_order(wrt='latent'):          # the recorded backward stops at `latents`: only the synthesis needs the any-order route
            fake_img, latents, _ = G(noise, param, return_latents=True)
        path_loss, self.mean_path_length, path_lengths = g_path_regularize(
            fake_img, latents, self.mean_path_length, self.sampler.randn_like(fake_img))
        G.zero_grad()
        weighted = a.path_regularize * a.g_reg_every * path_loss
        if a.path_batch_shrink:
            weighted = weighted + 0 * fake_img[0, 0, 0, 0]
        weighted.backward()
        self.g_sync.all_reduce('path')
        self.g_optim.step()
        refresh_packed_weights(self._packs)          # every packed layout of the updated model: one launch
        # :249 — the reference reads this scalar back with .item() right here (a host-device sync in the middle of the step that
        # leaves the GPU waiting for the next step's first launches); it is only ever logged, so the read-back happens on access
        self._mpl_avg = D.reduce_sum(self.mean_path_length)
        self.loss.update(path=path_loss.detach(), path_length=path_lengths.mean().detach())

    def spatial_step(self):
        """:252-285 — the same path-length penalty w.r.t. the P-space input (`regu_sapce == 'p'`) or the mapped P+ code."""
        G, a = self.generator, self.args
        n = max(1, a.batch // a.path_batch_shrink)
        noise, param = self.sampler.latents(n)
        with second_order():
            if a.regu_sapce == 'p':
                wrt = param.requires_grad_()
                fake_img, _, _ = G(noise, wrt)
            else:
                # :266-268 — the mapped code stays attached (requires_grad_() on a non-leaf is a no-op in the reference too),
                # so the penalty's backward also updates the spatial mapping network
                wrt = G(noise, param, return_only_mapped_p=True)
                wrt.requires_grad_()
                fake_img, _, _ = G(noise, wrt, use_spatial_mapping=False)
        loss, self.mean_spatial_path_length, lengths = g_path_regularize(
            fake_img, wrt, self.mean_spatial_path_length, self.sampler.randn_like(fake_img))
        G.zero_grad()
        weighted = a.spatial_path_regularize * a.g_reg_every * loss
        if a.path_batch_shrink:
            weighted = weighted + 0 * fake_img[0, 0, 0, 0]
        weighted.backward()
        self.g_sync.all_reduce('spatial')
        self.g_optim.step()
        refresh_packed_weights(self._packs)          # every packed layout of the updated model: one launch
        self._mspl_avg = D.reduce_sum(self.mean_spatial_path_length)
        self.loss.update(spatial_path=loss.detach(), spatial_path_length=lengths.mean().detach())

    @property
    def mean_path_length_avg(self):
        """mean path length over ranks (train_spatial_query.py:249-251), read back from the device when asked for"""
        v = self._mpl_avg
        return (float(v) if torch.is_tensor(v) else v) / D.get_world_size()

    @property
    def mean_spatial_path_length_avg(self):
        v = self._mspl_avg
        return (float(v) if torch.is_tensor(v) else v) / D.get_world_size()

    # ---- checkpoints in the reference's layout (train_spatial_query.py:361-371 writes, :478-492 / test_spatial_query.py:285 read)
    def checkpoint(self):
        """{'g', 'd', 'g_ema', 'g_optim', 'd_optim'}: the dictionary the reference saves every 10000 iterations."""
        return {'g': self.generator.state_dict(), 'd': self.discriminator.state_dict(), 'g_ema': self.g_ema.state_dict(),
                'g_optim': self.g_optim.state_dict(), 'd_optim': self.d_optim.state_dict()}

    def save_checkpoint(self, directory, i):
        import os
        path = os.path.join(directory, f'{str(i).zfill(6)}.pt')
        torch.save(self.checkpoint(), path)
        return path

    def load_checkpoint(self, ckpt, g_ema_only_ok=False):
        """`ckpt`: a path or an already loaded dictionary.  Returns the start iteration parsed from the file name (:481-484),
        or None.  A file that only holds 'g_ema' (the published inference checkpoints) raises KeyError like the reference's
        restore (:487) unless `g_ema_only_ok=True`: then the EMA generator alone is loaded and None is returned."""
        return load_checkpoint_into(ckpt, self.g_ema, self.generator, self.discriminator, self.g_optim, self.d_optim, self.device,
                                    g_ema_only_ok=g_ema_only_ok)

    def iteration(self, i, real_img):
        """One iteration `i` of the reference loop on a batch of real images already on the device."""
        a = self.args
        # every write to the weights inside this loop goes through FusedAdam / MultiTensorEMA (which move the version
        # counters), so packed weight layouts can be reused between the optimiser steps
        with packed_weights_cache(self._packs):
            self.d_step(real_img)
            if i % a.d_reg_every == 0:
                self.r1_step(real_img)
            self.g_step()
            if i % a.g_reg_every == 0:
                self.path_step()
            if a.spatial_regu and i % a.g_reg_every == 0:
                self.spatial_step()
        self._ema.update(self.accum)                                         # :294
        return D.reduce_loss_dict(self.loss)
