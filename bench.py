"""Benchmark of the TransEditor hot path on MI355X.

    python bench.py --gpus 1 --steps 16 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json's metric: the FFHQ-256 full G+D training iteration at batch 16 per GPU
(configs[2] at N = 1, configs[3] at N > 1; reference train_spatial_query.py:166-294): D step, lazy R1 every 16
iterations, G step, lazy path-length regulariser every 4 (double backward), Adam, EMA.  One "step" = one such
iteration over one synthetic batch already resident in HBM.  The timed window starts at iteration index 0, so the
lazy regularisers fire at i % 16 == 0 and i % 4 == 0 inside the K timed steps (K a multiple of 16 reproduces the
cadence exactly; other K over-represent them, never under-represent).  With N > 1 every rank runs its own batch (weak
scaling, data parallel) and every backward ends in the bucketed gradient all-reduce over RCCL/xGMI.  Rank 0 prints
ONE JSON line.

Objects in the JSON beside the driver contract:
  roofline      the MFMA convolution launches (forward / data-gradient / weight-gradient of every 3x3 kind, G and D) in the timed
                region, timed live with HIP events on the launch stream.  `frac` = executed matrix-pipe work / peak, per pipe:
                executed fp32-MFMA FLOPs / 157.3 T + executed bf16-MFMA FLOPs / 2 516 T (`achieved` / `peak`: the same ratio in
                bf16-pipe-equivalent TFLOP/s); `dominant_kernel` = the split-bf16 Winograd kernel alone against the bf16 peak;
                `achieved_algorithmic` = algorithmic FLOPs / time (round 4's "frac" divided this by the fp32 peak: kept as
                `algorithmic_vs_fp32_peak`); `whole_step_frac` = frac x the share of the wall-clock step those launches take.
                `traffic`, `mfma_util_pct`, `mhz` and `counters` come from rocprofv3 --pmc passes spawned by THIS run after the
                timed region (N = 1; `traffic_source` "live", or "static" = the committed summary if the passes could not run).
  substeps      HIP-event time of each sub-step (D / R1 / G / path) and the cadence-weighted ms per iteration.
  sub_benchmarks  (N = 1) configs[1] generator fwd+bwd at batch 16 and configs[4] FFHQ-1024 generator fwd+bwd at
                batch 4, each with its own value and roofline.
  cpu_baseline  (N = 1) the CPU oracle (oracle/te_oracle.py, a port of the reference's algorithm) running the same
                training iteration on this box's host cores on a bounded sample.
  comm          (N > 1) backend, bytes all-reduced per iteration, isolated per-bucket all-reduce time, the exposed
                communication time (step with exchange - step without) and a bit-identity check of the averaged grads.

`--workload generator` runs only configs[1] (or `--size 1024 --batch 4`: configs[4]) as the headline line.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 (vector == fp32-MFMA), MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2516.0         # dense bf16 MFMA: 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz (same guide)
PEAK_HBM_GBS = 8000.0
METRIC = '256x256 images/sec/GPU, G+D fwd+bwd, batch 16; 1/2/4/8-GPU scaling'


def _arithmetic_note():
    from transeditor_amd.op import modconv
    if modconv.USE_WINOGRAD and modconv.USE_SPLIT_BF16:
        return ('fp32 in, fp32 out, fp32 accumulation everywhere.  The 3x3 stride-1 convolutions with K % 32 == 0, M % 64 == 0 form their '
                'products on the bf16 matrix pipe from a three-piece split of every fp32 operand (x = h + m + l, 24 mantissa bits; six '
                'exact piece products per multiply, the dropped ones below 2^-24): fp32-equivalent - measured deviation from fp64 BELOW '
                "the fp32-MFMA Winograd kernel's at every tested shape (tests/test_gpu_winograd.py, profiles/r04_pytest_gpu.log).  Round 5: "
                'the strided / transposed 3x3 convolutions (csrc/s2s6.hip, t2s6.hip) and the weight gradients of the 3x3 and transposed '
                'kinds (csrc/wgrad6.hip; Co % 64 == 0, Ci % 64 == 0) take the same split form - tests/test_gpu_s2s6.py, test_gpu_t2s6.py, '
                "test_gpu_wgrad6.py hold them to the fp32 kernels' 5e-6 bar against fp64.  "
                'TE_SPLIT_BF16=0 runs the fp32 matrix instructions everywhere (same-box A/B of this line, round 5: 137.3 against 103.8 '
                'img/s, profiles/r05_bench_quick_split_bf16_{on,off}.json).')
    return 'fp32 matrix / vector instructions everywhere (TE_SPLIT_BF16=0)'




ARITHMETIC_SHORT = ('fp32 tensors + fp32 accumulation; 3x3 products on the bf16 MFMA pipe from an exact 3-piece split of each fp32 '
                    'operand (fp32-equivalent, <= fp32-MFMA error vs fp64); TE_SPLIT_BF16=0 = fp32 MFMA everywhere')


def _arithmetic_switches():
    """what `dtype: "f32"` stands on in this run: fp32 tensors and accumulation everywhere; which launches form their products on the
    bf16 pipe (three-piece split, fp32-equivalent) and in which kernel form"""
    from transeditor_amd.op import modconv
    from transeditor_amd import _lib
    return {'split_bf16': bool(modconv.USE_WINOGRAD and modconv.USE_SPLIT_BF16),
            'split_bf16_strided': bool(modconv.USE_SPLIT_BF16 and modconv.USE_SPLIT_S2), 'split_bf16_transposed': bool(modconv.USE_SPLIT_BF16 and modconv.USE_SPLIT_T2),
            'split_bf16_weight_gradient_3x3': bool(_lib.wgrad_split()), 'split_bf16_1x1': bool(modconv.USE_SPLIT_BF16 and modconv.USE_SPLIT_1X1),
            'arithmetic_short': ARITHMETIC_SHORT if modconv.USE_SPLIT_BF16 else 'fp32 MFMA / vector instructions everywhere',
            'split_bf16_kernel_form': W6_KERNEL.get(_w6_form(), None)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 16; 4 at --size 1024)')
    ap.add_argument('--workload', choices=['train', 'generator', 'sample'], default='train',
                    help='train = BASELINE configs[2]/[3], the full G+D iteration (default, = the metric); '
                         'generator = configs[1] / configs[4] generator fwd+bwd only; sample = the inference-only pipeline '
                         '(frozen g_ema sampling loop, test_spatial_query.py:20-31)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-sub', action='store_true', help='skip the sub-benchmarks (configs[1], configs[4])')
    ap.add_argument('--no-pmc', action='store_true', help='skip the live rocprofv3 --pmc passes (roofline.traffic falls back to '
                                                          'the committed summary, labelled static)')
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of individual te_conv / te_wgrad launches (they run on torch's current stream, so
    torch.cuda.Event brackets exactly the kernel)."""

    def __init__(self):
        self.records = []      # (class, flops, start, end)
        self.enabled = False
        self.installed = False

    def install(self):
        if self.installed:
            return
        self.installed = True
        from transeditor_amd import _lib
        timer = self
        orig_conv, orig_wgrad = _lib.conv, _lib.wgrad_slabs
        names = {_lib.CONV_3X3: 'conv3x3', _lib.CONV_T2: 'convT2', _lib.CONV_S2: 'convS2', _lib.CONV_1X1: 'conv1x1',
                 _lib.CONV_3X3W: 'conv3x3', _lib.CONV_3X3W6: 'conv3x3',         # (Winograd forms of the same convolution: same algorithmic FLOPs)
                 _lib.CONV_S2S6: 'convS2', _lib.CONV_T2S6: 'convT2',            # (the strided / transposed kinds on the bf16 pipe)
                 _lib.CONV_1X1S6: 'conv1x1'}                                   # (round 6: the plain 1x1 product on the bf16 pipe)

        def conv(x, wp, kind, M, H, W, *a, **k):
            if not timer.enabled:
                return orig_conv(x, wp, kind, M, H, W, *a, **k)
            B, K = x.shape[0], x.shape[1]
            taps = 1 if kind in (_lib.CONV_1X1, _lib.CONV_1X1S6) else 9
            flops = 2.0 * taps * K * M * H * W * B       # algorithmic: T2/S2 counted on the low-res grid
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_conv(x, wp, kind, M, H, W, *a, **k)
            e.record()
            # the Winograd form EXECUTES 12/18 of the direct form's multiply-adds for the same (algorithmic) convolution; the split form
            # (TE_CONV_3X3W6) executes them as SIX bf16 piece products each on the bf16 matrix pipe and none on the fp32 one
            # (TE_CONV_S2S6: no Winograd form, six piece products per multiply-add = 6x the algorithmic FLOPs on the bf16 pipe)
            ex32 = flops * (2.0 / 3.0 if kind == _lib.CONV_3X3W else (0.0 if kind in (_lib.CONV_3X3W6, _lib.CONV_S2S6, _lib.CONV_T2S6, _lib.CONV_1X1S6) else 1.0))
            ex16 = flops * (4.0 if kind == _lib.CONV_3X3W6 else (6.0 if kind in (_lib.CONV_S2S6, _lib.CONV_T2S6, _lib.CONV_1X1S6) else 0.0))
            timer.records.append((names[kind], flops, s, e, ex32, ex16))
            if kind == _lib.CONV_3X3W6:       # the dominant kernel on its own (a VIEW of the conv3x3 class, never summed with it)
                timer.records.append(('conv3x3_split_bf16', flops, s, e, ex32, ex16))
            if kind == _lib.CONV_S2S6:
                timer.records.append(('convS2_split_bf16', flops, s, e, ex32, ex16))
            if kind == _lib.CONV_T2S6:
                timer.records.append(('convT2_split_bf16', flops, s, e, ex32, ex16))
            if kind == _lib.CONV_1X1S6:
                timer.records.append(('conv1x1_split_bf16', flops, s, e, ex32, ex16))
            return out

        def wgrad(g, x, kind, H, W, *a, **k):
            if not timer.enabled:
                return orig_wgrad(g, x, kind, H, W, *a, **k)
            taps = 1 if kind == _lib.CONV_1X1 else 9
            flops = 2.0 * taps * g.shape[1] * x.shape[1] * H * W * g.shape[0]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_wgrad(g, x, kind, H, W, *a, **k)
            e.record()
            # (the pair form of the 3x3 weight gradient: 12/18 of the direct form's multiply-adds, as above; csrc/wgrad6.hip runs the
            #  pair form as six bf16 piece products per multiply-add: 4x the algorithmic FLOPs on the bf16 pipe, none on the fp32 one)
            split = bool(_lib.wgrad_split() and _lib.wgrad_split_ok(kind, g.shape[1], x.shape[1], H, W))
            ex32 = 0.0 if split else flops * (2.0 / 3.0 if _lib.wgrad_pair_form(kind, g.shape[1], x.shape[1], H, W) else 1.0)
            ex16 = (4.0 if kind == _lib.CONV_3X3 else 6.0) * flops if split else 0.0      # (transposed kind: no pair form, 6 products)
            timer.records.append(('wgrad_' + names[kind], flops, s, e, ex32, ex16))
            if split:
                timer.records.append(('wgrad_' + names[kind] + '_split_bf16', flops, s, e, ex32, ex16))
            return out

        _lib.conv, _lib.wgrad_slabs = conv, wgrad

    def reset(self):
        self.records = []

    def summary(self):
        agg = {}
        for name, flops, s, e, executed, executed16 in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += s.elapsed_time(e) * 1e-3
            a[3] += executed
            a[4] += executed16
        return {k: {'launches': v[0], 'avg_ms': 1e3 * v[2] / v[0], 'tflops': v[1] / v[2] / 1e12, 'total_ms': 1e3 * v[2],
                    'gflop': v[1] / 1e9, 'executed_tflops': v[3] / v[2] / 1e12, 'executed_bf16_tflops': v[4] / v[2] / 1e12}
                for k, v in agg.items()}

    def roofline(self, wall_s, steps):
        """roofline object over the MFMA convolution launches recorded since reset().

        `frac` = share of the matrix pipes' time the ISSUED instructions account for at peak rate: executed fp32-MFMA FLOPs / 157.3 T
        + executed bf16-MFMA FLOPs / 2 516 T (a Winograd launch executes 2/3 of its algorithmic multiply-adds, a split-bf16 launch six
        bf16 piece products per multiply-add = 4x its algorithmic FLOPs on the bf16 pipe).  `achieved` / `peak` express the same ratio
        in one unit, bf16-pipe-equivalent TFLOP/s (an fp32 MFMA FLOP occupies the pipe 16x as long as a bf16 one), so frac ==
        achieved / peak.  The algorithmic rate (2 * 9 * K * M * H * W * B per launch / time) is `achieved_algorithmic`; round 4 divided
        THAT by the fp32 peak and called it frac (1.098) - kept as `algorithmic_vs_fp32_peak`, it is not a fraction of any roofline."""
        ks = self.summary()
        conv_keys = [k for k in ks if not k.endswith('1x1') and not k.endswith('_split_bf16')]   # (1x1: not 3x3 GEMMs; *_split_bf16: a view)
        gflop = sum(ks[k]['gflop'] for k in conv_keys)
        tms = sum(ks[k]['total_ms'] for k in conv_keys)
        ach = gflop / tms if tms else 0.0                            # GFLOP / ms == TFLOP/s
        executed = sum(ks[k]['executed_tflops'] * ks[k]['total_ms'] for k in conv_keys) / tms if tms else 0.0
        executed16 = sum(ks[k]['executed_bf16_tflops'] * ks[k]['total_ms'] for k in conv_keys) / tms if tms else 0.0
        ex_frac = executed / PEAK_FP32_TFLOPS + executed16 / PEAK_BF16_TFLOPS
        share = tms * 1e-3 / wall_s if wall_s else None
        traffic, note = _pmc_traffic()                               # (static; attach_counters() replaces it with this run's counters)
        dom = ks.get('conv3x3_split_bf16')
        dominant = None
        if dom:
            dominant = {'kernel': W6_KERNEL.get(_w6_form(), 'wino6_kernel'), 'pipe': 'bf16 MFMA (v_mfma_f32_32x32x16_bf16)',
                        'launches': dom['launches'], 'ms_per_step': dom['total_ms'] / steps,
                        'share_of_step': dom['total_ms'] * 1e-3 / wall_s if wall_s else None,
                        'achieved_algorithmic': dom['tflops'], 'achieved': dom['executed_bf16_tflops'], 'peak': PEAK_BF16_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': dom['executed_bf16_tflops'] / PEAK_BF16_TFLOPS,
                        'executed_factor': dom['executed_bf16_tflops'] / dom['tflops'] if dom['tflops'] else None}
        return {'bound': 'mfma', 'achieved': ex_frac * PEAK_BF16_TFLOPS, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ex_frac,
                'achieved_unit_note': 'bf16-pipe-equivalent executed TFLOP/s = executed_bf16_tflops + executed_tflops x (2516 / 157.3)',
                'peaks': {'bf16_mfma_dense': PEAK_BF16_TFLOPS, 'fp32_mfma': PEAK_FP32_TFLOPS},
                'traffic': traffic, 'traffic_note': note, 'traffic_source': 'static',
                'kernel': W6_KERNEL.get(_w6_form(), 'wino6_kernel') +
                          ' / s2s6q_kernel / t2s6q_kernel / wgrad6_kernel / wgrad6tw_kernel (v_mfma_f32_32x32x16_bf16, three-piece split) / '
                          'wino3x3_kernel / conv_mfma_kernel / wgrad_mfma_kernel (fp32 v_mfma_f32_32x32x2) where the split kernels do '
                          'not apply; all 3x3 kinds',
                'dominant_kernel': dominant,
                'executed_tflops': executed, 'executed_bf16_tflops': executed16, 'executed_frac': ex_frac,
                'achieved_algorithmic': ach, 'algorithmic_vs_fp32_peak': ach / PEAK_FP32_TFLOPS,
                'achieved_note': 'achieved_algorithmic = ALGORITHMIC FLOPs (2 * 9 * K * M * H * W * B per launch) / measured time.  The 3x3 '
                                 'stride-1 launches from 32x32 up run the 1-D Winograd F(2,3) form (csrc/wino.hip), which executes 2/3 of '
                                 'those multiply-adds (and so does the pair form, F(3,2), of the 3x3 weight gradient on the 8-wave tile); '
                                 'the 3x3 stride-1 launches with K % 32 == 0, M % 64 == 0 run the same Winograd form on the bf16 matrix '
                                 'pipe (csrc/wino6.hip: every fp32 operand split into three bf16 pieces, six exact piece products '
                                 'accumulated in fp32 - fp32-equivalent results, deviation from fp64 not larger than the fp32 MFMA '
                                 "chain's, tests/test_gpu_winograd.py): executed_bf16_tflops = 4 x their algorithmic FLOPs, priced "
                                 'against the 2516 TFLOP/s dense bf16 peak (the strided / transposed kinds and the transposed weight '
                                 'gradient, which have no Winograd form: 6 x; the pair-form 3x3 weight gradient of csrc/wgrad6.hip: 4 x).  '
                                 'frac prices the MFMA work actually ISSUED on the pipe it was issued on - moving a launch from the fp32 '
                                 'pipe (peak 157.3) to the bf16 pipe (peak 2516 for 4 - 6 x the FLOPs) makes it faster AND lowers frac: '
                                 'under dense bf16 MFMA work the part is power-limited at ~1 000 - 1 150 TFLOP/s executed (clock 1.45 - '
                                 '1.75 GHz), profiles/experiments/r05_w6p_phase_profile.log',
                'kernel_time_share': share,
                'algorithmic_gflop_per_step': gflop / steps,
                'whole_step_tflops': gflop / 1e3 / wall_s if wall_s else None,
                # the same executed work priced against the WHOLE wall-clock step (every other kernel and gap charged to it)
                'whole_step_frac': ex_frac * share if share else None,
                'whole_step_algorithmic_vs_fp32_peak': gflop / 1e3 / wall_s / PEAK_FP32_TFLOPS if wall_s else None,
                'per_kernel': ks}


W6_KERNEL = {2: 'wino6q_kernel (wino6p_kernel where M % 128 != 0 or W == 16)', 1: 'wino6p_kernel', 0: 'wino6_kernel'}


def _w6_form():
    try:
        from transeditor_amd import _lib
        return _lib.wino6_form(-1)
    except Exception:
        return None


PMC_PASSES = (('mfma', 'SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32'),
              ('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE'))
PMC_KERNELS = {'conv3x3_fwd_128to128_at256_b16': 'wino6', 'conv3x3_fp32_winograd_kernel_same_shape': 'wino3x3_kernel',
               'conv3x3_direct_kernel_same_shape': 'conv_mfma_kernel<0',
               'wgrad3x3_128x128_at256_b16': 'wgrad6_kernel', 'wgrad3x3_fp32_kernel_same_shape': 'wgrad_mfma_kernel<0',
               'convT2_256to128_at128_b16': 't2s6', 'convS2_128to256_at128_b16': 's2s6',
               'convT2_fp32_kernel_same_shape': 'conv_mfma_kernel<1, 0, true, false, 2, true', 'convS2_fp32_kernel_same_shape': 'conv_mfma_kernel<2',
               'convT2_last_row_and_column': 't2_edge_kernel',
               'wgradT2_256x128_at128_b16': 'wgrad6t', 'wgradT2_fp32_kernel_same_shape': 'wgrad_mfma_kernel<1'}


def live_counters(timeout_s=150):
    """Hardware counters collected in THIS run (outside the timed region): three `rocprofv3 --kernel-trace --pmc` passes (MFMA
    busy cycles, FETCH_SIZE, WRITE_SIZE: separate passes, counters only, as MI355X_MICROARCH.md prescribes) over
    tools/kernel_once.py, which launches every hot kernel at its FFHQ-256 / batch-16 top shape; parsed by
    tools/pmc_summary.py (FETCH_SIZE doubled: the gfx950 correction).  Returns (per-kernel dict | None, note)."""
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return None, 'rocprofv3 not found'
    out = tempfile.mkdtemp(prefix='te_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp', REP='2')
    t0 = time.perf_counter()
    try:
        for name, counters in PMC_PASSES:
            cmd = [prof, '--kernel-trace', '--pmc', *counters.split(), '-d', out, '-o', name, '--output-format', 'csv', '--',
                   sys.executable, os.path.join(ROOT, 'tools', 'kernel_once.py')]
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0 or not os.path.exists(os.path.join(out, name + '_counter_collection.csv')):
                return None, f'rocprofv3 pass {name} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}'
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import pmc_summary
        summ = pmc_summary.summarise(out, need_lds=False)
    except Exception as e:                                          # a missing counter, a timeout: the static file is the fallback
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(out, ignore_errors=True)
    return summ, f'{len(PMC_PASSES)} rocprofv3 --pmc passes in {time.perf_counter() - t0:.0f} s'


def attach_counters(roof, live=True):
    """fill roofline.traffic / mfma_util_pct / mhz (+ `counters` for the top shapes of every MFMA kernel family) from counters
    collected in this run; the committed PMC summary only as a labelled fallback"""
    summ, note = live_counters() if live else (None, 'skipped (--no-pmc)')
    if summ is None:
        roof['traffic'], roof['traffic_note'] = _pmc_traffic()
        roof['traffic_source'] = 'static'
        roof['live_counters_error'] = note
        return roof
    ctr = {}
    for label, prefix in PMC_KERNELS.items():
        v = next((val for k, val in summ.items() if k.startswith(prefix)), None)
        if v is None:
            continue
        hbm = (v['hbm_read_mb'] + v['hbm_write_mb']) * 1e6
        ctr[label] = {'kernel': next(k for k in summ if k.startswith(prefix)), 'us_under_counters': v['us'], 'mhz': v['mhz'],
                      'mfma_util_pct': v['mfma_util_pct'], 'tflops_under_counters': v['tflops'],
                      'hbm_read_bytes': v['hbm_read_mb'] * 1e6, 'hbm_write_bytes': v['hbm_write_mb'] * 1e6,
                      'algorithmic_bytes': v['algorithmic_mb'] * 1e6 if v['algorithmic_mb'] else None,
                      'traffic_over_algorithmic': hbm / (v['algorithmic_mb'] * 1e6) if v['algorithmic_mb'] else None,
                      'hbm_GBps': v['gbs']}
    hbm_tail = {}
    for k, v in summ.items():
        if any(t in k for t in ('blur44', 'fir_tile', 'bias_act', 'rgb_', 'wgrad_reduce')) and v['algorithmic_mb']:
            hbm_tail[k[:60]] = {'us': v['us'], 'algorithmic_TBps': v['algorithmic_mb'] / v['us'],        # MB / us == TB/s
                                'frac_of_hbm_peak': v['algorithmic_mb'] / v['us'] / (PEAK_HBM_GBS / 1e3),
                                'traffic_over_algorithmic': (v['hbm_read_mb'] + v['hbm_write_mb']) / v['algorithmic_mb']}
    top = ctr.get('conv3x3_fwd_128to128_at256_b16') or ctr.get('conv3x3_direct_kernel_same_shape')
    if top is None:
        roof['traffic'], roof['traffic_note'] = _pmc_traffic()
        roof['traffic_source'] = 'static'
        roof['live_counters_error'] = 'top-shape kernel missing from the counter CSVs'
        return roof
    roof['traffic'] = top['hbm_read_bytes'] + top['hbm_write_bytes']
    roof['traffic_source'] = 'live'
    roof['mfma_util_pct'], roof['mhz'] = top['mfma_util_pct'], top['mhz']
    roof['traffic_note'] = (f'LIVE: {note} over tools/kernel_once.py inside this bench run (outside the timed region); per launch of '
                            f'the dominant kernel at its top shape (3x3 128->128 @256x256, batch 16): FETCH_SIZE x2 (gfx950 '
                            f'correction) + WRITE_SIZE vs {(top["algorithmic_bytes"] or 0) / 1e6:.0f} MB algorithmic')
    roof['counters'] = ctr
    roof['hbm_bound_kernels'] = hbm_tail
    return roof


def _pmc_traffic():
    """FALLBACK (labelled static): HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (tools/pmc_round.sh + tools/pmc_summary.py), used only when live_counters() cannot run.  Latest round's summary wins."""
    try:
        prof = os.path.join(ROOT, 'profiles')
        cands = sorted(f for f in os.listdir(prof) if f.endswith('_pmc_summary.json'))
        pm = json.load(open(os.path.join(prof, cands[-1])))
        k3 = next(v for k, v in pm.items() if k.startswith('conv_mfma_kernel<0'))
        traffic = (k3['hbm_read_mb'] + k3['hbm_write_mb']) * 1e6
        note = (f"static, from profiles/{cands[-1]} (PMC pass of the 128->128 @256x256 batch-16 3x3 launch): FETCH_SIZE x2 "
                f"(gfx950 correction) = {k3['hbm_read_mb']:.0f} MB read + WRITE_SIZE {k3['hbm_write_mb']:.0f} MB written vs "
                f"{k3['algorithmic_mb']:.0f} MB algorithmic; MFMA utilisation {k3['mfma_util_pct']:.1f} % at {k3['mhz']:.0f} MHz")
        return traffic, note
    except Exception:
        return None, None


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_model():
    try:
        return next(l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name'))
    except Exception:
        return 'unknown'


def _oracle_params(module_sd):
    P = {}
    for k, v in module_sd.items():
        v = v.detach().cpu()
        trainable = (v.is_floating_point() and 'noises' not in k and 'kernel' not in k and not k.startswith('token'))
        P[k] = v.clone().requires_grad_(True) if trainable else v
    return P, [v for v in P.values() if v.requires_grad]


def _pick_threads(O, Pg, size):
    """Thread count for the CPU legs: the fastest of a short sweep on one batch-2 generator forward (all hardware threads
    oversubscribe the small CPU convolutions badly: 187 s/iter measured in round 1 with 256)."""
    from transeditor_amd import synth
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cands = sorted({t for t in (16, 32, 64, 128) if t <= ncpu} | {min(ncpu, 32)})
    z, p = synth.latents(2, 77)
    best, sweep = None, {}
    for t in cands:
        torch.set_num_threads(t)
        with torch.no_grad():
            O.generator_forward(Pg, z[:1], p[:1], size)                 # touch
            t0 = time.perf_counter()
            O.generator_forward(Pg, z, p, size)
            dt = time.perf_counter() - t0
        sweep[t] = round(dt, 3)
        if best is None or dt < sweep[best]:
            best = t
    torch.set_num_threads(best)
    return best, sweep, ncpu


def cpu_baseline_train(size, config_batch=16):
    """The reference's training iteration (train_spatial_query.py:166-294) through the CPU oracle on a bounded sample.
    The two steps that carry the iteration - D and G - run ONCE AT THE CONFIG BATCH (16; D sees 16 real + 16 fake images), after
    one untimed warm-up pass at batch 2; the two lazy regularisers run at a reduced batch (R1: 2 images, path length: 1 image) and
    are scaled per image to their config batches (16 and 8) - they weigh 1/16 and 1/4 of an iteration.  Adam included.  The iteration
    is composed with the cadence exactly as the GPU line is:  t = t_D + t_G + t_R1 / 16 + t_path / 4.  If the batch-2 pass predicts
    more than ~4 minutes for the config-batch pair, the pair runs at batch 8 instead and `sample` says so."""
    from oracle import te_oracle as O
    from transeditor_amd import synth
    from transeditor_amd.model_spatial_query import Discriminator, Generator
    token = 2 * (int(math.log2(size)) - 1)
    torch.manual_seed(5)
    Pg, g_leaves = _oracle_params(Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1).state_dict())
    Pd, d_leaves = _oracle_params(Discriminator(size).state_dict())
    threads, sweep, ncpu = _pick_threads(O, Pg, size)
    g_opt = torch.optim.Adam(g_leaves, lr=0.002 * 0.8, betas=(0.0, 0.99 ** 0.8))
    d_opt = torch.optim.Adam(d_leaves, lr=0.002 * 16 / 17, betas=(0.0, 0.99 ** (16 / 17)))
    t = {}

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        fn(*a)
        t[name] = time.perf_counter() - t0

    def d_step(B):
        real = torch.randn(B, 3, size, size).clamp(-1, 1)
        z, p = synth.latents(B, 900)
        with torch.no_grad():
            fake = O.generator_forward(Pg, z, p, size)[0]
        loss = O.d_logistic_loss(O.discriminator_forward(Pd, real, size), O.discriminator_forward(Pd, fake, size))
        d_opt.zero_grad()
        for v, g in zip(d_leaves, torch.autograd.grad(loss, d_leaves)):
            v.grad = g
        d_opt.step()

    def r1_step(B):
        r = torch.randn(B, 3, size, size).clamp(-1, 1).requires_grad_(True)
        pred = O.discriminator_forward(Pd, r, size)
        loss = 10.0 / 2 * O.d_r1_loss(pred, r) * 16 + 0 * pred[0]
        d_opt.zero_grad()
        for v, g in zip(d_leaves, torch.autograd.grad(loss.sum(), d_leaves, allow_unused=True)):
            v.grad = g
        d_opt.step()

    def g_step(B):
        z, p = synth.latents(B, 901)
        fake = O.generator_forward(Pg, z, p, size)[0]
        for v in d_leaves:
            v.requires_grad_(False)
        loss = O.g_nonsaturating_loss(O.discriminator_forward(Pd, fake, size))
        g_opt.zero_grad()
        for v, g in zip(g_leaves, torch.autograd.grad(loss, g_leaves, allow_unused=True)):
            v.grad = g
        for v in d_leaves:
            v.requires_grad_(True)
        g_opt.step()

    def path_step(n):
        z, p = synth.latents(n, 902)
        img, lat, _ = O.generator_forward(Pg, z, p, size)
        noise = torch.randn_like(img) / math.sqrt(size * size)
        pen, _, _ = O.g_path_regularize(img, lat, 0.0, noise)
        g_opt.zero_grad()
        for v, g in zip(g_leaves, torch.autograd.grad(2.0 * 4 * pen + 0 * img[0, 0, 0, 0], g_leaves, allow_unused=True)):
            v.grad = g
        g_opt.step()

    # warm-up / calibration pass at batch 2 (first-touch cost: oneDNN primitive creation, thread pool, page faults), timed so that
    # the config-batch pair can be bounded
    timed('d2', d_step, 2)
    timed('g2', g_step, 2)
    timed('d2', d_step, 2)
    timed('g2', g_step, 2)
    BC = config_batch if (t['d2'] + t['g2']) * config_batch / 2 < 240 else config_batch // 2
    timed('d', d_step, BC)
    timed('g', g_step, BC)
    timed('r1', r1_step, 2)
    timed('path', path_step, 1)
    r1_scaled = t['r1'] * BC / 2                  # R1 runs on the whole batch
    path_scaled = t['path'] * (BC // 2) / 1       # the path-length step on batch / path_batch_shrink (= 2)
    it = t['d'] + t['g'] + r1_scaled / 16 + path_scaled / 4
    # per-image cost vs batch: generator forward at batch 2 and batch 16
    with torch.no_grad():
        z, p = synth.latents(16, 903)
        t0 = time.perf_counter()
        O.generator_forward(Pg, z[:2], p[:2], size)
        f2 = (time.perf_counter() - t0) / 2
        t0 = time.perf_counter()
        O.generator_forward(Pg, z, p, size)
        f16 = (time.perf_counter() - t0) / 16
    # BASELINE configs[1] on the host at the CONFIG batch: generator forward + backward, batch 16, once
    z, p = synth.latents(16, 904)
    t0 = time.perf_counter()
    img = O.generator_forward(Pg, z, p, size)[0]
    torch.autograd.grad(img.sum(), g_leaves, allow_unused=True)
    fb16 = time.perf_counter() - t0
    del img
    return {'value': BC / it, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'CPU oracle (PyTorch fp32 restatement of the reference) running ONE FFHQ-{size} G+D training iteration: the D step '
                      f'({t["d"]:.1f} s) and the G step ({t["g"]:.1f} s) at batch {BC}' +
                      (' = THE CONFIG BATCH' if BC == config_batch else f' (config batch {config_batch}: predicted too long)') +
                      f', once each, after a warm-up pass at batch 2 (D {t["d2"]:.1f} s, G {t["g2"]:.1f} s); the lazy regularisers at a reduced '
                      f'batch, scaled per image: R1 on 2 images {t["r1"]:.1f} s -> x{BC // 2}, path length on 1 image {t["path"]:.1f} s -> '
                      f'x{BC // 2}; Adam included; iteration = D + G + R1/16 + path/4 = {it:.1f} s',
            'sample_short': f'oracle: one FFHQ-{size} G+D iteration, D+G once at batch {BC} after a batch-2 warm-up; R1 (2 img) and '
                            f'path (1 img) scaled per image; {it:.0f} s/iteration',
            'batch': BC, 'seconds': {k: round(v, 3) for k, v in t.items()},
            'batch_scaling': {'generator_fwd_s_per_image_batch2': f2, 'generator_fwd_s_per_image_batch16': f16,
                              'iteration_images_per_sec_from_the_batch2_pass': 2 / (t['d2'] + t['g2'] + t['r1'] / 16 + t['path'] / 4),
                              'note': 'per-image cost at the config batch relative to a batch-2 sample'},
            'generator_fwd_bwd_batch16': {'seconds': fb16, 'images_per_sec': 16 / fb16,
                                          'note': 'BASELINE configs[1] (generator fwd+bwd at the config batch 16) on the same cores, one pass'},
            'thread_sweep_s_generator_fwd_batch2': sweep, 'usable_threads': ncpu,
            'cpu_model': _cpu_model(), 'host_threads': os.cpu_count()}


def cpu_baseline_generator(size):
    """Oracle generator fwd+bwd on the host cores; bounded sample (batch 2, 1 warm-up + 4 timed iterations)."""
    from oracle import te_oracle as O
    from transeditor_amd import synth
    from transeditor_amd.model_spatial_query import Generator
    token = 2 * (int(math.log2(size)) - 1)
    P, leaves = _oracle_params(Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1).state_dict())
    threads, sweep, ncpu = _pick_threads(O, P, size)
    B, iters = 2, 4
    times = []
    for it in range(iters + 1):
        z, p = synth.latents(B, 900 + it)
        t0 = time.perf_counter()
        img, _, _ = O.generator_forward(P, z, p, size)
        torch.autograd.grad(img.sum(), leaves, allow_unused=True)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / iters
    return {'value': B / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'CPU oracle (PyTorch fp32 restatement of the reference), generator fwd+bwd {size}x{size}, '
                      f'batch {B}, 1 warm-up + {iters} timed iterations, {dt:.2f} s/iter',
            'thread_sweep_s_generator_fwd_batch2': sweep, 'usable_threads': ncpu,
            'cpu_model': _cpu_model(), 'host_threads': os.cpu_count()}


# ------------------------------------------------------------------------------------------------ GPU legs
def fence(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def max_over_ranks(elapsed, world, dev):
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def run_generator(size, B, steps, warmup, dev, world, rank, timer, G=None):
    """configs[1] / configs[4]: generator forward + backward; returns (elapsed_s, roofline | None)."""
    from transeditor_amd.model_spatial_query import Generator
    from transeditor_amd.utils import distributed as D
    token = 2 * (int(math.log2(size)) - 1)
    if G is None:
        torch.manual_seed(1234)                                    # same init on every rank (reference-style randn init)
        G = Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1).to(dev)
        D.broadcast_module(G)
    for p in G.parameters():
        p.requires_grad_(True)
    sync = D.GradSync(G)
    params = list(G.parameters())
    torch.manual_seed(1000 + rank)
    n_in = steps + warmup
    zs = [torch.randn(B, 512, 16, device=dev) for _ in range(n_in)]   # resident in HBM before the timed region
    ps = [torch.randn(B, 512, 16, device=dev) for _ in range(n_in)]
    wimg = torch.randn(B, 3, size, size, device=dev)

    def step(i):
        for p in params:
            p.grad = None
        img = G(zs[i], ps[i])[0]
        (img * wimg).sum().backward()
        sync.all_reduce()                                          # no-op at world == 1

    for i in range(warmup):
        step(i)
    fence(world)
    timer.reset()
    timer.enabled = timer.installed
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    fence(world)
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    elapsed = max_over_ranks(elapsed, world, dev)
    roof = timer.roofline(elapsed, steps) if timer.installed else None
    sync.remove_hooks()
    for p in params:
        p.grad = None
    return elapsed, roof


def run_sampling(args, size, B, dev, base):
    """(f.3) the reference's sample_generation loop (test_spatial_query.py:20-31): a frozen generator under no_grad,
    fixed batch, fresh latents every iteration.  Three ways: the training-path forward, the frozen-weight cache, and the
    cache + hipGraph replay (inference.GeneratorSampler)."""
    from transeditor_amd.inference import GeneratorSampler
    from transeditor_amd.model_spatial_query import Generator
    token = 2 * (int(math.log2(size)) - 1)
    torch.manual_seed(1234)
    G = Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1).to(dev).eval()
    n = args.steps + args.warmup
    zs = [torch.randn(B, 512, 16, device=dev) * 0.7 for _ in range(n)]
    p = torch.randn(B, 512, 16, device=dev) * 0.7

    def timed(fn):
        for i in range(args.warmup):
            fn(zs[i], p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            fn(zs[args.warmup + i], p)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def plain(z, q):
        with torch.no_grad():
            return G(z, q)
    t_plain = timed(plain)
    t_cache = timed(GeneratorSampler(G, use_graph=False))
    t_graph = timed(GeneratorSampler(G, use_graph=True))
    out = dict(base, metric=f'{size}x{size} images/sec/GPU, frozen generator sampling loop (inference-only pipeline)',
               value=B * args.steps / t_graph, ms_per_step=1e3 * t_graph / args.steps,
               config={'workload': f'FFHQ-{size} g_ema sampling loop (test_spatial_query.py:20-31), batch {B}, no_grad, '
                                   f'{args.steps} batches = {B * args.steps} samples', 'global_batch': B, 'parallelism': 'dp1'},
               sampling={'training_path_forward_img_s': B * args.steps / t_plain,
                         'frozen_weight_cache_img_s': B * args.steps / t_cache,
                         'cache_plus_hipgraph_img_s': B * args.steps / t_graph,
                         'ms_per_batch': {'training_path': 1e3 * t_plain / args.steps, 'cache': 1e3 * t_cache / args.steps,
                                          'graph': 1e3 * t_graph / args.steps}})
    out['detail'] = write_detail(out)
    line = json.loads(compact_line(out))
    line['sampling'] = {k: _num(v) for k, v in out['sampling'].items() if not isinstance(v, dict)}
    print(json.dumps(line, allow_nan=False, separators=(',', ':')), flush=True)


class SubstepClock:
    """HIP events around the four sub-steps of TrainStep.iteration (wraps the bound methods)."""

    def __init__(self, ts):
        self.rec = {k: [] for k in ('d', 'r1', 'g', 'path')}
        self.enabled = False
        for key, name in (('d', 'd_step'), ('r1', 'r1_step'), ('g', 'g_step'), ('path', 'path_step')):
            setattr(ts, name, self._wrap(key, getattr(ts, name)))

    def _wrap(self, key, fn):
        def run(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **k)
            e.record()
            self.rec[key].append((s, e))
            return out
        return run

    def summary(self, args):
        ms = {k: (sum(s.elapsed_time(e) for s, e in v) / len(v) if v else None) for k, v in self.rec.items()}
        out = {f'{k}_ms': v for k, v in ms.items()}
        out['calls'] = {k: len(v) for k, v in self.rec.items()}
        if all(v is not None for v in ms.values()):
            out['cadence_weighted_ms_per_iteration'] = (ms['d'] + ms['g'] + ms['r1'] / args.d_reg_every
                                                        + ms['path'] / args.g_reg_every)
            out['note'] = ('GPU time of each sub-step incl. its optimiser step (EMA excluded); cadence-weighted = '
                           'D + G + R1/16 + path/4, the long-run mean iteration (K = multiple of 16)')
        return out


def grads_checksum(module, world, dev):
    """Every rank must hold bit-identical averaged gradients after GradSync.all_reduce: compare an integer checksum of
    the raw gradient bits across ranks (outside the timed region)."""
    import torch.distributed as dist
    acc = torch.zeros(1, dtype=torch.int64, device=dev)
    n = 0
    for p in module.parameters():
        if p.grad is not None:
            acc += p.grad.detach().contiguous().view(torch.int32).to(torch.int64).sum()
            n += p.grad.numel()
    allv = [torch.zeros_like(acc) for _ in range(world)]
    dist.all_gather(allv, acc)
    vals = [int(v.item()) for v in allv]
    return {'identical_across_ranks': len(set(vals)) == 1, 'checksum_int64': vals[0], 'elements': n}


def isolated_allreduce(sync, dev, reps=5):
    """event-timed all-reduce of each bucket size on an otherwise idle GPU (outside the timed region)"""
    import torch.distributed as dist
    out = []
    for bucket in sync.buckets:
        n = sum((p.numel() + 3) & ~3 for p in bucket)
        buf = torch.zeros(n, device=dev)
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            dist.all_reduce(buf)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        w = dist.get_world_size()
        out.append({'bytes': 4 * n, 'ms': ms, 'bus_GBps': 4 * n * 2 * (w - 1) / w / ms / 1e6})
    return out


class Watchdog:
    """A communication step that hangs (rendezvous, the first RCCL collective: ring set-up over xGMI, IPC handles) must not eat
    the driver's whole time limit without a trace: if `arm(what, seconds)` is not followed by `disarm()` in time, every rank
    says so on stderr, rank 0 prints ONE JSON line with `comm.error` (so the failure can be judged from the record), and the
    process exits with code 3.  The timer runs on its own thread (blocking HIP / RCCL calls release the GIL)."""

    def __init__(self, base, rank, info):
        self.base, self.rank, self.info, self.timer = base, rank, info, None

    def arm(self, what, seconds):
        import threading
        self.disarm()
        self.timer = threading.Timer(seconds, self._fire, args=(what, seconds))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self, what, seconds):
        msg = f'{what} did not complete within {seconds:.0f} s on rank {self.rank}'
        print(f'bench.py watchdog: {msg}', file=sys.stderr, flush=True)
        if self.rank == 0:
            print(compact_line(dict(self.base, value=None, ms_per_step=None, comm=dict(self.info, error=msg))), flush=True)
        os._exit(3)


def comm_preflight(args, backend, world, rank, local_rank, dev):
    """init_process_group + ONE tiny all-reduce under a watchdog, before any model is built; returns what identifies the
    communication set-up (library version, device of every rank) for the `comm` object."""
    import torch.distributed as dist
    info = {'backend': backend + (' (RCCL)' if backend == 'nccl' else ''), 'world_size': world,
            'env': {k: os.environ.get(k) for k in ('HSA_ENABLE_IPC_MODE_LEGACY', 'NCCL_DEBUG', 'NCCL_SOCKET_IFNAME', 'MASTER_ADDR',
                                                    'MASTER_PORT', 'HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'OMP_NUM_THREADS')}}
    try:
        info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:
        info['rccl_version'] = f'unavailable ({type(e).__name__})'
    base = {'metric': METRIC, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic'}
    if backend == 'nccl':
        # With NCCL_DEBUG=VERSION (exported on the GPU boxes) RCCL printf()s a five-line banner to STDOUT from every rank - measured
        # in the round-4 rehearsal (profiles/r04_bench_rccl_world1_before_fix.txt): it came out AFTER the JSON line, at exit, and
        # NCCL_DEBUG_FILE does not redirect it.  The driver wants ONE JSON line on stdout, so the banner is switched off here (the
        # original setting is recorded in `comm.env`; the library version comes from torch.cuda.nccl.version()).
        info['env']['NCCL_DEBUG_as_found'] = os.environ.get('NCCL_DEBUG')
        os.environ['NCCL_DEBUG'] = os.environ.get('TE_BENCH_NCCL_DEBUG', 'WARN')
        os.environ.setdefault('NCCL_DEBUG_FILE', f'/tmp/te_rccl_debug.{os.getpid()}.log')      # (warnings, if any, off stdout too)
    dog = Watchdog(base, rank, info)
    limit = float(os.environ.get('TE_BENCH_COMM_TIMEOUT', '60'))
    try:
        dog.arm('init_process_group (env:// rendezvous)', 3 * limit)
        dist.init_process_group(backend, init_method='env://')
        dog.arm(f'the first all-reduce over {info["backend"]}', limit)
        t0 = time.perf_counter()
        probe = torch.full((1024,), float(rank + 1), device=dev)
        dist.all_reduce(probe)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        info['preflight_allreduce_s'] = time.perf_counter() - t0
        want = world * (world + 1) / 2
        if abs(float(probe[0].item()) - want) > 1e-3:
            raise RuntimeError(f'preflight all-reduce returned {float(probe[0].item())}, expected {want}')
        dog.arm('the device-id exchange', limit)
        props = torch.cuda.get_device_properties(dev) if dev.type == 'cuda' else None
        mine = {'rank': rank, 'local_rank': local_rank, 'device': torch.cuda.current_device() if props else 'cpu',
                'name': props.name if props else 'cpu', 'pci_bus_id': getattr(props, 'pci_bus_id', None), 'host': os.uname().nodename}
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        info['ranks'] = allv
        info['hip_version'] = getattr(torch.version, 'hip', None)
        # from here on only a coarse limit for the whole run (a rank that dies or diverges mid-run leaves the others waiting
        # in a collective): the record then says where instead of the driver's limit killing a silent job
        dog.arm('the benchmark run after a successful preflight', float(os.environ.get('TE_BENCH_RUN_TIMEOUT', '1500')))
    except Exception as e:                                          # an error (not a hang): same record, then stop
        dog.disarm()
        msg = f'{type(e).__name__}: {e}'
        print(f'bench.py: communication preflight failed on rank {rank}: {msg}', file=sys.stderr, flush=True)
        if rank == 0:
            print(compact_line(dict(base, value=None, ms_per_step=None, comm=dict(info, error=msg[:2000]))), flush=True)
        os._exit(3)
    return info, dog


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU path for the product)'
    # TE_BENCH_SHARE_GPU=1 (rehearsal of the multi-process path on a 1-GPU box): every rank uses cuda:0 and the
    # collectives go through gloo instead of RCCL, which refuses two ranks on one device.  Never set by the driver.
    share = os.environ.get('TE_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    backend = None
    comm_info = watchdog = None
    # TE_BENCH_FORCE_DIST=1 (rehearsal on a 1-GPU box, never set by the driver): run the multi-rank code path - RCCL process
    # group, preflight, hook-driven bucketed exchange, `comm` object - with a world of ONE rank
    dist_on = world > 1 or os.environ.get('TE_BENCH_FORCE_DIST') == '1'
    if os.environ.get('TE_BENCH_FORCE_DIST') == '1':
        os.environ['TE_GRADSYNC_FORCE'] = '1'
        for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29577'), ('RANK', '0'), ('WORLD_SIZE', '1')):
            os.environ.setdefault(k, v)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = 'gloo' if share else 'nccl'                       # "nccl" == RCCL on ROCm
        comm_info, watchdog = comm_preflight(args, backend, world, rank, local_rank, dev)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    size = args.size
    B = args.batch if args.batch is not None else (4 if size >= 1024 else 16)
    timer = KernelTimer()
    if not args.no_kernel_timing:
        timer.install()
    base = {'metric': METRIC, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic'}

    if args.workload == 'sample':
        run_sampling(args, size, B if args.batch is not None else 8, dev, base)
        return

    if args.workload == 'generator':
        elapsed, roof = run_generator(size, B, args.steps, args.warmup, dev, world, rank, timer)
        if rank == 0:
            cfg = 'configs[4]' if size >= 1024 else 'configs[1]'
            out = dict(base, value=world * B * args.steps / elapsed, ms_per_step=1e3 * elapsed / args.steps,
                       config={'workload': f'FFHQ-{size} generator fwd+bwd ONLY (BASELINE {cfg}; NOT the G+D metric), batch '
                                           f'{B}/GPU, num_trans=8, random-init weights, random latents',
                               'global_batch': world * B, 'parallelism': f'dp{world}',
                               'per_gpu_images_per_sec': B * args.steps / elapsed, 'arithmetic': _arithmetic_note(),
                               **_arithmetic_switches()})
            if roof:
                out['roofline'] = roof
                if world == 1:
                    torch.cuda.synchronize()
                    attach_counters(roof, live=not args.no_pmc)
            if world == 1 and not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline_generator(size)
        final_line(out if rank == 0 else None, rank, world, dist_on)
        if dist_on:
            watchdog.disarm()
            torch.distributed.destroy_process_group()
        return

    # ---------------------------------------------------------------- default: the full G+D training iteration
    from transeditor_amd.train_step import TrainStep, default_args
    targs = default_args(size=size, batch=B)
    torch.manual_seed(1234)                                        # same init on every rank
    ts = TrainStep(targs, dev)
    clock = SubstepClock(ts)
    torch.manual_seed(1000 + rank)
    reals = [torch.randn(B, 3, size, size, device=dev).clamp(-1, 1) for _ in range(4)]   # resident in HBM

    for i in range(args.warmup):                                   # iteration 0 fires both lazy regularisers
        ts.iteration(i, reals[i % 4])
    if dist_on:
        # GradSync learns the used-parameter mask of every call kind during its first two calls (a tiny MAX all-reduce with a
        # host read-back); make sure no kind is still learning inside the timed window
        for _ in range(2):
            ts.r1_step(reals[0])
            ts.path_step()
    fence(world)
    timer.reset()
    timer.enabled = timer.installed
    clock.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts.iteration(i, reals[i % 4])
    fence(world)
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    clock.enabled = False
    elapsed = max_over_ranks(elapsed, world, dev)

    n_r1 = sum(1 for i in range(args.steps) if i % targs.d_reg_every == 0)
    n_path = sum(1 for i in range(args.steps) if i % targs.g_reg_every == 0)
    out = dict(base, value=world * B * args.steps / elapsed, ms_per_step=1e3 * elapsed / args.steps,
               config={'workload': f'FFHQ-{size} full G+D training iteration (BASELINE configs[{2 if world == 1 else 3}]; '
                                   f'train_spatial_query.py:166-294): D step, R1 every {targs.d_reg_every}, G step, path-length '
                                   f'regulariser every {targs.g_reg_every} on batch {B // targs.path_batch_shrink}, Adam, EMA; '
                                   f'batch {B}/GPU, num_trans=8, random-init weights, synthetic real images',
                       'global_batch': world * B, 'parallelism': f'dp{world}',
                       'per_gpu_images_per_sec': B * args.steps / elapsed,
                       'lazy_steps_in_window': {'r1': n_r1, 'path': n_path, 'of_iterations': args.steps},
                       'arithmetic': _arithmetic_note(), **_arithmetic_switches()})
    if timer.installed:
        out['roofline'] = timer.roofline(elapsed, args.steps)
    out['substeps'] = clock.summary(targs)
    cw = out['substeps'].get('cadence_weighted_ms_per_iteration')
    if cw:
        out['substeps']['cadence_weighted_images_per_sec_per_gpu'] = 1e3 * B / cw

    if dist_on:
        # ---- comm evidence, outside the timed region
        import torch.distributed as dist
        ts.g_step()                                                # leaves averaged G grads in place
        torch.cuda.synchronize()
        chk = grads_checksum(ts.generator, world, dev)
        per_iter_bytes = ts.g_sync.bytes_per_call() * (1 + 1 / targs.g_reg_every) \
            + ts.d_sync.bytes_per_call() * (1 + 1 / targs.d_reg_every)
        iso_g = isolated_allreduce(ts.g_sync, dev)
        # exposed communication: the same iterations with the exchange switched off
        ts.g_sync.enabled = ts.d_sync.enabled = False
        k2 = min(args.steps, 8)
        fence(world)
        t0 = time.perf_counter()
        for i in range(k2):
            ts.iteration(1 + i, reals[i % 4])                      # indices 1..8: path at 4, 8; no R1
        fence(world)
        nocomm = max_over_ranks(time.perf_counter() - t0, world, dev) / k2
        ts.g_sync.enabled = ts.d_sync.enabled = True
        fence(world)
        t0 = time.perf_counter()
        for i in range(k2):
            ts.iteration(1 + i, reals[i % 4])
        fence(world)
        withcomm = max_over_ranks(time.perf_counter() - t0, world, dev) / k2
        out['comm'] = {**comm_info, 'backend': backend + (' (RCCL)' if backend == 'nccl' else ''), 'world_size': dist.get_world_size(),
                       'bytes_allreduced_per_iteration_per_gpu': per_iter_bytes,
                       'g_bytes_per_exchange': ts.g_sync.bytes_per_call(), 'd_bytes_per_exchange': ts.d_sync.bytes_per_call(),
                       'buckets_g': len(ts.g_sync.buckets), 'buckets_d': len(ts.d_sync.buckets),
                       'isolated_allreduce_per_bucket_g': iso_g,
                       'ms_per_iteration_with_exchange': 1e3 * withcomm, 'ms_per_iteration_without_exchange': 1e3 * nocomm,
                       'exposed_comm_ms_per_iteration': 1e3 * (withcomm - nocomm),
                       'grad_bit_identity': chk}
        if not chk['identical_across_ranks'] and rank == 0:       # reported in the line itself; never lose the measurement over it
            print(f'WARNING: averaged gradients differ across ranks: {chk}', file=sys.stderr, flush=True)

    if world == 1 and rank == 0 and not args.no_sub:
        # ---- named sub-benchmarks: configs[1] (the generator of the train step, same weights) and configs[4]
        sub = {}
        el, roof = run_generator(size, B, 10, 3, dev, 1, 0, timer, G=ts.generator)
        sub['generator_fwd_bwd_256_b16'] = {
            'config': f'BASELINE configs[1]: FFHQ-{size} generator fwd+bwd, batch {B}, 3 warm-up + 10 timed steps',
            'value': B * 10 / el, 'unit': 'images/sec', 'ms_per_step': 1e2 * el, 'roofline': _slim(roof)}
        del ts, clock
        torch.cuda.empty_cache()
        if size == 256:
            # (4 warm-up steps: with 2 the caching allocator was still growing inside the 6 timed steps on one evidence box - a fresh model
            #  right after empty_cache() - and the value read 176 img/s where the stand-alone run of the same library gives 216 - 222)
            el, roof = run_generator(1024, 4, 10, 4, dev, 1, 0, timer)
            sub['generator_fwd_bwd_1024_b4'] = {
                'config': 'BASELINE configs[4]: FFHQ-1024 generator fwd+bwd, batch 4, 4 warm-up + 10 timed steps',
                'value': 4 * 10 / el, 'unit': 'images/sec', 'ms_per_step': 1e3 * el / 10, 'roofline': _slim(roof)}
        out['sub_benchmarks'] = sub
    if world == 1 and rank == 0 and 'roofline' in out:
        torch.cuda.synchronize()
        attach_counters(out['roofline'], live=not args.no_pmc)       # counters of THIS run, outside the timed region
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_train(size)
    final_line(out, rank, world, dist_on)
    if dist_on:
        watchdog.disarm()
        torch.distributed.destroy_process_group()


LINE_TARGET, LINE_LIMIT = 4096, 8192      # bytes of the ONE stdout line (the round-5 line was 24.9 KB and the driver's parser gave up)


def _num(v, digits=5):
    """a JSON-safe number: floats rounded to `digits` significant digits, NaN / +-Inf -> None"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        f = float(v)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(f):
        return None
    if f == 0.0:
        return 0.0
    return round(f, digits - 1 - int(math.floor(math.log10(abs(f)))))


def _pick(d, keys):
    return {k: _num(d[k]) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def compact_roofline(roof):
    """the contract's roofline object for the DOMINANT kernel (the split-bf16 Winograd kernel: executed bf16-pipe FLOPs / launch time
    against the dense bf16 MFMA peak), numbers only; `frac_all_launches` = the composite over every MFMA convolution launch"""
    if not roof:
        return None
    dom = roof.get('dominant_kernel')
    r = {'bound': roof.get('bound', 'mfma'), 'unit': 'TFLOP/s'}
    if dom:
        r.update(kernel=dom['kernel'], achieved=_num(dom['achieved']), peak=_num(dom['peak']), frac=_num(dom['frac']),
                 achieved_algorithmic=_num(dom['achieved_algorithmic']), executed_per_algorithmic_flop=_num(dom.get('executed_factor')),
                 launches=dom['launches'], ms_per_step=_num(dom['ms_per_step']), share_of_step=_num(dom['share_of_step']))
    else:                                                            # (fp32 instructions everywhere: TE_SPLIT_BF16=0)
        r.update(kernel='all 3x3 MFMA launches', achieved=_num(roof.get('achieved')), peak=_num(roof.get('peak')), frac=_num(roof.get('frac')),
                 achieved_algorithmic=_num(roof.get('achieved_algorithmic')))
    r.update(frac_all_launches=_num(roof.get('frac')), achieved_algorithmic_all_launches=_num(roof.get('achieved_algorithmic')),
             mfma_time_share_of_step=_num(roof.get('kernel_time_share')), algorithmic_gflop_per_step=_num(roof.get('algorithmic_gflop_per_step')),
             whole_step_algorithmic_tflops=_num(roof.get('whole_step_tflops')),
             traffic=_num(roof.get('traffic'), 6), traffic_source=roof.get('traffic_source'))
    top = (roof.get('counters') or {}).get('conv3x3_fwd_128to128_at256_b16')
    if top:
        r.update(traffic_over_algorithmic=_num(top.get('traffic_over_algorithmic')), traffic_algorithmic=_num(top.get('algorithmic_bytes'), 6))
    r.update(mfma_util_pct=_num(roof.get('mfma_util_pct')), mhz=_num(roof.get('mhz')))
    if roof.get('live_counters_error'):
        r['counters_error'] = str(roof['live_counters_error'])[:120]
    return {k: v for k, v in r.items() if v is not None}


def compact_line(out):
    """The ONE stdout line: the driver-contract keys, `config` (short workload string, switches as booleans), `roofline` (dominant
    kernel, numbers only), `cpu_baseline`, `substeps`, `sub_benchmarks` (value + frac), `comm` (N > 1, numbers only).  Everything
    else - per-kernel tables, counters, HBM-bound kernel list, notes - goes to bench_detail.json (write_detail).  Target <= 4 KB;
    never above 8 KB: optional groups are dropped, in a fixed order, until it fits."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')
    line = {k: _num(out.get(k), 7) for k in keep if k in out}
    cfg = out.get('config') or {}
    c = {'workload': str(cfg.get('workload', ''))[:300]}
    for k in ('global_batch', 'parallelism', 'per_gpu_images_per_sec', 'split_bf16', 'split_bf16_strided', 'split_bf16_transposed',
              'split_bf16_weight_gradient_3x3', 'split_bf16_1x1'):
        if k in cfg:
            c[k] = _num(cfg[k])
    if cfg.get('lazy_steps_in_window'):
        c['lazy_steps_in_window'] = cfg['lazy_steps_in_window']
    if 'arithmetic_short' in cfg:
        c['arithmetic'] = str(cfg['arithmetic_short'])[:200]
    line['config'] = c
    roof = compact_roofline(out.get('roofline'))
    if roof:
        line['roofline'] = roof
    cb = out.get('cpu_baseline')
    if cb:
        line['cpu_baseline'] = {**_pick(cb, ('value', 'unit', 'cores', 'kind')), 'sample': str(cb.get('sample_short') or cb.get('sample', ''))[:160],
                                **_pick(cb, ('batch', 'cpu_model', 'host_threads')),
                                'seconds': {k: _num(v, 4) for k, v in (cb.get('seconds') or {}).items()}}
    ss = out.get('substeps')
    if ss:
        line['substeps'] = _pick(ss, ('d_ms', 'r1_ms', 'g_ms', 'path_ms', 'cadence_weighted_ms_per_iteration', 'cadence_weighted_images_per_sec_per_gpu'))
    sub = out.get('sub_benchmarks')
    if sub:
        line['sub_benchmarks'] = {}
        for name, v in sub.items():
            rr = v.get('roofline') or {}
            dom = rr.get('dominant_kernel') or {}
            line['sub_benchmarks'][name] = {**_pick(v, ('value', 'unit', 'ms_per_step')), 'frac': _num(dom.get('frac', rr.get('frac'))),
                                            'frac_all_launches': _num(rr.get('frac')),
                                            'achieved_algorithmic_all_launches': _num(rr.get('achieved_algorithmic'))}
    cm = out.get('comm')
    if cm:
        k = _pick(cm, ('backend', 'world_size', 'rccl_version', 'hip_version', 'preflight_allreduce_s', 'bytes_allreduced_per_iteration_per_gpu',
                       'g_bytes_per_exchange', 'd_bytes_per_exchange', 'buckets_g', 'buckets_d', 'ms_per_iteration_with_exchange',
                       'ms_per_iteration_without_exchange', 'exposed_comm_ms_per_iteration'))
        iso = cm.get('isolated_allreduce_per_bucket_g') or []
        if iso:
            k['isolated_allreduce_g'] = {'buckets': len(iso), 'bytes': sum(b['bytes'] for b in iso), 'ms': _num(sum(b['ms'] for b in iso)),
                                         'bus_GBps_largest': _num(max(iso, key=lambda b: b['bytes'])['bus_GBps'])}
        if cm.get('grad_bit_identity'):
            k['grads_identical_across_ranks'] = bool(cm['grad_bit_identity'].get('identical_across_ranks'))
        if cm.get('error'):
            k['error'] = str(cm['error'])[:300]
        line['comm'] = k
    if out.get('detail'):
        line['detail'] = out['detail']
    for drop in (None, ('sub_benchmarks',), ('substeps',), ('comm',), ('cpu_baseline', 'seconds'), ('config', 'lazy_steps_in_window')):
        if drop is not None:
            tgt = line
            for k in drop[:-1]:
                tgt = tgt.get(k, {})
            tgt.pop(drop[-1], None)
        text = json.dumps(line, allow_nan=False, separators=(',', ':'))
        if len(text) <= LINE_LIMIT:
            break
    assert len(text) <= LINE_LIMIT and '\n' not in text, len(text)
    return text


def write_detail(out):
    """the full record (per-kernel tables, counters, notes) beside the line: gpurun_out/bench_detail.json (merged back from the GPU
    box) or $TE_BENCH_DETAIL; returns the path written (relative to the repo root) or None.  Never raises."""
    path = os.environ.get('TE_BENCH_DETAIL') or os.path.join(ROOT, 'gpurun_out', 'bench_detail.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(out, f, indent=1, default=str)
        return os.path.relpath(path, ROOT)
    except Exception:
        try:
            path = f'/tmp/te_bench_detail.{os.getpid()}.json'
            with open(path, 'w') as f:
                json.dump(out, f, indent=1, default=str)
            return path
        except Exception:
            return None


def final_line(out, rank, world, dist_on):
    """rank 0's ONE JSON line, as the LAST thing any rank writes to stdout: whatever native libraries left in the C stdio
    buffers of the ranks is flushed first, then a barrier, then the line (compact_line; the full record goes to write_detail)"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist_on and world > 1:
        torch.distributed.barrier()
    if rank == 0:
        out['detail'] = write_detail(out)
        print(compact_line(out), flush=True)


def _slim(roof):
    if roof is None:
        return None
    return {k: v for k, v in roof.items() if k not in ('traffic_note', 'kernel', 'traffic', 'traffic_source')}


if __name__ == '__main__':
    main()
