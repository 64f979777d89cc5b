"""Socket power and shader clock under each matrix-pipe kernel (numbers behind "power-limited").

Each kernel runs back to back for SECONDS (default 3) at its FFHQ-256 / batch-16 top shape while a sampler thread reads
the SMU metrics table through the amdsmi Python binding (>= 20 Hz): socket power, the eight per-XCD shader clocks, hotspot
temperature, throttle status.  The first 0.7 s of every loop (clock ramp) is excluded from the means.  Prints one table row
per kernel: launches, HIP-event time per launch, algorithmic and executed TFLOP/s, mean / max socket power, mean shader
clock, the power cap.

    python tools/power_probe.py                         # the product library
    VARIANT=mfma_only python tools/power_probe.py       # tools/exp/libte_<name>.so (tools/exp_build.py mfma_only -DW6_SKIP_COMMIT)
"""
import math
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

if os.environ.get('VARIANT'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f"libte_{os.environ['VARIANT']}.so")
DEV, B = 'cuda', 16
SECONDS = float(os.environ.get('SECONDS', '3'))
ONLY = os.environ.get('ONLY', '')


class Sampler:
    def __init__(self):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[0]
        self.rows, self.run, self.thread = [], False, None
        try:
            self.cap = amdsmi.amdsmi_get_power_cap_info(self.h)
        except Exception as e:
            self.cap = {'error': str(e)}

    def read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clk = [c for c in (m.get('current_gfxclks') or []) if isinstance(c, (int, float)) and 0 < c < 10000]
        self.last = m
        return (time.perf_counter(), m.get('current_socket_power'), sum(clk) / len(clk) if clk else None,
                m.get('temperature_hotspot'), m.get('throttle_status'), m.get('average_gfx_activity'))

    RESIDENCY = ('accumulation_counter', 'prochot_residency_acc', 'ppt_residency_acc', 'socket_thm_residency_acc', 'vr_thm_residency_acc',
                 'hbm_thm_residency_acc', 'energy_accumulator')

    def residency(self):
        """the SMU's throttler residency accumulators (which limiter held the clock down, in accumulation ticks)"""
        self.read()
        return {k: self.last.get(k) for k in self.RESIDENCY}

    def start(self):
        self.rows, self.run = [], True

        def loop():
            while self.run:
                try:
                    self.rows.append(self.read())
                except Exception as e:
                    self.rows.append((time.perf_counter(), None, None, None, f'err {e}', None))
                time.sleep(0.02)
        self.thread = threading.Thread(target=loop, daemon=True)
        self.thread.start()

    def stop(self, t_from):
        self.run = False
        self.thread.join()
        rows = [r for r in self.rows if r[0] >= t_from and isinstance(r[1], (int, float)) and r[1] < 60000]
        n = max(1, len(rows))
        hz = len(self.rows) / max(1e-9, self.rows[-1][0] - self.rows[0][0]) if len(self.rows) > 1 else 0.0
        clk = [r[2] for r in rows if r[2]]
        return {'samples': len(rows), 'hz': hz, 'power_mean': sum(r[1] for r in rows) / n, 'power_max': max((r[1] for r in rows), default=0),
                'mhz_mean': sum(clk) / max(1, len(clk)), 'mhz_min': min(clk, default=0), 'temp_max': max((r[3] or 0 for r in rows), default=0),
                'throttle': sorted({str(r[4]) for r in rows})[:4]}


def kernels():
    """(name, fn, algorithmic FLOP, executed bf16 FLOP factor, executed fp32 factor)"""
    from transeditor_amd.op.modconv import bwd_kinds, fwd_kinds
    out = []
    for K, M, H in ((128, 128, 256), (256, 256, 128), (512, 512, 64)):
        x = torch.randn(B, K, H, H, device=DEV)
        w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
        isc, osc, bias = torch.rand(B, K, device=DEV) + 0.5, torch.rand(B, M, device=DEV) + 0.5, torch.randn(M, device=DEV)
        fl = 2.0 * 9 * K * M * H * H * B
        u6 = _lib.conv_pack(w, _lib.PACK_W6FWD)
        out.append((f'wino6q 3x3 {K}->{M} @{H}', (lambda x=x, u6=u6, M=M, H=H, isc=isc, osc=osc, bias=bias: _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, H, isc, osc, bias, 3)), fl, 4.0, 0.0))
        if H == 256:
            uw = _lib.conv_pack(w, _lib.PACK_WFWD)
            out.append((f'wino3x3 fp32 {K}->{M} @{H}', (lambda x=x, uw=uw, M=M, H=H, isc=isc, osc=osc, bias=bias: _lib.conv(x, uw, _lib.CONV_3X3W, M, H, H, isc, osc, bias, 3)), fl, 0.0, 2.0 / 3))
            ud = _lib.conv_pack(w, _lib.PACK_FWD)
            out.append((f'direct fp32 {K}->{M} @{H}', (lambda x=x, ud=ud, M=M, H=H, isc=isc, osc=osc, bias=bias: _lib.conv(x, ud, _lib.CONV_3X3, M, H, H, isc, osc, bias, 3)), fl, 0.0, 1.0))
        y = torch.randn(B, M, H, H, device=DEV)
        if _lib.wgrad_split() and _lib.wgrad_split_ok(_lib.CONV_3X3, M, K, H, H):
            out.append((f'wgrad6 3x3 {M}x{K} @{H}', (lambda y=y, x=x, H=H: _lib.wgrad_slabs(y, x, _lib.CONV_3X3, H, H)), fl, 4.0, 0.0))
    xl = torch.randn(B, 256, 128, 128, device=DEV)
    wu = torch.randn(128, 256, 3, 3, device=DEV) / 48
    iscu, osc = torch.rand(B, 256, device=DEV) + 0.5, torch.rand(B, 128, device=DEV) + 0.5
    pku, cku = fwd_kinds('up', B, wu, 128, 128)
    wpu = _lib.conv_pack(wu, pku)
    fl = 2.0 * 9 * 256 * 128 * 128 * 128 * B
    t = _lib.conv(xl, wpu, cku, 128, 128, 128, iscu, osc)
    out.append(('t2s6 256->128 @128', (lambda: _lib.conv(xl, wpu, cku, 128, 128, 128, iscu, osc)), fl, 6.0 if cku == _lib.CONV_T2S6 else 0.0, 0.0 if cku == _lib.CONV_T2S6 else 1.0))
    pks, cks = bwd_kinds('up', B, wu, 128, 128)
    wps = _lib.conv_pack(wu, pks)
    out.append(('s2s6 128->256 @128out', (lambda: _lib.conv(t, wps, cks, 256, 128, 128, osc, iscu)), fl, 6.0 if cks == _lib.CONV_S2S6 else 0.0, 0.0 if cks == _lib.CONV_S2S6 else 1.0))
    out.append(('wgrad6t 128x256 @128', (lambda: _lib.wgrad_slabs(t, xl, _lib.CONV_T2, 128, 128)), fl, 6.0, 0.0))
    # an HBM-bound kernel for contrast
    a = torch.randn(B, 128, 256, 256, device=DEV)
    o = torch.randn(B, 128, 256, 256, device=DEV)
    out.append(('bias_act_bwd (HBM) 128 @256', (lambda: _lib.bias_act_bwd(a, o, 0.2, 2 ** 0.5)), 0.0, 0.0, 0.0))
    return out


def main():
    smp = Sampler()
    idle = smp.read()
    print(f'variant {os.environ.get("VARIANT", "product")}; device {torch.cuda.get_device_name(0)}; power cap info {smp.cap}; '
          f'idle: {idle[1]} W, {idle[2]} MHz', flush=True)
    print(f'{"kernel":34s} {"launches":>8s} {"us":>9s} {"alg TF/s":>9s} {"exec bf16":>10s} {"exec fp32":>10s} {"W mean":>7s} {"W max":>6s} '
          f'{"MHz mean":>9s} {"MHz min":>8s} {"T max":>6s} {"pJ/exec FLOP":>13s}  samples@Hz throttle', flush=True)
    for name, fn, flops, f16, f32 in kernels():
        if ONLY and ONLY not in name:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        res0 = smp.residency()
        smp.start()
        t0 = time.perf_counter()
        n, n_timed, ev = 0, 0, None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.perf_counter() - t0 < SECONDS:
            if ev is None and time.perf_counter() - t0 > 0.7:            # event timing starts after the clock ramp
                s.record()
                ev = n
            for _ in range(8):
                fn()
            n += 8
            torch.cuda.synchronize()          # keeps the queue short so wall time == GPU time
        e.record()
        torch.cuda.synchronize()
        r = smp.stop(t0 + 0.7)
        res1 = smp.residency()
        dres = {k.replace('_residency_acc', ''): (res1[k] - res0[k]) for k in res0 if isinstance(res0[k], (int, float)) and isinstance(res1[k], (int, float))}
        us = 1e3 * s.elapsed_time(e) / max(1, n - (ev or 0))
        ex = flops * (f16 + f32)
        pj = r['power_mean'] * us * 1e-6 / ex * 1e12 if ex else float('nan')
        print(f'{name:34s} {n:8d} {us:9.1f} {flops / us / 1e6 if flops else 0:9.1f} {flops * f16 / us / 1e6:10.1f} {flops * f32 / us / 1e6:10.1f} '
              f'{r["power_mean"]:7.0f} {r["power_max"]:6.0f} {r["mhz_mean"]:9.0f} {r["mhz_min"]:8.0f} {r["temp_max"]:6.0f} {pj:13.3f}  '
              f'{r["samples"]}@{r["hz"]:.0f} {r["throttle"]} residency deltas {dres}', flush=True)
        time.sleep(1.0)


if __name__ == '__main__':
    main()
