"""Summarise the rocprofv3 --pmc passes of tools/pmc_round.sh (gpurun_out/pmc/*.csv) per kernel:
MFMA utilisation, effective clock, HBM bytes (FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950
streaming reads; WRITE_SIZE as reported), LDS bank-conflict share.  Writes profiles/<tag>_pmc_summary.{txt,json}."""
import collections
import csv
import json
import re
import sys

KEEP = ('conv_mfma', 'wino3x3', 'wino6', 's2s6', 't2s6', 't2_edge', 'wgrad6', 'wgrad_mfma', 'wgrad_reduce', 'fir_tile', 'blur44', 'bias_act', 'rgb_')
# algorithmic bytes / flops of the launches in tools/kernel_once.py (B = 16)
ALG = {
    'wino6_kernel': dict(flops=2 * 9 * 128 * 128 * 256 * 256 * 16, bytes=2 * 16 * 128 * 256 * 256 * 4),       # algorithmic (direct-form) FLOPs
    'wino6p_kernel': dict(flops=2 * 9 * 128 * 128 * 256 * 256 * 16, bytes=2 * 16 * 128 * 256 * 256 * 4),      # (the ping-pong form, round 5)
    'wino6q_kernel': dict(flops=2 * 9 * 128 * 128 * 256 * 256 * 16, bytes=2 * 16 * 128 * 256 * 256 * 4),      # (the two-image form, round 6)
    's2s6_kernel': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),
    's2s6q_kernel': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),     # (the two-image form, round 6)
    't2_edge_kernel': dict(flops=2 * 3 * 256 * 128 * (257 + 256) * 16, bytes=16 * (256 * 128 * 2 + 128 * 513) * 4 + 3 * 256 * 128 * 4),
    't2s6_kernel': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),
    't2s6q_kernel': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),     # (the two-image form, round 6)
    'wino3x3_kernel': dict(flops=2 * 9 * 128 * 128 * 256 * 256 * 16, bytes=2 * 16 * 128 * 256 * 256 * 4),     # algorithmic (direct-form) FLOPs
    'conv_mfma_kernel<0': dict(flops=2 * 9 * 128 * 128 * 256 * 256 * 16, bytes=2 * 16 * 128 * 256 * 256 * 4),
    'wgrad6_kernel': dict(flops=2 * 9 * 128 * 128 * 256 * 256 * 16, bytes=2 * 16 * 128 * 256 * 256 * 4),      # split-bf16 weight gradient (round 5)
    'wgrad6t_kernel': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),
    'wgrad6tw_kernel': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),  # (the 64 x 128 form, round 6)
    'wgrad_mfma_kernel<0': dict(flops=2 * 9 * 128 * 128 * 256 * 256 * 16, bytes=2 * 16 * 128 * 256 * 256 * 4),
    'conv_mfma_kernel<1, 0, true, false, 2, true': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),
    'conv_mfma_kernel<2': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),
    'wgrad_mfma_kernel<1': dict(flops=2 * 9 * 256 * 128 * 128 * 128 * 16, bytes=16 * (256 * 128 * 128 + 128 * 257 * 257) * 4),
    'blur44_kernel<1': dict(flops=0, bytes=16 * 128 * (257 * 257 + 2 * 256 * 256) * 4),       # AG: g + ref (256^2) -> 257^2
    'blur44_kernel<2': dict(flops=0, bytes=16 * 128 * (257 * 257 + 2 * 256 * 256) * 4),       # gradact: g (257^2) + ref -> 256^2
    'blur44_kernel<0, 2, true': dict(flops=0, bytes=16 * 128 * (257 * 257 + 256 * 256) * 4),  # adjoint blur 256^2 -> 257^2
    'blur44_kernel<0': dict(flops=0, bytes=16 * 128 * (257 * 257 + 256 * 256) * 4),           # blur + bias + lrelu 257^2 -> 256^2
    'bias_act_bwd_rgb': dict(flops=0, bytes=(3 * 128 + 3) * 16 * 256 * 256 * 4),
    'fir_tile_kernel<1, 1, 4, 4, true': dict(flops=0, bytes=16 * 128 * (257 * 257 + 2 * 256 * 256) * 4),
    'fir_tile_kernel<1, 1': dict(flops=0, bytes=16 * 128 * (257 * 257 + 256 * 256) * 4),
    'wgrad_reduce_fused': dict(flops=0, bytes=16 * 16 * 128 * 128 * 9 * 4),
    'bias_act_bwd_rows': dict(flops=0, bytes=3 * 16 * 128 * 256 * 256 * 4),
    'rgb_fwd': dict(flops=0, bytes=16 * 131 * 256 * 256 * 4),
    'rgb_dgrad': dict(flops=0, bytes=16 * 131 * 256 * 256 * 4),
    'rgb_wgrad': dict(flops=0, bytes=16 * 131 * 256 * 256 * 4),
}


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0]


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not any(s in k for s in KEEP):
            continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
    return agg, dur


def summarise(src, need_lds=True):
    """per-kernel dict from the counter CSVs under `src` (mfma / fetch / write passes; the LDS pass is optional)"""
    import os
    mf, dur = load(f'{src}/mfma_counter_collection.csv')
    fe, _ = load(f'{src}/fetch_counter_collection.csv')
    wr, _ = load(f'{src}/write_counter_collection.csv')
    have_lds = os.path.exists(f'{src}/lds_counter_collection.csv')
    if need_lds and not have_lds:
        raise FileNotFoundError(f'{src}/lds_counter_collection.csv')
    ld = load(f'{src}/lds_counter_collection.csv')[0] if have_lds else {}
    out = {}
    for k in mf:
        last = lambda d, c: (d[k][c][-1] if k in d and c in d[k] else float('nan'))
        us = sorted(dur[k])[len(dur[k]) // 2]
        gui = last(mf, 'GRBM_GUI_ACTIVE') / 8.0                       # summed over the 8 XCDs
        mhz = gui / us
        busy = last(mf, 'SQ_VALU_MFMA_BUSY_CYCLES')
        util = 100.0 * busy / (gui * 1024) if gui else 0.0            # 256 CUs x 4 SIMDs
        rd = 2.0 * last(fe, 'FETCH_SIZE') * 1024 / 1e6                 # gfx950: FETCH_SIZE reports half of a streaming read
        wrm = last(wr, 'WRITE_SIZE') * 1024 / 1e6
        alg = next((v for p, v in ALG.items() if k.startswith(p)), None)
        tf = alg['flops'] / us / 1e6 if alg and alg['flops'] else 0.0
        conf = (100.0 * last(ld, 'SQ_LDS_BANK_CONFLICT') / max(last(ld, 'SQ_LDS_IDX_ACTIVE'), 1.0)) if have_lds else None
        out[k] = dict(us=us, mhz=mhz, mfma_util_pct=util, tflops=tf, hbm_read_mb=rd, hbm_write_mb=wrm,
                      algorithmic_mb=(alg['bytes'] / 1e6 if alg else None), gbs=(rd + wrm) / us * 1e3, lds_conflict_pct=conf)
    return out


def main(src='gpurun_out/pmc', tag='profiles/r04'):
    out = summarise(src, need_lds=False)
    lines = [f'{"kernel":42s} {"us":>8s} {"MHz":>6s} {"MFMA%":>6s} {"TF/s":>6s} {"HBM rd MB":>10s} {"wr MB":>8s} {"alg MB":>8s} '
             f'{"GB/s":>7s} {"LDSconf%":>8s}']
    for k, v in out.items():
        alg = v['algorithmic_mb']
        conf = v['lds_conflict_pct']
        lines.append(f'{k[:42]:42s} {v["us"]:8.1f} {v["mhz"]:6.0f} {v["mfma_util_pct"]:6.1f} {v["tflops"]:6.1f} {v["hbm_read_mb"]:10.1f} '
                     f'{v["hbm_write_mb"]:8.1f} {(alg if alg is not None else float("nan")):8.1f} {v["gbs"]:7.0f} '
                     f'{(conf if conf is not None else float("nan")):8.1f}')
    open(tag + '_pmc_summary.txt', 'w').write('\n'.join(lines) + '\n')
    json.dump(out, open(tag + '_pmc_summary.json', 'w'), indent=1)
    print('\n'.join(lines))


def refresh(tag):
    """recompute the derived columns (TF/s, algorithmic MB) of a stored <tag>_pmc_summary.json after ALG gained entries"""
    out = json.load(open(tag + '_pmc_summary.json'))
    for k, v in out.items():
        alg = next((a for p, a in ALG.items() if k.startswith(p)), None)
        if alg:
            v['tflops'] = alg['flops'] / v['us'] / 1e6 if alg['flops'] else 0.0
            v['algorithmic_mb'] = alg['bytes'] / 1e6
    lines = [f'{"kernel":42s} {"us":>8s} {"MHz":>6s} {"MFMA%":>6s} {"TF/s":>6s} {"HBM rd MB":>10s} {"wr MB":>8s} {"alg MB":>8s} '
             f'{"GB/s":>7s} {"LDSconf%":>8s}']
    for k, v in out.items():
        alg, conf = v['algorithmic_mb'], v['lds_conflict_pct']
        lines.append(f'{k[:42]:42s} {v["us"]:8.1f} {v["mhz"]:6.0f} {v["mfma_util_pct"]:6.1f} {v["tflops"]:6.1f} {v["hbm_read_mb"]:10.1f} '
                     f'{v["hbm_write_mb"]:8.1f} {(alg if alg is not None else float("nan")):8.1f} {v["gbs"]:7.0f} '
                     f'{(conf if conf is not None else float("nan")):8.1f}')
    open(tag + '_pmc_summary.txt', 'w').write('\n'.join(lines) + '\n')
    json.dump(out, open(tag + '_pmc_summary.json', 'w'), indent=1)


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--refresh':
        refresh(sys.argv[2])
    else:
        main(*sys.argv[1:])
