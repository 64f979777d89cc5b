#!/bin/bash
# A/B of the tile order over the XCDs (banded, the default, vs TE_XCD_INTERLEAVED=1): time of the conv launches at the FFHQ-256
# batch-16 shapes (tools/exp_time.py) and FETCH_SIZE of the top shapes (tools/kernel_once.py under rocprofv3 --pmc).
mkdir -p gpurun_out/band
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 0 1; do
  export TE_XCD_INTERLEAVED=$mode
  ( cd $R; timeout 200 python tools/exp_time.py product ) > $R/gpurun_out/band/time_$mode.log 2>&1
  ( cd /tmp; REP=2 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/band -o fetch_$mode --output-format csv -- python $R/tools/kernel_once.py ) > $R/gpurun_out/band/fetch_$mode.log 2>&1
done
cd $R
python - <<'PY'
import csv, collections, re
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); return n.split('(')[0]
for mode in (0, 1):
    agg = collections.defaultdict(list); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f'gpurun_out/band/fetch_{mode}_counter_collection.csv')):
        k = short(r['Kernel_Name'])
        if 'conv_mfma' in k or 'wino3x3' in k:
            agg[k].append(float(r['Counter_Value'])); dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
    print('== interleaved' if mode else '== banded')
    for k in agg: print(f'  {k[:60]:60s} HBM read {2 * agg[k][-1] * 1024 / 1e6:9.1f} MB   {sorted(dur[k])[len(dur[k]) // 2]:9.1f} us (under counters)')
PY
for mode in 0 1; do echo "== time, TE_XCD_INTERLEAVED=$mode"; grep -E "TF/s" gpurun_out/band/time_$mode.log | grep -v -E "W3X3|WT2"; done
rm -f gpurun_out/band/*.csv gpurun_out/band/*.db
( timeout 300 python tools/conv_fuzz.py 600 7 ) 2>&1 | tail -1
( timeout 300 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv_fuzz.py -m gpu -x -q 2>&1 | tail -2 )
