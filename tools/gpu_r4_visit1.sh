#!/bin/bash
# round-4 visit 1: full GPU tier (with the pinned-slope / determinism / 256-px iteration tests), bench with live PMC, 2-rank rehearsal
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1100 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rA --maxfail=25 ) > gpurun_out/r4a_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r4a_pytest_gpu.log | tail -3
grep -E "pinned|FAILED|ERROR" gpurun_out/r4a_pytest_gpu.log | head -60
( timeout 600 python bench.py ) > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r4a_bench.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('bench', d['value'], d['ms_per_step'], r['frac'], r['whole_step_frac'], r.get('traffic_source'), r.get('traffic'), r.get('mfma_util_pct'), r.get('live_counters_error'))
    print(json.dumps(r.get('counters'))[:1500])
    print(d['substeps'])
    print({k:(v['value'], v['roofline']['whole_step_frac']) for k,v in d['sub_benchmarks'].items()})
except Exception as e:
    print('bench parse failed', e)
PY
tail -5 gpurun_out/r4a_bench.err
( TE_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 ) > gpurun_out/r4a_bench_n2_shared.json 2> gpurun_out/r4a_bench_n2_shared.err; echo "n2 rc=$?"
tail -c 1500 gpurun_out/r4a_bench_n2_shared.json
