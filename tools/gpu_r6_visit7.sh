#!/bin/bash
# round 6, visit 7: wgrad6 chunk order (XCD-banded, adjacent column tiles side by side) + batch-split slab reducer: tests, counters, timing
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_gpu_wgrad6.py tests/test_gpu_determinism.py tests/test_gpu_train_step.py tests/test_gpu_timed_shapes.py -q --no-header -p no:cacheprovider -x ) > gpurun_out/r6v7_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6v7_pytest.log
( timeout 300 python tools/wgrad6_check.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v7_wgrad6_check.log; tail -14 gpurun_out/r6v7_wgrad6_check.log | cut -c1-250
rm -rf gpurun_out/pmc
bash tools/pmc_round.sh > gpurun_out/r6v7_pmc_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r6v7 > gpurun_out/r6v7_pmc_summary_stdout.txt 2>&1
rm -rf gpurun_out/pmc
grep -E "^kernel|wgrad" gpurun_out/r6v7_pmc_summary.txt | cut -c1-160
for i in 1 2; do timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r6v7_bench.$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r6v7_bench.$i.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['substeps'])"; done
timeout 300 python bench.py --workload generator --size 1024 --steps 8 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/r6v7_bench_g1024.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r6v7_bench_g1024.json').read().strip().splitlines()[-1]); print('g1024', d['value'], d['ms_per_step'])"
