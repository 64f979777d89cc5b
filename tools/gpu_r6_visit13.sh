#!/bin/bash
# round 6, visit 13: sum_parts with eight loads in flight + the split cap: determinism / reducer tests, FFHQ-1024 + iteration timing
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_wgrad6.py tests/test_gpu_ops.py -q --no-header -p no:cacheprovider -x ) > gpurun_out/r6v13_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r6v13_pytest.log
for i in 1 2; do timeout 300 python bench.py --workload generator --size 1024 --steps 8 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/r6v13_bench_g1024.$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r6v13_bench_g1024.$i.json').read().strip().splitlines()[-1]); print('g1024', d['value'], d['ms_per_step'])"; done
for i in 1 2; do timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r6v13_bench.$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r6v13_bench.$i.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"; done
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g1024 -o g1024 -- python $R/bench.py --workload generator --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pmc ) > $R/gpurun_out/r6v13_rocprof_g1024.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_g1024/g1024_results.db > gpurun_out/r6v13_g1024_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_g1024
grep -E "sum_parts|wgrad_reduce_fused|TOTAL" gpurun_out/r6v13_g1024_kernel_stats.txt | cut -c1-170
