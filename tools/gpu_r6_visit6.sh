#!/bin/bash
# round 6, visit 6: the 32-channel forms (wgrad6 pair / narrow, wino6p narrow): GPU tests, FFHQ-1024 generator bench + kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_wgrad6.py tests/test_gpu_winograd.py -q --no-header -p no:cacheprovider -rA ) > gpurun_out/r6v6_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r6v6_pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r6v6_pytest.log | head -20
grep -E "pair form|reducer on pair|transposed kind (64|32|96)" gpurun_out/r6v6_pytest.log | head -20
( timeout 300 python bench.py --workload generator --size 1024 --steps 8 --warmup 3 --no-cpu-baseline --no-pmc ) > gpurun_out/r6v6_bench_g1024.json 2> gpurun_out/r6v6_bench_g1024.err; echo "bench 1024 rc=$?"; cat gpurun_out/r6v6_bench_g1024.json | cut -c1-1500
( TE_SPLIT_WGRAD=0 timeout 300 python bench.py --workload generator --size 1024 --steps 8 --warmup 3 --no-cpu-baseline --no-pmc ) > gpurun_out/r6v6_bench_g1024_nowg.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r6v6_bench_g1024_nowg.json').read().strip().splitlines()[-1]); print('TE_SPLIT_WGRAD=0:', d['value'], d['ms_per_step'])"
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g1024 -o g1024 -- python $R/bench.py --workload generator --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pmc ) > $R/gpurun_out/r6v6_rocprof_g1024.log 2>&1; echo "rocprof 1024 rc=$?"
cd $R
python tools/rocpd_stats.py gpurun_out/prof_g1024/g1024_results.db > gpurun_out/r6v6_g1024_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_g1024
head -24 gpurun_out/r6v6_g1024_kernel_stats.txt | cut -c1-150
