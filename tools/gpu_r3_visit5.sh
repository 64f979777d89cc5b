#!/bin/bash
# round-3 visit: rocBLAS-free recorded backward (closed bmm family), latent hint for the path step
mkdir -p gpurun_out; R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_step.py tests/test_gpu_generator.py tests/test_gpu_timed_second_order.py -m gpu -x -q > gpurun_out/v5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v5_pytest.log
tail -5 gpurun_out/v5_pytest.log
cd /tmp; export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_path5 -o path -- python $R/tools/reg_steps_trace.py path ) > $R/gpurun_out/v5_rocprof_path.log 2>&1; echo "rocprof path rc=$?"
cd $R
python tools/rocpd_stats.py gpurun_out/prof_path5/path_results.db > gpurun_out/v5_path_step_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_path5
grep -c Cijk gpurun_out/v5_path_step_kernel_stats.txt; tail -1 gpurun_out/v5_path_step_kernel_stats.txt
timeout 300 python bench.py --steps 16 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/v5_bench.json 2> gpurun_out/v5_bench.err; tail -1 gpurun_out/v5_bench.json | cut -c1-330
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v5_bench.json').read().strip().splitlines()[-1]); print(d['substeps'])
PY
