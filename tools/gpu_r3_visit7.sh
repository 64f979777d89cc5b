#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_train_step.py -m gpu -q > gpurun_out/v7_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v7_pytest.log
tail -4 gpurun_out/v7_pytest.log
timeout 300 python bench.py --steps 16 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/v7_bench.json 2> gpurun_out/v7_bench.err; tail -1 gpurun_out/v7_bench.json | cut -c1-330
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v7_bench.json').read().strip().splitlines()[-1]); print(d['substeps'])
PY
