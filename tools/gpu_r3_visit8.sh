#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_resblock.py tests/test_gpu_train_step.py tests/test_gpu_generator.py tests/test_gpu_timed_second_order.py tests/test_gpu_conv_fuzz.py -m gpu -q -x > gpurun_out/v8_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v8_pytest.log
tail -5 gpurun_out/v8_pytest.log
timeout 300 python bench.py --steps 16 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/v8_bench.json 2> gpurun_out/v8_bench.err; tail -1 gpurun_out/v8_bench.json | cut -c1-330
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v8_bench.json').read().strip().splitlines()[-1]); print(d['substeps'])
PY
