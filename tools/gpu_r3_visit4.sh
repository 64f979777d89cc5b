#!/bin/bash
# round-3 visit: D's fake + real passes as one batch (chunked minibatch stddev)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_generator.py -m gpu -x -q > gpurun_out/v4_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v4_pytest.log
tail -5 gpurun_out/v4_pytest.log
timeout 300 python bench.py --steps 16 --warmup 5 --no-cpu-baseline > gpurun_out/v4_bench.json 2> gpurun_out/v4_bench.err; tail -1 gpurun_out/v4_bench.json | cut -c1-400
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v4_bench.json').read().strip().splitlines()[-1]); print(d['substeps'])
PY
