#!/bin/bash
TAG=r5v8
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_t2s6.py tests/test_gpu_s2s6.py tests/test_gpu_resblock.py tests/test_gpu_generator.py tests/test_gpu_train_iteration_256.py tests/test_gpu_determinism.py tests/test_gpu_timed_shapes.py -m gpu -q --no-header -p no:cacheprovider -rA ) > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | head -20
grep -E "transposed.*L2|pinned" gpurun_out/${TAG}_pytest.log | cut -c1-260 | head -16
for f in 1 0 1 0; do
( TE_SPLIT_T2=$f timeout 400 python bench.py --gpus 1 --steps 16 --warmup 3 --no-sub --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_bench_t2_$f.json 2> gpurun_out/${TAG}_bench_t2_$f.err; echo "bench TE_SPLIT_T2=$f rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_t2_$f.json").read().strip().splitlines()[-1])
print("split_t2=$f", round(d["value"],2), round(d["ms_per_step"],2), {k:round(v["tflops"],1) for k,v in d["roofline"]["per_kernel"].items()}, round(d["roofline"]["frac"],3))
PY
done
