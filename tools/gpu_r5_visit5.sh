#!/bin/bash
TAG=r5v5
( timeout 300 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_determinism.py -m gpu -q --no-header -p no:cacheprovider -x ) > gpurun_out/${TAG}_pytest_winograd.log 2>&1; echo "pytest winograd+determinism rc=$?"; tail -2 gpurun_out/${TAG}_pytest_winograd.log
for f in 1 0 1 0; do
( TE_W6_FORM=$f timeout 400 python bench.py --gpus 1 --steps 16 --warmup 3 --no-sub --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_bench_form$f.json 2> gpurun_out/${TAG}_bench_form$f.err; echo "bench form$f rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_form$f.json").read().strip().splitlines()[-1])
print("form$f", round(d["value"],2), round(d["ms_per_step"],2), {k:round(v["tflops"],1) for k,v in d["roofline"]["per_kernel"].items()})
PY
done
