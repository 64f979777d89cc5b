#!/bin/bash
# round 6, visit 5: Winograd GPU tests with the two-image form; same-box A/B of the iteration: TE_W6_FORM=1 (ping-pong) against 2 (default)
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_winograd.py -q --no-header -p no:cacheprovider -x ) > gpurun_out/r6v5_pytest_winograd.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6v5_pytest_winograd.log
for i in 1 2; do for f in 1 2; do
  TE_W6_FORM=$f timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r6v5_bench_form$f.$i.json 2> gpurun_out/r6v5_bench_form$f.$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6v5_bench_form$f.$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("TE_W6_FORM=$f run $i:", d["value"], "img/s", d["ms_per_step"], "ms  dominant", r["kernel"], r["frac"], "alg", r["achieved_algorithmic"], "ms/step", r["ms_per_step"], d["substeps"])
PY
done; done
