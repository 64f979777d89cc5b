#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_wgrad6.py -q --no-header -p no:cacheprovider -rA ) > gpurun_out/r5v14_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r5v14_tests.log | tail -5
for i in 1 2; do for cfg in "0 0" "1 0" "1 1"; do set -- $cfg
  TE_SPLIT_WGRAD=$1 TE_SPLIT_WGRAD_T2=$2 timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r5v14_bench_$1$2.$i.json 2> gpurun_out/r5v14_bench_$1$2.$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5v14_bench_$1$2.$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("WGRAD=$1 T2=$2 run $i:", round(d["value"],2), "img/s", round(d["ms_per_step"],2), "ms  frac", round(r["frac"],3), {k:round(v["tflops"],1) for k,v in r["per_kernel"].items() if k.startswith("wgrad")}, {k:round(v,1) for k,v in d["substeps"].items() if k.endswith("_ms")})
PY
done; done
