#!/bin/bash
# round-4 visit 5: the split-bf16 Winograd kernel end to end - its own tests, then the whole GPU tier and the driver's bench command
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_winograd.py -m gpu -x -q -s 2>&1 | grep -E "split-bf16|passed|failed|Error|error" | tail -20
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rA ) > gpurun_out/r4s_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r4s_pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r4s_pytest_gpu.log | head
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4s_bench_n1.json 2> gpurun_out/r4s_bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4s_bench_n1.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], r["frac"], r["whole_step_frac"], r.get("executed_frac"), {k:round(v,2) for k,v in d["substeps"].items() if k.endswith("_ms")})
print({k:(round(v["tflops"],1), round(v["total_ms"],1)) for k,v in r["per_kernel"].items()})
for k,v in d["sub_benchmarks"].items(): print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["whole_step_frac"])
print(r.get("traffic_source"), r.get("live_counters_error"), {k:(round(v["traffic_over_algorithmic"],2), round(v["mfma_util_pct"],1)) for k,v in r.get("counters",{}).items()})
PY
