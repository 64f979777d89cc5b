#!/bin/bash
TAG=r5v6
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_stock_ddp_loop.py tests/test_gpu_train_step.py::test_checkpoint_layout_round_trip_and_device_prefetcher -m gpu -q --no-header -p no:cacheprovider -rA ) > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | head -20
grep -E "scale sweep|input scale|stock DDP|pinned" gpurun_out/${TAG}_pytest.log | cut -c1-300 | head -40
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench_n1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_n1.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], "frac", r["frac"], "achieved", r["achieved"], "peak", r["peak"], "alg", r["achieved_algorithmic"], "whole", r["whole_step_frac"], "mfma_util", r.get("mfma_util_pct"), r.get("mhz"), r.get("traffic_source"))
print("dominant", r["dominant_kernel"])
print({k:round(v["tflops"],1) for k,v in r["per_kernel"].items()})
for k,v in d.get("sub_benchmarks",{}).items(): print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["whole_step_frac"])
c=d["cpu_baseline"]; print({k:c[k] for k in ("value","cores","batch","seconds")}); print(c["sample"])
print({k:(v["mfma_util_pct"], v["mhz"], v["traffic_over_algorithmic"]) for k,v in r.get("counters",{}).items()})
PY
