#!/bin/bash
# re-record the default bench line and the PMC summary at HEAD (the counter summary now keeps the split weight-gradient kernels)
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_n1.json 2> gpurun_out/r5_bench_n1.err; echo "bench rc=$?"
rm -rf gpurun_out/pmc
bash tools/pmc_round.sh > gpurun_out/r5_pmc_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r5 > gpurun_out/r5_pmc_summary_stdout.txt 2>&1
rm -f gpurun_out/pmc/*.db gpurun_out/pmc/*.csv
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_bench_n1.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], "frac", r["frac"], "alg", r["achieved_algorithmic"], "whole", r["whole_step_frac"], "dominant", r["dominant_kernel"]["frac"], r.get("mfma_util_pct"), r.get("mhz"), r.get("traffic_source"))
print({k:(round(v["mfma_util_pct"],1),round(v["mhz"])) for k,v in r.get("counters",{}).items()})
for k,v in d["sub_benchmarks"].items(): print(k, v["value"], v["ms_per_step"])
PY
cat gpurun_out/r5_pmc_summary.txt | cut -c1-125
