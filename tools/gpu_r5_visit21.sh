#!/bin/bash
# same-box A/B of the whole split-bf16 switch: every MFMA launch on the fp32 matrix instructions (TE_SPLIT_BF16=0) against the product
mkdir -p gpurun_out
for i in 1 2; do for on in 0 1; do
  TE_SPLIT_BF16=$on timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r5v21_bench_split$on.$i.json 2> gpurun_out/r5v21_bench_split$on.$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5v21_bench_split$on.$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("TE_SPLIT_BF16=$on run $i:", round(d["value"],2), "img/s", round(d["ms_per_step"],2), "ms  frac", round(r["frac"],3), "alg", round(r["achieved_algorithmic"],1), {k:round(v["tflops"],1) for k,v in r["per_kernel"].items() if not k.endswith("split_bf16")})
PY
done; done
