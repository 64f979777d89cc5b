#!/bin/bash
TAG=r5v9
export TMPDIR=/tmp
for b in 1 8; do ( timeout 300 python bench.py --workload sample --batch $b --steps 30 --warmup 5 ) > gpurun_out/r05_sampling_b$b.json 2> gpurun_out/${TAG}_sample_b$b.err; echo "sample b$b rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r05_sampling_b$b.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','unit') if k in d}, str(d.get('config'))[:200])"; done
( timeout 400 python bench.py --workload generator --size 1024 --batch 4 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_g1024.json 2> gpurun_out/${TAG}_g1024.err; echo "g1024 rc=$?"
( timeout 400 python bench.py --workload generator --steps 10 --warmup 3 --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_g256.json 2> gpurun_out/${TAG}_g256.err; echo "g256 rc=$?"
python - <<PY
import json
for f in ("g1024","g256"):
    d=json.loads(open(f"gpurun_out/${TAG}_{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"],1), round(d["ms_per_step"],2), {k:round(v["tflops"],1) for k,v in d["roofline"]["per_kernel"].items()}, round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3))
PY
