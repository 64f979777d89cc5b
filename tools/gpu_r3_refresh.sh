#!/bin/bash
# short refresh of the headline evidence: the driver's bench command, the training-iteration kernel trace, the framework-op census
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3r_bench_n1.json 2> gpurun_out/r3r_bench_n1.err; echo "bench rc=$?"
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-sub --no-kernel-timing ) > $R/gpurun_out/r3r_rocprof_train.log 2>&1; echo "rocprof train rc=$?"
cd $R
python tools/rocpd_stats.py gpurun_out/prof_train/train_results.db > gpurun_out/r3r_train_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_train
( timeout 200 python tools/step_census.py 16 ) > gpurun_out/r3r_step_census.txt 2> gpurun_out/r3r_step_census.err; echo "census rc=$?"
tail -1 gpurun_out/r3r_train_kernel_stats.txt; grep "^TOTAL" gpurun_out/r3r_step_census.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r3r_bench_n1.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_step_frac"], {k:round(v,2) for k,v in d["substeps"].items() if k.endswith("_ms")})
for k,v in d["sub_benchmarks"].items(): print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["whole_step_frac"])
PY
