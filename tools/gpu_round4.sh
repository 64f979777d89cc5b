#!/bin/bash
# Round-4 evidence run on the GPU box: full GPU test tier, smoke, default bench (the driver's command), 2-rank rehearsal of
# the data-parallel path on one GPU (gloo), kernel traces of the benchmarked workloads and of the two regulariser steps, a
# marker trace (TE_ROCTX=1), PMC passes, per-shape census, conv fuzz.   Tag of the outputs: $TAG (default r4).
TAG=${TAG:-r4}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rA ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/${TAG}_pytest_gpu.log | head
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; echo "bench rc=$?"
( TE_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 4 --warmup 2 ) > gpurun_out/${TAG}_bench_n2_shared_gloo.json 2> gpurun_out/${TAG}_bench_n2.err; echo "bench n2 (shared GPU, gloo) rc=$?"
( TE_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --gpus 1 --steps 4 --warmup 2 --no-sub --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_bench_rccl_world1.json 2> gpurun_out/${TAG}_bench_rccl_world1.err; echo "bench RCCL world-1 rehearsal rc=$? lines=$(wc -l < gpurun_out/${TAG}_bench_rccl_world1.json)"
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-sub --no-kernel-timing ) > $R/gpurun_out/${TAG}_rocprof_train.log 2>&1; echo "rocprof train rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gen -o gen -- python $R/bench.py --workload generator --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > $R/gpurun_out/${TAG}_rocprof_gen.log 2>&1; echo "rocprof gen rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g1024 -o g1024 -- python $R/bench.py --workload generator --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > $R/gpurun_out/${TAG}_rocprof_g1024.log 2>&1; echo "rocprof 1024 rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_path -o path -- python $R/tools/reg_steps_trace.py path ) > $R/gpurun_out/${TAG}_rocprof_path.log 2>&1; echo "rocprof path rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o r1 -- python $R/tools/reg_steps_trace.py r1 ) > $R/gpurun_out/${TAG}_rocprof_r1.log 2>&1; echo "rocprof r1 rc=$?"
( TE_ROCTX=1 LD_PRELOAD=/opt/rocm/lib/librocprofiler-sdk-roctx.so timeout 300 rocprofv3 --kernel-trace --marker-trace -d $R/gpurun_out/prof_mark -o mark -- python $R/bench.py --workload generator --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing ) > $R/gpurun_out/${TAG}_rocprof_marker.log 2>&1; echo "rocprof marker rc=$?"
cd $R
for t in train gen g1024 path r1; do python tools/rocpd_stats.py gpurun_out/prof_$t/${t}_results.db > gpurun_out/${TAG}_${t}_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_$t; done
python tools/rocpd_markers.py gpurun_out/prof_mark/mark_results.db > gpurun_out/${TAG}_marker_ranges.txt 2>&1; rm -rf gpurun_out/prof_mark
rm -rf gpurun_out/pmc
bash tools/pmc_round.sh > gpurun_out/${TAG}_pmc_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/${TAG} > gpurun_out/${TAG}_pmc_summary_stdout.txt 2>&1
rm -f gpurun_out/pmc/*.db gpurun_out/pmc/*.csv
( timeout 300 python tools/conv_shape_census.py ) > gpurun_out/${TAG}_conv_shape_census.log 2>&1; echo "shape census rc=$?"
( timeout 300 python tools/conv_fuzz.py 1000 5 ) > gpurun_out/${TAG}_conv_fuzz.log 2>&1; echo "conv fuzz rc=$?"; tail -1 gpurun_out/${TAG}_conv_fuzz.log
( timeout 200 python tools/step_census.py 16 ) > gpurun_out/${TAG}_step_census.txt 2> gpurun_out/${TAG}_step_census.err; echo "step census rc=$?"
( timeout 150 python tools/wgrad_pair_time.py; TE_WGRAD_DIRECT=1 timeout 150 python tools/wgrad_pair_time.py ) > gpurun_out/${TAG}_wgrad_pair_ab.log 2>&1
( timeout 120 python tools/exp_time.py product fir ) > gpurun_out/${TAG}_fir_time.log 2>&1
grep "ms per step" gpurun_out/${TAG}_rocprof_path.log gpurun_out/${TAG}_rocprof_r1.log
tail -c 900 gpurun_out/${TAG}_bench_n1.json | head -c 300; echo
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_n1.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_step_frac"], {k:round(v,2) for k,v in d["substeps"].items() if k.endswith("_ms")})
for k,v in d["sub_benchmarks"].items(): print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["whole_step_frac"])
print({k:v for k,v in d["cpu_baseline"].items() if k in ("value","cores","generator_fwd_bwd_batch16")})
PY
