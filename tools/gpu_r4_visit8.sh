#!/bin/bash
# round-4 visit 8 (HEAD, split-bf16 kernel in the step): multi-rank rehearsals of bench.py on one GPU, then the remaining traces
TAG=r4
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( TE_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 200 python bench.py --gpus 1 --steps 4 --warmup 2 --no-sub --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_bench_rccl_world1.json 2> gpurun_out/${TAG}_bench_rccl_world1.err; echo "bench RCCL world-1 rehearsal rc=$? lines=$(wc -l < gpurun_out/${TAG}_bench_rccl_world1.json)"
( TE_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 4 --warmup 2 ) > gpurun_out/${TAG}_bench_n2_shared_gloo.json 2> gpurun_out/${TAG}_bench_n2.err; echo "bench n2 (shared GPU, gloo) rc=$?"
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g1024 -o g1024 -- python $R/bench.py --workload generator --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pmc ) > $R/gpurun_out/${TAG}_rocprof_g1024.log 2>&1; echo "rocprof 1024 rc=$?"
( timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_path -o path -- python $R/tools/reg_steps_trace.py path ) > $R/gpurun_out/${TAG}_rocprof_path.log 2>&1; echo "rocprof path rc=$?"
( timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o r1 -- python $R/tools/reg_steps_trace.py r1 ) > $R/gpurun_out/${TAG}_rocprof_r1.log 2>&1; echo "rocprof r1 rc=$?"
cd $R
for t in g1024 path r1; do python tools/rocpd_stats.py gpurun_out/prof_$t/${t}_results.db > gpurun_out/${TAG}_${t}_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_$t; done
python - <<PY
import json
for f in ("gpurun_out/r4_bench_rccl_world1.json", "gpurun_out/r4_bench_n2_shared_gloo.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["n_gpus"], {k:(v if not isinstance(v,(dict,list)) else "...") for k,v in d.get("comm",{}).items()} )
PY
grep TOTAL gpurun_out/r4_g1024_kernel_stats.txt gpurun_out/r4_path_kernel_stats.txt gpurun_out/r4_r1_kernel_stats.txt
