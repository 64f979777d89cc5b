#!/bin/bash
# Round-2 evidence run on the GPU box: full GPU test tier, smoke, default bench (the driver's command), 2-rank rehearsal of
# the data-parallel path on one GPU (gloo), kernel traces of the three benchmarked workloads, PMC passes.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) > gpurun_out/r2_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r2_pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc=$?"
( time timeout 900 python bench.py --gpus 1 --steps 16 --warmup 5 ) > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err; echo "bench rc=$?"
( TE_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 4 --warmup 2 ) > gpurun_out/r2_bench_n2_shared_gloo.json 2> gpurun_out/r2_bench_n2.err; echo "bench n2 (shared GPU, gloo) rc=$?"
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-sub --no-kernel-timing ) > $R/gpurun_out/r2_rocprof_train.log 2>&1; echo "rocprof train rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gen -o gen -- python $R/bench.py --workload generator --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > $R/gpurun_out/r2_rocprof_gen.log 2>&1; echo "rocprof gen rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g1024 -o g1024 -- python $R/bench.py --workload generator --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > $R/gpurun_out/r2_rocprof_g1024.log 2>&1; echo "rocprof 1024 rc=$?"
cd $R
for t in train gen g1024; do python tools/rocpd_stats.py gpurun_out/prof_$t/${t}_results.db > gpurun_out/r2_${t}_kernel_stats.txt 2>&1; rm -f gpurun_out/prof_$t/*.db; done
bash tools/pmc_round.sh > gpurun_out/r2_pmc_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r02 > gpurun_out/r2_pmc_summary_stdout.txt 2>&1
rm -f gpurun_out/pmc/*.db
( timeout 300 python tools/conv_shape_census.py ) > gpurun_out/r2_conv_shape_census.log 2>&1; echo "shape census rc=$?"
( timeout 300 python tools/conv_fuzz.py 1000 5 ) > gpurun_out/r2_conv_fuzz.log 2>&1; echo "conv fuzz rc=$?"; tail -1 gpurun_out/r2_conv_fuzz.log
tail -2 gpurun_out/r2_bench_n1.json | cut -c1-600
