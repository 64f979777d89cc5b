#!/bin/bash
# round-4 visit 7: kernel traces + PMC passes at HEAD (split-bf16 Winograd kernel in the step)
TAG=r4
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-sub --no-kernel-timing --no-pmc ) > $R/gpurun_out/${TAG}_rocprof_train.log 2>&1; echo "rocprof train rc=$?"
( timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gen -o gen -- python $R/bench.py --workload generator --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pmc ) > $R/gpurun_out/${TAG}_rocprof_gen.log 2>&1; echo "rocprof gen rc=$?"
cd $R
for t in train gen; do python tools/rocpd_stats.py gpurun_out/prof_$t/${t}_results.db > gpurun_out/${TAG}_${t}_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_$t; done
rm -rf gpurun_out/pmc
bash tools/pmc_round.sh > gpurun_out/${TAG}_pmc_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/${TAG} > gpurun_out/${TAG}_pmc_summary_stdout.txt 2>&1
rm -f gpurun_out/pmc/*.db gpurun_out/pmc/*.csv
head -8 gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-60,112-160; grep TOTAL gpurun_out/${TAG}_train_kernel_stats.txt; grep value gpurun_out/${TAG}_rocprof_train.log | cut -c1-200 | tail -1
head -5 gpurun_out/${TAG}_pmc_summary.txt
