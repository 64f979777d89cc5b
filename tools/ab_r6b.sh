# one-off visit: T2 edge kernel check (tests, kernel durations under rocprofv3)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_t2s6.py -x -q -m gpu 2>&1 | tail -3
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t2 -o t2 -- python $R/tools/t2s6_check.py ) > $R/gpurun_out/r6b_t2s6_check_prof.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/prof_t2/t2_results.db > gpurun_out/r6b_t2_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_t2
grep -E "split" gpurun_out/r6b_t2s6_check_prof.log | cut -c1-190 | head -9
grep -E "t2_edge|t2s6|TOTAL" gpurun_out/r6b_t2_kernel_stats.txt | cut -c1-60,100-175
