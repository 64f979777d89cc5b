#!/bin/bash
# round-3 visit: full GPU test tier, FIR XCD-order A/B, quick bench, kernel traces of the path step and the train step
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider ) > gpurun_out/r3c_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3c_pytest_gpu.log
( timeout 120 python tools/exp_time.py firxcd0 fir; timeout 120 python tools/exp_time.py product fir ) > gpurun_out/r3c_fir_ab.log 2>&1; cat gpurun_out/r3c_fir_ab.log
( timeout 300 python bench.py --no-cpu-baseline --no-sub ) > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err; echo "bench rc=$?"
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_path -o path -- python $R/tools/reg_steps_trace.py path ) > $R/gpurun_out/r3c_rocprof_path.log 2>&1; echo "rocprof path rc=$?"
( timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-sub --no-kernel-timing ) > $R/gpurun_out/r3c_rocprof_train.log 2>&1; echo "rocprof train rc=$?"
cd $R
for t in path train; do python tools/rocpd_stats.py gpurun_out/prof_$t/${t}_results.db > gpurun_out/r3c_${t}_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_$t; done
grep "ms per step" gpurun_out/r3c_rocprof_path.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['substeps'])
PY
