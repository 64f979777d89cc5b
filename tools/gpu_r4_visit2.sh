#!/bin/bash
# round-4 visit 2: closed modulated-conv family (correctness + path-step A/B), wgrad XCD order A/B (time + fetch counters)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generator.py tests/test_gpu_timed_second_order.py tests/test_gpu_train_step.py tests/test_gpu_optim.py tests/test_gpu_determinism.py tests/test_gpu_conv_fuzz.py -m gpu -q --no-header -p no:cacheprovider -rA --maxfail=25 ) > gpurun_out/r4b_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r4b_pytest.log | tail -3; grep -E "pinned|FAILED|ERROR" gpurun_out/r4b_pytest.log | head -40
for v in "" oldcomposite; do timeout 200 python tools/reg_steps_trace.py path $v 2>&1 | grep "ms per step" | sed "s/^/[$v] /"; done
python tools/exp_build.py wxcd0 -DTE_WGRAD_XCD=0 > gpurun_out/r4b_build.log 2>&1; tail -1 gpurun_out/r4b_build.log
( timeout 200 python tools/exp_time.py wxcd0; timeout 200 python tools/exp_time.py product ) 2>&1 | grep -E " W3X3| WT2" > gpurun_out/r4b_wgrad_xcd_ab.log; cat gpurun_out/r4b_wgrad_xcd_ab.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_r4b -o fetch --output-format csv -- python $R/tools/kernel_once.py > $R/gpurun_out/r4b_pmc.log 2>&1
cd $R
python - <<'PY'
import csv, collections
d=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc_r4b/fetch_counter_collection.csv')):
    if 'wgrad_mfma' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        d[r['Kernel_Name'][:40]].append(2*float(r['Counter_Value'])*1024/1e6)
for k,v in d.items(): print('FETCH x2 MB', k, [round(x) for x in v])
PY
( timeout 300 python bench.py --no-cpu-baseline --no-pmc ) > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['frac'], r['whole_step_frac']); print(d['substeps'])
print({k:(round(v['tflops'],1)) for k,v in r['per_kernel'].items()})
print({k:(v['value'], v['roofline']['whole_step_frac']) for k,v in d['sub_benchmarks'].items()})
PY
