#!/bin/bash
# round 6, visit 8: the 1x1 split-bf16 kernel (p1s6): tests, timing against the fp32 kernel, ResBlock tests, iteration A/B
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_p1s6.py -q --no-header -p no:cacheprovider -rA ) > gpurun_out/r6v8_pytest_p1s6.log 2>&1; echo "pytest p1s6 rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r6v8_pytest_p1s6.log | tail -5; grep "split-bf16 1x1" gpurun_out/r6v8_pytest_p1s6.log | head -8 | cut -c1-200
( timeout 300 python tools/p1s6_check.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v8_p1s6_check.log; cat gpurun_out/r6v8_p1s6_check.log | cut -c1-220
( timeout 900 python -m pytest tests/test_gpu_resblock.py tests/test_gpu_train_step.py -q --no-header -p no:cacheprovider -x ) > gpurun_out/r6v8_pytest_resblock.log 2>&1; echo "pytest resblock rc=$?"; tail -3 gpurun_out/r6v8_pytest_resblock.log
for i in 1 2; do for f in 0 1; do
  TE_SPLIT_1X1=$f timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r6v8_bench_1x1_$f.$i.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r6v8_bench_1x1_$f.$i.json').read().strip().splitlines()[-1]); print('TE_SPLIT_1X1=$f run $i:', d['value'], d['ms_per_step'], d['substeps'])"
done; done
