#!/bin/bash
# round 6, visit 12: t2s6 with the double-buffered weight image: tests, timing, phase profile, iteration
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_t2s6.py tests/test_gpu_generator.py -q --no-header -p no:cacheprovider -x ) > gpurun_out/r6v12_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6v12_pytest.log
( timeout 200 python tools/t2s6_check.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v12_t2s6_check.log; tail -12 gpurun_out/r6v12_t2s6_check.log | cut -c1-220
( timeout 300 python tools/s2s6_phase_prof.py st_prof ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v12_st_phase_profile.log; grep -A18 "^t2s6" gpurun_out/r6v12_st_phase_profile.log | cut -c1-200 | head -44
for i in 1 2; do timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r6v12_bench.$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r6v12_bench.$i.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['substeps'])"; done
