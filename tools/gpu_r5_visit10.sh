#!/bin/bash
# fetch issued by the multiplying role (wino6p, s2s6, t2s6): correctness, A/B timing, phase profiles
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_s2s6.py tests/test_gpu_t2s6.py -q --no-header -p no:cacheprovider -x ) > gpurun_out/r5v10_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5v10_tests.log
( timeout 200 python tools/wino6_ab.py; timeout 200 python tools/s2s6_check.py; timeout 200 python tools/t2s6_check.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r5v10_checks.log
grep -E "MISMATCH|BAD|bad|B16|B32" gpurun_out/r5v10_checks.log | cut -c1-260
( timeout 200 python tools/s2s6_phase_prof.py st_prof; timeout 200 python tools/w6p_phase_prof.py w6p_prof ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r5v10_prof.log
cat gpurun_out/r5v10_prof.log
