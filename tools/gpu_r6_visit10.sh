#!/bin/bash
# round 6, visit 10: p1s6 with the double-buffered weight image (one barrier per phase): tests + timing
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_p1s6.py -q --no-header -p no:cacheprovider ) > gpurun_out/r6v10_pytest_p1s6.log 2>&1; echo "pytest p1s6 rc=$?"; tail -3 gpurun_out/r6v10_pytest_p1s6.log
( timeout 300 python tools/p1s6_check.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v10_p1s6_check.log; cat gpurun_out/r6v10_p1s6_check.log | cut -c1-220
