#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_p1s6.py -q --no-header -p no:cacheprovider -rA ) > gpurun_out/r6v9_pytest_p1s6.log 2>&1; echo "pytest p1s6 rc=$?"; grep -E "passed|failed" gpurun_out/r6v9_pytest_p1s6.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r6v9_pytest_p1s6.log | head; grep "split-bf16 1x1" gpurun_out/r6v9_pytest_p1s6.log | head -14 | cut -c1-200
