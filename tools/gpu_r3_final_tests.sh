#!/bin/bash
# full GPU tier + smoke at HEAD (the log that goes with the round's evidence)
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ) > gpurun_out/r3f_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r3f_pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/r3f_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3f_smoke.log
