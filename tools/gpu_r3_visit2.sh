#!/bin/bash
# round-3 visit: tests touched by the attention-stack removal, PMC passes (HBM traffic of the FIR kernels after the XCD-aware tile order)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_inference.py tests/test_gpu_fir_fuzz.py -m gpu -q -x --no-header -p no:cacheprovider ) > gpurun_out/r3f_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3f_tests.log
rm -rf gpurun_out/pmc
bash tools/pmc_round.sh > gpurun_out/r3f_pmc_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r3f > gpurun_out/r3f_pmc_summary_stdout.txt 2>&1
rm -f gpurun_out/pmc/*.db gpurun_out/pmc/*.csv
cat gpurun_out/r3f_pmc_summary_stdout.txt
