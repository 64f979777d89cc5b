#!/bin/bash
# One GPU-box visit: diagnostics, parity tests, bench, rocprof.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/gpu_diag.py ) > gpurun_out/diag.log 2>&1
( timeout 600 python tools/gpu_grad_probe.py ) > gpurun_out/gradprobe.log 2>&1
echo "diag rc=$?" 
( timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_EXTRA} ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"
( timeout 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 ) > gpurun_out/bench.log 2>&1
echo "bench rc=$?"
tail -2 gpurun_out/bench.log
if [ "${DO_PROF:-1}" = "1" ]; then
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?"
  find gpurun_out/prof -name "*stats*" | head
fi
