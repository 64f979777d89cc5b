#!/bin/bash
# round-3 visit: batched attention projections - targeted tests, then the metric + generator workloads
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generator.py tests/test_gpu_train_step.py tests/test_gpu_timed_second_order.py -m gpu -x -q > gpurun_out/v3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v3_pytest.log
tail -5 gpurun_out/v3_pytest.log
