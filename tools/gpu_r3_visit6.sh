#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_step.py tests/test_gpu_timed_second_order.py tests/test_gpu_generator.py tests/test_gpu_resblock.py -m gpu -q > gpurun_out/v6_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v6_pytest.log
tail -6 gpurun_out/v6_pytest.log
