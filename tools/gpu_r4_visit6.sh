#!/bin/bash
# round-4 visit 6: the three tests of visit 5 whose bars were edited + the kernel's neighbours, with durations
timeout 400 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_resblock.py tests/test_gpu_determinism.py -m gpu -q -s --durations=8 2>&1 | grep -E "split-bf16|passed|failed|Error|error|^[0-9.]+s " | tail -30
