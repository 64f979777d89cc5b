set -x
mkdir -p gpurun_out/wp
timeout 200 python tools/wgrad_pair_time.py > gpurun_out/wp/pair2.log 2>&1
cat gpurun_out/wp/pair2.log
timeout 600 python -m pytest tests/test_gpu_conv_fuzz.py tests/test_gpu_ops.py -m gpu -x -q -k "wgrad or modconv or modulated or fuzz" 2>&1 | tail -3
