#!/bin/bash
# round 6, visit 1: the driver's exact bench command with the compact line; what power / clock telemetry the box offers
mkdir -p gpurun_out
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6v1_bench_n1.json 2> gpurun_out/r6v1_bench_n1.err; echo "bench rc=$?"
echo "stdout lines: $(wc -l < gpurun_out/r6v1_bench_n1.json) bytes: $(wc -c < gpurun_out/r6v1_bench_n1.json)"
cat gpurun_out/r6v1_bench_n1.json
ls -la gpurun_out/bench_detail.json
( which amd-smi rocm-smi; amd-smi version; timeout 20 amd-smi metric --power --clock 2>&1 | head -60; timeout 20 amd-smi static --limit 2>&1 | head -40; timeout 20 rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 ) > gpurun_out/r6v1_smi_probe.txt 2>&1
tail -5 gpurun_out/r6v1_bench_n1.err
