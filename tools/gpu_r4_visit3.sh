#!/bin/bash
# round-4 visit 3: FFHQ-1024 generator bench, product (XCD-ordered wgrad grid) vs wxcd0 (3-D grid order), interleaved on one box
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -k "modulated_conv2d_module_golden" 2>&1 | tail -2
python tools/exp_build.py wxcd0 -DTE_WGRAD_XCD=0 > gpurun_out/r4c_build.log 2>&1; tail -1 gpurun_out/r4c_build.log
for i in 1 2; do for v in product wxcd0; do
  timeout 200 python tools/bench_with_lib.py $v --workload generator --size 1024 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', round(d['value'],1), round(d['ms_per_step'],2), round(r['kernel_time_share'],3), {k:round(v['tflops'],1) for k,v in r['per_kernel'].items()})"
done; done 2>&1 | tee gpurun_out/r4c_1024_ab.log
for v in product wxcd0; do
  timeout 200 python tools/bench_with_lib.py $v --workload generator --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('256 $v', round(d['value'],1), round(d['ms_per_step'],2), round(r['kernel_time_share'],3), {k:round(v['tflops'],1) for k,v in r['per_kernel'].items()})"
done 2>&1 | tee -a gpurun_out/r4c_1024_ab.log
