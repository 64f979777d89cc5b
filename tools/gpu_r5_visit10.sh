timeout 120 python tools/t2s6_check.py 2>&1 | grep -v amdgpu | grep "^B16\|^B32\|swap\|FAIL"
echo "== side stream off"; TE_T2_SIDE_STREAM=0 timeout 120 python tools/t2s6_check.py 2>&1 | grep "^B16\|^B32\|FAIL"
timeout 300 python -m pytest tests/test_gpu_t2s6.py tests/test_gpu_inference.py tests/test_gpu_determinism.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
for f in 1 0 1 0; do TE_T2_SIDE_STREAM=$f timeout 300 python bench.py --gpus 1 --steps 16 --warmup 3 --no-sub --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side=$f', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['per_kernel']['convT2']['tflops'],1))"; done
