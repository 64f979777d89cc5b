#!/bin/bash
# round 6, visit 14: 16-column images in the split Winograd kernel (two samples side by side): tests, A/B of the iteration
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_generator.py tests/test_gpu_resblock.py tests/test_gpu_train_step.py tests/test_gpu_timed_shapes.py tests/test_gpu_determinism.py -q --no-header -p no:cacheprovider ) > gpurun_out/r6v14_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r6v14_pytest.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r6v14_pytest.log | head
for i in 1 2; do for f in 0 1; do
  TE_SPLIT_W16=$f timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r6v14_bench_w16_$f.$i.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r6v14_bench_w16_$f.$i.json').read().strip().splitlines()[-1]); print('TE_SPLIT_W16=$f run $i:', d['value'], d['ms_per_step'], d['substeps'])"
done; done
python - <<'PY'
import math, sys, torch
sys.path.insert(0, '.')
from transeditor_amd import _lib
from tools.exp_time import timeit
for B in (32, 16, 8):
    x = torch.randn(B, 512, 16, 16, device='cuda'); w = torch.randn(512, 512, 3, 3, device='cuda') / 68
    u6 = _lib.conv_pack(w, _lib.PACK_W6FWD); ud = _lib.conv_pack(w, _lib.PACK_FWD)
    f6 = lambda: _lib.conv(x, u6, _lib.CONV_3X3W6, 512, 16, 16)
    fd = lambda: _lib.conv(x, ud, _lib.CONV_3X3, 512, 16, 16)
    fl = 2.0 * 9 * 512 * 512 * 256 * B
    t6, td = timeit(f6, n=20), timeit(fd, n=20)
    print(f'3x3 512->512 @16x16 B{B}: split {t6*1e3:.1f} us {fl/t6/1e9:.1f} TF/s, direct fp32 {td*1e3:.1f} us {fl/td/1e9:.1f} TF/s', flush=True)
PY
