#!/bin/bash
# round-5 visit 1: the ping-pong form of the split-bf16 Winograd kernel - bit identity with the block-phase form, timing of the
# product library and of the tuning variants, the Winograd GPU tests through the new form, a quick A/B of the whole iteration
TAG=r5v1
mkdir -p gpurun_out
export TMPDIR=/tmp
( SMALL=1 timeout 150 python tools/wino6_ab.py ) > gpurun_out/${TAG}_w6ab_small.log 2>&1; rc=$?; echo "small A/B rc=$rc"; tail -3 gpurun_out/${TAG}_w6ab_small.log
if [ $rc -ne 0 ]; then echo "small run failed: stopping"; exit 1; fi
( timeout 300 python tools/wino6_ab.py ) > gpurun_out/${TAG}_w6ab_product.log 2>&1; echo "product A/B rc=$?"; grep -E "TF/s|MISMATCH" gpurun_out/${TAG}_w6ab_product.log
for v in prio ilv1 ilv6 split1 split3 nocommit nomfma; do
  ( ONLY_BIG=1 VARIANT=w6p_$v timeout 200 python tools/wino6_ab.py ) > gpurun_out/${TAG}_w6ab_$v.log 2>&1; echo "variant $v rc=$?"; grep -E "^B16.*TF/s" gpurun_out/${TAG}_w6ab_$v.log | sed 's/.*| block/block/'
done
( timeout 600 python -m pytest tests/test_gpu_winograd.py -m gpu -q --no-header -p no:cacheprovider -x ) > gpurun_out/${TAG}_pytest_winograd.log 2>&1; echo "pytest winograd rc=$?"; tail -2 gpurun_out/${TAG}_pytest_winograd.log
( TE_W6_FORM=1 timeout 400 python bench.py --gpus 1 --steps 16 --warmup 3 --no-sub --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_bench_form1.json 2> gpurun_out/${TAG}_bench_form1.err; echo "bench form1 rc=$?"
( TE_W6_FORM=0 timeout 400 python bench.py --gpus 1 --steps 16 --warmup 3 --no-sub --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_bench_form0.json 2> gpurun_out/${TAG}_bench_form0.err; echo "bench form0 rc=$?"
python - <<PY
import json
for f in ("form1","form0"):
    try:
        d=json.loads(open(f"gpurun_out/${TAG}_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], {k:round(v,1) for k,v in d["roofline"].get("per_kernel",{}).items()} if isinstance(d["roofline"].get("per_kernel"),dict) else "")
    except Exception as e: print(f, "ERR", e)
PY
