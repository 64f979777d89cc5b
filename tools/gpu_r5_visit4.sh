#!/bin/bash
( timeout 300 python tools/wino6_ab.py ) > gpurun_out/r5v4_w6ab_product.log 2>&1; echo "product A/B rc=$?"; grep -E "^B16|MISMATCH|Error|error" gpurun_out/r5v4_w6ab_product.log | cut -c1-260
for v in p1 p3 slot0; do echo "== $v"; ONLY_BIG=1 VARIANT=w6p_$v timeout 100 python tools/wino6_ab.py 2>&1 | grep "^B16" | sed 's/.*| block/block/'; done
for v in prof prof_p1 prof_p3; do echo "== $v"; ( timeout 100 python tools/w6p_phase_prof.py w6p_$v ) 2>&1 | grep -v amdgpu.ids | head -20; done
