#!/bin/bash
# round-5 visit 3: ping-pong form with the staging arithmetic inside the multiplying wave's own MFMA stream
( timeout 300 python tools/wino6_ab.py ) > gpurun_out/r5v3_w6ab_product.log 2>&1; echo "product A/B rc=$?"; grep -E "TF/s|MISMATCH|Error|error" gpurun_out/r5v3_w6ab_product.log | cut -c1-260
( timeout 100 python tools/w6p_phase_prof.py w6p_prof ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5v3_phase_prof.log
echo "== nocommit"; ONLY_BIG=1 VARIANT=w6p_nocommit timeout 100 python tools/wino6_ab.py 2>&1 | grep "^B16" | sed 's/.*| block/block/'
