#!/bin/bash
# round 6, visit 4: the two-image form of TE_CONV_3X3W6 (wino6q_kernel): bit identity with forms 0 / 1, timing at the four large shapes
mkdir -p gpurun_out
( FORM2=1 timeout 600 python tools/wino6_ab.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v4_wino6_ab.log; echo "rc=$?"
cat gpurun_out/r6v4_wino6_ab.log | cut -c1-400
