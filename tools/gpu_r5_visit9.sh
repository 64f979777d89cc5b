#!/bin/bash
# phase profile of s2s6 / t2s6 and its timing decomposition (experiment switches ST_*)
mkdir -p gpurun_out
for v in st_prof st_nomfma st_nodsw st_nodma st_nofetch st_noarith; do
  echo "=== $v"; timeout 200 python tools/s2s6_phase_prof.py $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r5v9_st_prof.log 2>&1
grep -E "^===|^s2s6|^t2s6" gpurun_out/r5v9_st_prof.log
