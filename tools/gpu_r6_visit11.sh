#!/bin/bash
# round 6, visit 11: current phase profile of the strided / transposed split kernels
mkdir -p gpurun_out
( timeout 300 python tools/s2s6_phase_prof.py st_prof ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v11_st_phase_profile.log; cat gpurun_out/r6v11_st_phase_profile.log | cut -c1-230
