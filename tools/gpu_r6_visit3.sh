#!/bin/bash
# round 6, visit 3: census of the framework ops left in the timed iteration (sorted by device time)
mkdir -p gpurun_out
( timeout 600 python tools/step_census.py 16 time ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6v3_step_census.txt; echo "census rc=$?"
head -100 gpurun_out/r6v3_step_census.txt | cut -c1-330
