#!/bin/bash
# round 6, visit 2: power / clock telemetry under each matrix-pipe kernel (product, MFMA-only and staging-only builds of wino6p)
mkdir -p gpurun_out
( timeout 300 python tools/power_probe.py ) > gpurun_out/r6v2_power_product.txt 2>&1; echo "product rc=$?"
( VARIANT=mfma_only ONLY=wino6p timeout 200 python tools/power_probe.py ) > gpurun_out/r6v2_power_mfma_only.txt 2>&1; echo "mfma_only rc=$?"
( VARIANT=stage_only ONLY=wino6p timeout 200 python tools/power_probe.py ) > gpurun_out/r6v2_power_stage_only.txt 2>&1; echo "stage_only rc=$?"
grep -v amdgpu.ids gpurun_out/r6v2_power_product.txt | cut -c1-330
grep -v amdgpu.ids gpurun_out/r6v2_power_mfma_only.txt | cut -c1-330
grep -v amdgpu.ids gpurun_out/r6v2_power_stage_only.txt | cut -c1-330
