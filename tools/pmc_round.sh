#!/bin/bash
# PMC passes (counters only, no tracing domains besides kernel-trace) over tools/kernel_once.py
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $R/gpurun_out/pmc -o mfma --output-format csv -- python $R/tools/kernel_once.py > $R/gpurun_out/pmc/mfma.log 2>&1
echo "pmc mfma rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc -o fetch --output-format csv -- python $R/tools/kernel_once.py > $R/gpurun_out/pmc/fetch.log 2>&1
echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc -o write --output-format csv -- python $R/tools/kernel_once.py > $R/gpurun_out/pmc/write.log 2>&1
echo "pmc write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc -o lds --output-format csv -- python $R/tools/kernel_once.py > $R/gpurun_out/pmc/lds.log 2>&1
echo "pmc lds rc=$?"
ls -la $R/gpurun_out/pmc | head -30
