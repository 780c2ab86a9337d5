#!/bin/bash
# PMC passes for the GEMM kernel (counters only with --kernel-trace, separate passes as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_gemm1 -- python3 $R/tools/pmc_gemm.py > $R/gpurun_out/pmc_gemm1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_gemm2 -- python3 $R/tools/pmc_gemm.py > $R/gpurun_out/pmc_gemm2.log 2>&1
ls $R/gpurun_out/pmc_gemm1/*
