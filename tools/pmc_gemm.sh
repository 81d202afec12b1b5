#!/bin/bash
# PMC passes over the GEMM candidate kernels: usage tools/pmc_gemm.sh <tag> [gemm_check args]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcA_$TAG -- python3 $R/tools/gemm_check.py --iters 2 --check 16 "$@" > $R/gpurun_out/pmcA_$TAG.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmcB_$TAG -- python3 $R/tools/gemm_check.py --iters 2 --check 16 "$@" > $R/gpurun_out/pmcB_$TAG.log 2>&1
