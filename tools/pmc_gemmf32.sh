#!/bin/bash
# PMC passes over the f32 GEMM scan: usage tools/pmc_gemmf32.sh <tag> [gemmf32_check args]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcA_$TAG -- python3 $R/tools/gemmf32_check.py --iters 2 "$@" > $R/gpurun_out/pmcA_$TAG.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmcB_$TAG -- python3 $R/tools/gemmf32_check.py --iters 2 "$@" > $R/gpurun_out/pmcB_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcF_$TAG -- python3 $R/tools/gemmf32_check.py --iters 2 "$@" > $R/gpurun_out/pmcF_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcW_$TAG -- python3 $R/tools/gemmf32_check.py --iters 2 "$@" > $R/gpurun_out/pmcW_$TAG.log 2>&1
