#!/bin/bash
# PMC passes over the dense scan kernel: usage tools/pmc_scan.sh <variant> <queries> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$1; Q=$2; TAG=$3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcA_$TAG -- python3 $R/tools/scan_once.py --variant $V --queries $Q > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcB_$TAG -- python3 $R/tools/scan_once.py --variant $V --queries $Q > /dev/null 2>&1
