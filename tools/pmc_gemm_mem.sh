#!/bin/bash
# HBM / L2 counters of the GEMM candidate kernels: usage tools/pmc_gemm_mem.sh <tag> [gemm_check args]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcF_$TAG -- python3 $R/tools/gemm_check.py --iters 2 --check 16 "$@" > $R/gpurun_out/pmcF_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcW_$TAG -- python3 $R/tools/gemm_check.py --iters 2 --check 16 "$@" > $R/gpurun_out/pmcW_$TAG.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcR_$TAG -- python3 $R/tools/gemm_check.py --iters 2 --check 16 "$@" > $R/gpurun_out/pmcR_$TAG.log 2>&1
