#!/bin/bash
# PMC passes over the BM25 TAAT kernel: usage tools/pmc_bm25.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcA_$TAG -- python3 $R/tools/bm25_check.py --iters 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmcB_$TAG -- python3 $R/tools/bm25_check.py --iters 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcC_$TAG -- python3 $R/tools/bm25_check.py --iters 3 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for s in "ABC":
    for f in glob.glob("$R/gpurun_out/pmc%s_$TAG/*/*counter_collection.csv" % s):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "bm25_taat" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 8192 * 64:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()): print(s, k, "%.4g" % (sum(v)/len(v)), len(v))
PY
