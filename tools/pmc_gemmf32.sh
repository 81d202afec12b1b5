#!/bin/bash
# PMC passes over the streaming pass (msr_gemm_f32.hip): usage tools/pmc_gemmf32.sh <tag> [gemmf32_check args, e.g. --queries 256]
# Prints per-kernel averages of the counters for the emit passes (gemm_stream*_kernel<true>).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
A="--docs 1000000 --chunks 5000000 --iters 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcA_$TAG -- python3 $R/tools/gemmf32_check.py $A "$@" > $R/gpurun_out/pmcA_$TAG.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmcB_$TAG -- python3 $R/tools/gemmf32_check.py $A "$@" > $R/gpurun_out/pmcB_$TAG.log 2>&1
if [ -z "$PMC_NO_MEM" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcF_$TAG -- python3 $R/tools/gemmf32_check.py $A "$@" > $R/gpurun_out/pmcF_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcW_$TAG -- python3 $R/tools/gemmf32_check.py $A "$@" > $R/gpurun_out/pmcW_$TAG.log 2>&1
fi
python3 - <<PY
import csv, glob, collections, json
out = {}
for s in "ABFW":
    for f in glob.glob("$R/gpurun_out/pmc%s_$TAG/*/*counter_collection.csv" % s):
        agg = collections.defaultdict(list); dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm_stream" in n and "Lb1" in n or ("gemm_stream" in n and "<true>" in n):
                k = "stream256" if "stream256" in n else "stream128"
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            out.setdefault(k, {})[c] = sum(v) / len(v)
for k, d in out.items():
    if "SQ_WAVE_CYCLES" in d:
        w = d["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in d: d[c + "/WAVE_CYCLES"] = d[c] / w
    if "SQ_LDS_IDX_ACTIVE" in d and d["SQ_LDS_IDX_ACTIVE"]:
        d["LDS_CONFLICT_FRAC"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open("$R/gpurun_out/pmc_$TAG.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $R/gpurun_out/pmcA_$TAG $R/gpurun_out/pmcB_$TAG $R/gpurun_out/pmcF_$TAG $R/gpurun_out/pmcW_$TAG
