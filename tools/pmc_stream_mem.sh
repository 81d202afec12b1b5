#!/bin/bash
# Memory-path counters of the streaming pass inside bench.py: usage tools/pmc_stream_mem.sh <tag> [bench args]
# (latency from the L1's point of view, read level at the L2 <-> fabric interface, TLB misses; separate passes)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; shift; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="--steps 3 --warmup 1 --no-cpu-baseline --no-variants"
i=0
for C in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  # (a pass with TA_ADDR_STALLED_BY_TC_CYCLES / TA_DATA_STALLED_BY_TC_CYCLES / TA_TA_BUSY aborted rocprofv3 on this pool: left out)
  i=$((i+1))
  echo "pass $i: $C"
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${T}_m$i -- python3 $R/bench.py $B "$@" > /dev/null 2> $O/${T}_m$i.err
done
python3 - <<PY
import csv,glob,collections,json
out={}
for d in sorted(glob.glob("$O/${T}_m*/*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k=r["Kernel_Name"]
        if "gemm_stream" not in k and "rerank_cos" not in k and "dense_ksplit_kernel<2" not in k: continue
        agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in agg.items():
        for n,v in c.items(): out.setdefault(k,{})[n]=sum(v)/len(v)
json.dump(out,open("$O/${T}_stream_mem_pmc.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
rm -rf $O/${T}_m1 $O/${T}_m2 $O/${T}_m3 $O/${T}_m4
