#!/bin/bash
# Round profile set: usage tools/prof_round.sh <tag>   (on the GPU box; writes gpurun_out/<tag>_*)
# 1. plain bench lines (default; dense bf16 1024 queries), 2. rocprofv3 --kernel-trace --stats of the same commands,
# 3. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section).
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-variants"
D="--workload dense --dense-mode bf16 --queries-per-step 1024"
python3 $R/bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
python3 $R/bench.py $D --no-cpu-baseline > $O/${T}_bench_dense_bf16_q1024.json 2> $O/${T}_bench_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $R/bench.py --steps 6 --warmup 2 $B > $O/${T}_bench_under_rocprof.json 2> $O/${T}_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_bf16 -- python3 $R/bench.py --steps 6 --warmup 2 $D $B > $O/${T}_bench_bf16_under_rocprof.json 2> $O/${T}_prof_bf16.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${T}_pmc_$C -- python3 $R/bench.py --steps 3 --warmup 1 $B > /dev/null 2> $O/${T}_pmc_$C.err
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${T}_pmcbf16_$C -- python3 $R/bench.py --steps 3 --warmup 1 $D $B > /dev/null 2> $O/${T}_pmcbf16_$C.err
done
cd $R
for S in "" _bf16; do
  python3 tools/summarize_prof.py stats $(ls $O/${T}_prof$S/*/*kernel_stats.csv) $(ls $O/${T}_prof$S/*/*kernel_trace.csv) $O/${T}_kernel_stats$S.md
done
python3 tools/summarize_prof.py pmc $(ls $O/${T}_pmc_FETCH_SIZE/*/*counter_collection.csv) $(ls $O/${T}_pmc_WRITE_SIZE/*/*counter_collection.csv) $O/${T}_hbm_pmc.json
python3 tools/summarize_prof.py pmc $(ls $O/${T}_pmcbf16_FETCH_SIZE/*/*counter_collection.csv) $(ls $O/${T}_pmcbf16_WRITE_SIZE/*/*counter_collection.csv) $O/${T}_hbm_pmc_bf16.json
rm -rf $O/${T}_prof $O/${T}_prof_bf16 $O/${T}_pmc_FETCH_SIZE $O/${T}_pmc_WRITE_SIZE $O/${T}_pmcbf16_FETCH_SIZE $O/${T}_pmcbf16_WRITE_SIZE
ls -la $O | grep ${T}_
