#!/bin/bash
# Launches of several 256-query groups (the f16 image of the rows): usage tools/prof_groups.sh <tag> <queries per step>
# (on the GPU box; writes gpurun_out/<tag>_*): kernel table of the step + FETCH_SIZE / WRITE_SIZE in separate --pmc passes.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; Q=$2; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
A="--queries-per-step $Q --no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_q$Q -- python3 $R/bench.py --steps 6 --warmup 2 $A > $O/${T}_bench_q${Q}_under_rocprof.json 2> $O/${T}_prof_q$Q.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${T}_pmcq${Q}_$C -- python3 $R/bench.py --steps 3 --warmup 1 $A > /dev/null 2> $O/${T}_pmcq${Q}_$C.err
done
cd $R
python3 tools/summarize_prof.py stats $(ls $O/${T}_prof_q$Q/*/*kernel_stats.csv) $(ls $O/${T}_prof_q$Q/*/*kernel_trace.csv) $O/${T}_kernel_stats_q$Q.md
python3 tools/summarize_prof.py pmc $(ls $O/${T}_pmcq${Q}_FETCH_SIZE/*/*counter_collection.csv) $(ls $O/${T}_pmcq${Q}_WRITE_SIZE/*/*counter_collection.csv) $O/${T}_hbm_pmc_q$Q.json
rm -rf $O/${T}_prof_q$Q $O/${T}_pmcq${Q}_FETCH_SIZE $O/${T}_pmcq${Q}_WRITE_SIZE
ls -la $O | grep ${T}_ | grep q$Q
