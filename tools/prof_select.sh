#!/bin/bash
# tools/prof_select.sh <tag> [lib path]: tools/select_bench.py plain and under rocprofv3 -> gpurun_out/<tag>_select.json, <tag>_select_kernels.md
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out
[ -n "$2" ] && export MSR_LIB_PATH=$2
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/select_bench.py > $O/${T}_select.json 2> $O/${T}_select.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_sel -- python3 $R/tools/select_bench.py --iters 10 > /dev/null 2> $O/${T}_prof_sel.err
cd $R
python3 tools/summarize_prof.py stats $(ls $O/${T}_prof_sel/*/*kernel_stats.csv) $(ls $O/${T}_prof_sel/*/*kernel_trace.csv) $O/${T}_select_kernels.md
rm -rf $O/${T}_prof_sel
cat $O/${T}_select.json
