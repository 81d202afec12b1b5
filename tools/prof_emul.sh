#!/bin/bash
# Kernel table of the emulated per-rank step: usage tools/prof_emul.sh <tag> <N>   (on the GPU box; writes gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; N=$2; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
A="--emulate-ranks $N --docs $((1000000 / N)) --chunks $((5000000 / N)) --queries-per-step $((256 * N)) --no-cpu-baseline --no-variants"
python3 $R/bench.py $A > $O/${T}_emul_$N.json 2> $O/${T}_emul_$N.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_emul -- python3 $R/bench.py --steps 6 --warmup 2 $A > $O/${T}_emul_${N}_under_rocprof.json 2> $O/${T}_prof_emul.err
cd $R
python3 tools/summarize_prof.py stats $(ls $O/${T}_prof_emul/*/*kernel_stats.csv) $(ls $O/${T}_prof_emul/*/*kernel_trace.csv) $O/${T}_kernel_stats_emul_$N.md
rm -rf $O/${T}_prof_emul
