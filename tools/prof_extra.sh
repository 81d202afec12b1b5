#!/bin/bash
# The round's other bench lines: usage tools/prof_extra.sh <tag>   (on the GPU box; writes gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-variants"
python3 $R/bench.py --queries-per-step 512 $B > $O/${T}_bench_q512.json 2> $O/${T}_bench_q512.err
python3 $R/bench.py --queries-per-step 1024 $B > $O/${T}_bench_q1024.json 2> $O/${T}_bench_q1024.err
python3 $R/bench.py --dense-mode bf16 --queries-per-step 1024 $B > $O/${T}_bench_bf16cand_q1024.json 2> $O/${T}_bench_bf16cand_q1024.err
python3 $R/bench.py --workload bm25 --docs 100000 --chunks 0 --terms 200000 --queries-per-step 1024 --k1 100 --no-cpu-baseline > $O/${T}_bench_bm25_100k.json 2> $O/${T}_bench_bm25_100k.err
python3 $R/bench.py --workload bm25 --queries-per-step 256 --no-cpu-baseline > $O/${T}_bench_bm25_1m.json 2> $O/${T}_bench_bm25_1m.err
for N in 2 4 8; do
  python3 $R/bench.py --emulate-ranks $N --docs $((1000000 / N)) --chunks $((5000000 / N)) --queries-per-step $((256 * N)) $B > $O/${T}_emul_$N.json 2> $O/${T}_emul_$N.err
done
python3 $R/tools/encoder_bench.py > $O/${T}_encoder_bench.json 2> $O/${T}_encoder_bench.err
ls -la $O | grep ${T}_
