#!/bin/bash
# tools/prof_step.sh <tag> [extra bench flags]: rocprofv3 kernel trace of a short default bench run -> gpurun_out/<tag>_kernel_stats.md
# (per-kernel table + one timed step) and the raw trace of the timed steps (gpurun_out/<tag>_trace.csv) for timeline analysis.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; shift; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants "$@" > $O/${T}_bench_under_rocprof.json 2> $O/${T}_prof.err
cd $R
python3 tools/summarize_prof.py stats $(ls $O/${T}_prof/*/*kernel_stats.csv) $(ls $O/${T}_prof/*/*kernel_trace.csv) $O/${T}_kernel_stats.md
python3 - "$(ls $O/${T}_prof/*/*kernel_trace.csv)" $O/${T}_trace.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
starts = [i for i, r in enumerate(rows) if "bm25_taat" in r["Kernel_Name"]]
longest = max(dur(rows[i]) for i in starts)
big = [i for i in starts if dur(rows[i]) > 0.5 * longest]          # the batched (timed) steps
a, b = big[len(big) // 2], big[min(len(big) - 1, len(big) // 2 + 2)]   # two consecutive timed steps
keep = rows[a:b]
t0 = int(keep[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,end_us,queue,kernel\n")
    for r in keep:
        f.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - t0) / 1e3:.1f},{r.get("Queue_Id", "")},"{r["Kernel_Name"][:90]}"\n')
PY
rm -rf $O/${T}_prof
