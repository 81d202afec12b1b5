#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/, scratch) into the small summaries kept under profiles/.

    python tools/summarize_prof.py stats  <kernel_stats.csv> <kernel_trace.csv> <out.md>
    python tools/summarize_prof.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB and
are collected in separate passes; on gfx950 FETCH_SIZE counts exactly half of a wide (16 B/lane) streaming
read, so read bytes = 2 * FETCH_SIZE * 1024 for the dense scan (other access widths are uncalibrated and are
reported raw).
"""
import collections
import csv
import json
import re
import sys


def short(name):
    if name.startswith("_Z"):                       # a name the profiler could not demangle (f16 vector arguments)
        m = re.search(r"\d+([a-z]\w*?_kernel)", name)
        if m:
            return m.group(1)
    m = re.search(r"(\w+_kernel\w*)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def stats(stats_csv, trace_csv, out):
    rows = list(csv.DictReader(open(stats_csv)))
    ours = [r for r in rows if any(k in r["Name"] for k in ("dense_scan", "dense_ksplit", "build_qimage", "thr_compact", "rescore", "bm25_taat", "sel_", "rerank_", "best_chunk",
                                                             "prep_queries", "merge_kernel", "interleave", "row_inv_norm",
                                                             "fill_chunk_doc", "gemm_", "build_qimg", "qmat_kernel", "batch_margin",
                                                             "unit_bf16", "pad_inv", "rescore", "f16_", "build_qimg1"))]
    lines = ["| kernel | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|"]
    for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                     f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.3f} |")
    tr = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    # a step starts at its BM25 kernel; timed (batched) steps are the ones with the long BM25 launches
    starts = [i for i, r in enumerate(tr) if "bm25_taat" in r["Kernel_Name"]]
    step = []
    if len(starts) >= 4:
        longest = max(dur(tr[i]) for i in starts)
        big = [j for j, i in enumerate(starts[:-1]) if dur(tr[i]) > 0.5 * longest]
        j = big[len(big) // 2]
        a, b = starts[j], starts[j + 1]
        agg = collections.OrderedDict()
        for r in tr[a:b]:
            n = short(r["Kernel_Name"]); d = dur(r)
            agg.setdefault(n, [0, 0]); agg[n][0] += 1; agg[n][1] += d
        wall = (int(tr[b]["Start_Timestamp"]) - int(tr[a]["Start_Timestamp"])) / 1e3
        step = [f"", f"One timed step (dispatch {a}..{b}): wall {wall:.1f} us, {b - a} dispatches", "",
                "| kernel | dispatches | us |", "|---|---|---|"]
        for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
            step.append(f"| {n} | {c} | {d / 1e3:.1f} |")
    open(out, "w").write("\n".join(lines + step) + "\n")


def pmc(fetch_csv, write_csv, out):
    res = collections.defaultdict(dict)
    for key, f in (("FETCH_SIZE_KiB", fetch_csv), ("WRITE_SIZE_KiB", write_csv)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for n, v in agg.items():
            if any(k in n for k in ("dense_scan", "dense_ksplit", "gemm_stream", "gemm_kernel", "bm25_taat", "sel_", "rerank_", "rescore")):
                res[n][key] = {"per_launch_max": max(v), "per_launch_mean": sum(v) / len(v), "launches": len(v)}
    for n, d in res.items():
        if any(k in n for k in ("dense_scan", "dense_ksplit", "gemm_stream", "gemm_kernel")) and "FETCH_SIZE_KiB" in d and "WRITE_SIZE_KiB" in d:
            d["hbm_bytes_per_launch"] = 2 * d["FETCH_SIZE_KiB"]["per_launch_max"] * 1024 + d["WRITE_SIZE_KiB"]["per_launch_max"] * 1024
            d["note"] = "read bytes = 2 x FETCH_SIZE (gfx950 wide-load correction), write bytes = WRITE_SIZE"
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
