#!/usr/bin/env python3
"""Static check of the hand-driven row rings of msr_gemm_f32.hip (no GPU needed): compiles the file to gfx950 assembly and
reports, per streaming kernel, registers, scratch operations, s_waitcnt vmcnt(0) between the first counted wait and the last
block-end wait (there must be none: a spill reload or a flat load inside the row pipeline drains it), and the destination
registers of the inline-asm row loads -- every ring register must be the destination of exactly three loads (prologue, same
tile, next tile).  A copy of a ring slot that lives across the slot's reload makes the register allocator rename slots with
moves of registers whose loads are still in flight; this table is how that shows.

    python tools/ring_check.py            (exit code 1 if a kernel fails a check)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "modern-search-engines-project_amd", "csrc", "msr_gemm_f32.hip")
out = os.path.join(tempfile.mkdtemp(), "f32.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(SRC), "-S", "--cuda-device-only", "-o", out, SRC],
               check=True, stderr=subprocess.DEVNULL)
text = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if re.match(r"^_ZN.*gemm_stream\w*kernel.*:\s", l)]
bad = False
for i, name in starts:
    end = next(j for j in range(i, len(text)) if text[j].startswith(".Lfunc_end"))
    body = text[i:end]
    meta = "\n".join(text[end:end + 80])
    vgpr = int(re.search(r"NumVgprs: (\d+)", meta).group(1))
    scratch = sum("scratch_" in l for l in body)
    waits = [k for k, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\((8|12|16|20|28)\)", l)]
    inside0 = [k for k, l in enumerate(body) if "s_waitcnt vmcnt(0)" in l and waits and waits[0] < k < waits[-1]]
    dests = collections.Counter(re.findall(r"global_load_dwordx4 (v\[\d+:\d+\])", "\n".join(body)))
    # (the 128-query kernel's loop is unrolled over 8 steps and the allocator rotates its slots through the body: the
    # three-loads criterion is the 256-query kernel's)
    ring_ok = all(c == 3 for c in dests.values()) or "stream256" not in name
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    ok = scratch == 0 and not inside0 and ring_ok
    bad |= not ok
    print(f"{'ok  ' if ok else 'FAIL'} {demangled[:70]:70s} vgprs {vgpr:3d}  scratch ops {scratch:2d}  vmcnt(0) inside the pipeline "
          f"{len(inside0)}  row-load destinations {4 * len(dests)} registers"
          + (" (each the destination of exactly 3 loads)" if "stream256" in name and ring_ok else "" if ring_ok else f" {dict(dests)}"))
sys.exit(1 if bad else 0)
