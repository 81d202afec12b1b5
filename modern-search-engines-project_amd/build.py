"""Build recipe for libmsretr.so (HIP, gfx950 only) -- plain hipcc, no cmake.

    python -m msretr.build            # or: from msretr.build import build_library; build_library()

The shared library is written next to the sources (csrc/libmsretr.so) so that it travels with the tree.
`--diag` builds csrc/libmsretr_diag.so with -DMSR_DIAG instead: the only build that reads the MSR_* environment
knobs of the A/B tools (tools/ab_scan.py, tools/scan_once.py); the product library reads no environment variable.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmsretr.so")
ARCH = "gfx950"

# translation unit -> extra flags.  The float64 paths must not fuse a*b+c (Python does not).
UNITS = {
    "msr_engine.hip": [],
    "msr_topk.hip": [],
    "msr_bm25.hip": ["-ffp-contract=off"],
    "msr_dense.hip": [],
    "msr_dense_ks.hip": [],
    "msr_rerank.hip": ["-ffp-contract=off"],
    "msr_batch.hip": [],
    "msr_enc_linear.hip": [],
    "msr_gemm.hip": [],
    "msr_gemm_f32.hip": [],
    "msr_build.hip": [],
    "msr_encoder.hip": ["-ffp-contract=off"],
    "msr_format.cpp": [],             # host-only C++ (result-line formatter)
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, save_temps=False, diag=False):
    hipcc = _hipcc()
    lib = os.path.join(CSRC, "libmsretr_diag.so") if diag else LIB
    suffix = ".diag.o" if diag else ".o"
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "msretr.h"))
    headers.append(os.path.join(HERE, "..", "include", "msretr_encoder.h"))
    objs, jobs = [], []
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + suffix)
        objs.append(o)
        if force or _stale(o, [s] + headers):
            common = COMMON if src.endswith(".hip") else [f for f in COMMON if not f.startswith("--offload-arch")]
            cmd = [hipcc] + common + extra + (["-DMSR_DIAG"] if diag else []) + ["-c", s, "-o", o]
            if save_temps:
                cmd.insert(1, "-save-temps=obj")
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    if jobs:                                                # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(run, jobs))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, save_temps="--save-temps" in sys.argv,
                        diag="--diag" in sys.argv))
