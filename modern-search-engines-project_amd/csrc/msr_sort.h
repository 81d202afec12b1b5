// Bitonic sort of records (64-bit key, 32-bit key extension[, 32-bit value]) held in LDS, by one workgroup (gfx950).
// Used by the select's final kernel (msr_topk.hip) and the two sorts of the fuse kernel (msr_rerank.hip).
//
// The usual network (levels kk = 2, 4, .. P; stages j = kk / 2 .. 1), one compare-exchange per thread and stage.  Between two
// stages with partner distance <= 64 no workgroup barrier is needed: thread t's pair of such a stage lies in the 128-record
// block of the 64 consecutive pair indices its wave holds (the same block in every such stage), and a wave's LDS operations
// execute in order -- only the compiler has to be kept from moving LDS accesses across the stage boundary.
// Measured (tools/sel_final_phases.py, BM25 top-1000 of 1 M documents): 1024 records = 55 stages = 37 k cycles = 18 us of the
// final kernel's 24; ~45 instructions per compare-exchange of a 96-bit key (index arithmetic, four LDS reads, the compare,
// four predicated writes), two waves per SIMD at work: bound by instruction issue and latency, not by LDS bandwidth or barriers
// (two workgroups sharing a CU take 44 us each instead of 24).  Two
// stages per pass on four records per thread (one index computation, four compare-exchanges in registers) was built and
// measured at the same 18 us -- the compiler turns the in-register exchanges into branches and moves, 159 instructions per
// pass and wave, with half as many waves at work -- and dropped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msr_sort {

__device__ __forceinline__ void stage_sync(int j, int j_next) {
    if (j <= 64 && j_next <= 64) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// P: power of two >= 2, at most what the arrays hold.  ascending: final order (else descending) by (khi, klo); equal keys stay
// where they are.  Every thread of the workgroup must call it; the caller has a barrier between its last write to the arrays
// and the call; the arrays are complete for every thread on return.
template <int THREADS, bool HAS_VAL>
__device__ void bitonic_sort(uint64_t* khi, uint32_t* klo, uint32_t* val, int P, bool ascending) {
    static_assert(THREADS % 64 == 0, "whole waves");
    for (int kk = 2; kk <= P; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int idx = threadIdx.x; idx < (P >> 1); idx += THREADS) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int p = i | j;
                const bool want_desc = ((i & kk) == 0) != ascending;    // the larger record belongs at i
                const uint64_t ah = khi[i], bh = khi[p];
                const uint32_t al = klo[i], bl = klo[p];
                const bool a_lt_b = ah < bh || (ah == bh && al < bl);
                const bool b_lt_a = bh < ah || (bh == ah && bl < al);
                if (want_desc ? a_lt_b : b_lt_a) {
                    khi[i] = bh; klo[i] = bl; khi[p] = ah; klo[p] = al;
                    if (HAS_VAL) { const uint32_t t = val[i]; val[i] = val[p]; val[p] = t; }
                }
            }
            stage_sync(j, j > 1 ? j >> 1 : kk);                  // (the stage after j = 1 is the next level's first: j = kk)
        }
    }
    __syncthreads();
}

}  // namespace msr_sort
