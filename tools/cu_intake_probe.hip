// tools/cu_intake_probe.hip -- what can ONE compute unit of an MI355X take in per second, by source and by loads in flight?
//
// Both streaming kernels of this repository (gemm_stream256_kernel over f32 rows and over the bf16 image) stop at about
// 29-34 GB/s per CU of combined intake (rows from HBM or L2 + the query image from L2), whatever is multiplied behind it
// (DESIGN.md section 3).  This probe measures the ceiling itself with nothing behind the loads: one 512-thread workgroup
// per CU (the kernels' shape), every wave streaming 1 KB per load instruction (16 B per lane, whole cache lines), D loads
// in flight per wave, from
//     hbm   a buffer far larger than every cache, each byte read once          (the rows)
//     hot   a 384 KB region every workgroup reads again and again             (the query image: L2 hits)
//     mix   two hbm loads for every hot load                                   (the 256-query pass' 786 KB : 384 KB)
//     dma   the hot region through LDS-DMA (global_load_lds, 16 B per lane)    (how the image actually travels)
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_intake_probe tools/cu_intake_probe.hip && /tmp/cu_intake_probe
// Prints one JSON object per configuration: GB/s per CU and chip-wide.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void glb_void;
typedef __attribute__((address_space(3))) void lds_void;

// mode 0: hbm, 1: hot, 2: mix (2 hbm : 1 hot)
template <int D, int MODE>
__global__ __launch_bounds__(512) void probe_kernel(const u32x4* __restrict__ src, size_t vec_per_wave, int steps,
                                                    const u32x4* __restrict__ hot, uint32_t hot_vecs, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 8 + wv;
    const u32x4* p = src + wave * vec_per_wave + lane;
    uint32_t hpos = (uint32_t)((wave * 977u) % (hot_vecs / 64)) * 64u + lane;       // waves start at different lines of the region
    u32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
        u32x4 v[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const bool from_hot = MODE == 1 || (MODE == 2 && (j % 3) == 2);
            if (from_hot) {
                v[j] = hot[hpos];
                hpos += 64; if (hpos >= hot_vecs) hpos -= hot_vecs;
            } else {
                v[j] = *p; p += 64;
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;                  // (keeps the loads alive)
}

// LDS-DMA of the hot region: every wave moves D x 1 KB per step into its own 16 KB of LDS, then waits
template <int D>
__global__ __launch_bounds__(512) void probe_dma_kernel(const char* __restrict__ hot, uint32_t hot_bytes, int steps,
                                                        uint32_t* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 8 + wv;
    uint32_t pos = (uint32_t)((wave * 977u) % (hot_bytes / 1024)) * 1024u;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void*)(hot + pos + lane * 16), (lds_void*)(smem + wv * 16384 + (j & 15) * 1024), 16, 0, 0);
            pos += 1024; if (pos >= hot_bytes) pos -= hot_bytes;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (smem[threadIdx.x] == 0x7f && steps < 0) sink[0] = 1;
}

static int n_cus() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    return p.multiProcessorCount;
}

template <int D, int MODE>
static void run(const char* name, const u32x4* src, size_t src_bytes, const u32x4* hot, uint32_t hot_bytes, uint32_t* sink, int cus) {
    const size_t waves = (size_t)cus * 8;
    size_t vec_per_wave = src_bytes / 16 / waves / 64 * 64;
    int steps = (int)(vec_per_wave / 64 / D);
    if (MODE == 1) steps = 8192 / D;                                             // 8 MB per wave from the hot region
    if (MODE == 2) steps = (int)(vec_per_wave / 64 / ((D * 2 + 2) / 3));        // (an upper bound of the hbm loads per step: stay inside the buffer)
    if (steps > 16384 / D) steps = 16384 / D;                                   // <= 16 MB per wave: a few ms per launch
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        probe_kernel<D, MODE><<<cus, 512>>>(src, vec_per_wave, steps, hot, hot_bytes / 16, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)waves * steps * D * 1024.0;
    printf("{\"source\": \"%s\", \"loads_in_flight_per_wave\": %d, \"KB_in_flight_per_CU\": %d, \"ms\": %.4f, \"GBps_per_CU\": %.2f, \"TBps_chip\": %.3f}\n",
           name, D, D * 8, best, bytes / best / 1e6 / cus, bytes / best / 1e9);
    fflush(stdout);
}

template <int D>
static void run_dma(const char* hot, uint32_t hot_bytes, uint32_t* sink, int cus) {
    const int steps = 8192 / D;
    CHECK(hipFuncSetAttribute((const void*)probe_dma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        probe_dma_kernel<D><<<cus, 512, 8 * 16384>>>(hot, hot_bytes, steps, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)cus * 8 * steps * D * 1024.0;
    printf("{\"source\": \"dma_hot\", \"loads_in_flight_per_wave\": %d, \"KB_in_flight_per_CU\": %d, \"ms\": %.4f, \"GBps_per_CU\": %.2f, \"TBps_chip\": %.3f}\n",
           D, D * 8, best, bytes / best / 1e6 / cus, bytes / best / 1e9);
    fflush(stdout);
}

int main() {
    const int cus = n_cus();
    const size_t src_bytes = (size_t)24 << 30;
    const uint32_t hot_bytes = 384 * 1024;
    u32x4 *src, *hot; uint32_t* sink;
    CHECK(hipMalloc((void**)&src, src_bytes)); CHECK(hipMalloc((void**)&hot, hot_bytes)); CHECK(hipMalloc((void**)&sink, 4));
    CHECK(hipMemset(src, 1, src_bytes)); CHECK(hipMemset(hot, 2, hot_bytes)); CHECK(hipMemset(sink, 0, 4));
    CHECK(hipDeviceSynchronize());
    printf("{\"cus\": %d, \"workgroup\": \"512 threads, one per CU\", \"load\": \"16 B per lane = 1 KB per wave instruction\"}\n", cus);
    run<2, 0>("hbm", src, src_bytes, hot, hot_bytes, sink, cus);  run<4, 0>("hbm", src, src_bytes, hot, hot_bytes, sink, cus);
    run<8, 0>("hbm", src, src_bytes, hot, hot_bytes, sink, cus);  run<16, 0>("hbm", src, src_bytes, hot, hot_bytes, sink, cus);
    run<24, 0>("hbm", src, src_bytes, hot, hot_bytes, sink, cus);
    run<2, 1>("hot", src, src_bytes, hot, hot_bytes, sink, cus);  run<4, 1>("hot", src, src_bytes, hot, hot_bytes, sink, cus);
    run<8, 1>("hot", src, src_bytes, hot, hot_bytes, sink, cus);  run<16, 1>("hot", src, src_bytes, hot, hot_bytes, sink, cus);
    run<24, 1>("hot", src, src_bytes, hot, hot_bytes, sink, cus);
    run<6, 2>("mix_2hbm_1hot", src, src_bytes, hot, hot_bytes, sink, cus);  run<12, 2>("mix_2hbm_1hot", src, src_bytes, hot, hot_bytes, sink, cus);
    run<24, 2>("mix_2hbm_1hot", src, src_bytes, hot, hot_bytes, sink, cus);
    run_dma<2>((const char*)hot, hot_bytes, sink, cus); run_dma<4>((const char*)hot, hot_bytes, sink, cus);
    run_dma<8>((const char*)hot, hot_bytes, sink, cus); run_dma<16>((const char*)hot, hot_bytes, sink, cus);
    return 0;
}
