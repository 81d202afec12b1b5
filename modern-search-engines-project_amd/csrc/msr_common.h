// Shared device helpers for libmsretr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSR_WAVE 64

// Order-preserving maps float -> unsigned (larger float <=> larger unsigned), and back.
__host__ __device__ __forceinline__ uint32_t msr_ord32(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if (f == 0.0f) u = 0;                    // -0.0 == 0.0 in the reference's sort: one key (ties go by index)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float msr_unord32(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ __forceinline__ uint64_t msr_ord64(double d) {
    uint64_t u;
    __builtin_memcpy(&u, &d, 8);
    if (d == 0.0) u = 0;                     // (as above)
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ __forceinline__ double msr_unord64(uint64_t u) {
    u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}

// A score takes part in selection only if it is > -inf (this also rejects NaN).
__device__ __forceinline__ bool msr_valid(float s) { return s > -__builtin_inff(); }
__device__ __forceinline__ bool msr_valid(double s) { return s > -__builtin_inf(); }
