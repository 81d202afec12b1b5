// Matrix-core fragment types and the f32 -> 2 x f16 split shared by the scan kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// MODE: how a row group is multiplied
//   0  f32 rows, v_mfma_f32_16x16x4_f32 (exact f32: bit-for-bit a k-ordered fmaf chain)
//   1  bf16 rows, v_mfma_f32_16x16x32_bf16 (candidate generator of the batched path)
//   2  f32 rows split on the fly into two f16 pieces (x = hi + lo); three v_mfma_f32_16x16x32_f16
//      (hi*hi + hi*lo + lo*hi) replace eight f32 MFMAs; error bound in DESIGN.md section 3
//   3  rows already stored as f16 hi/lo pieces in the positions mode 2 loads them from (msr_presplit_rows): mode 2
//      without the split instructions, at the price of a second copy of the matrix
constexpr int MODE_F32 = 0, MODE_BF16 = 1, MODE_F16X2 = 2, MODE_PRE = 3;

// x (8 floats in two float4) -> hi, lo with x ~= hi + lo.  cvt_pkrtz rounds toward zero, so the residual x - hi is
// exact in f32 and smaller than 2^-10 |x|; after the second truncation |x - hi - lo| < 2^-20 |x|.
__device__ __forceinline__ void split_f16(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x[j], x[j + 1]));
        const f16x2 l = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x[j] - (float)h[0], x[j + 1] - (float)h[1]));
        hi[j] = h[0]; hi[j + 1] = h[1];
        lo[j] = l[0]; lo[j + 1] = l[1];
    }
}

// x (8 floats) -> 8 f16, round to nearest even (v_cvt_f16_f32).  The single-product mode of msr_gemm_f32.hip; its error
// is MEASURED with f16_err2() below, element by element with this same conversion.
__device__ __forceinline__ f16x8 cvt_f16_rtn(const f32x4& a, const f32x4& b) {
    f16x8 h;
    h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
    h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
    return h;
}
// squared error of one element under cvt_f16_rtn, conservative about the matrix cores' treatment of f16 subnormals: a
// value that converts to a subnormal (or zero) is counted as lost entirely.
__device__ __forceinline__ float f16_err2(float x) {
    const float h = (float)(_Float16)x;
    const float d = __builtin_fabsf(h) < 6.103515625e-05f ? x : h - x;
    return d * d;
}
