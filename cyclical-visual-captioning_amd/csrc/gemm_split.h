// Split-product arithmetic shared by the GEMM kernels (gemm_skinny.hip, gemm_packed.hip) and the switch that
// selects it (cvc_gemm_packed_split in include/cvc_hip.h).
#pragma once
#include "cvc_common.h"

// 0 = fp32 MFMA; 1 / 2 = products as exact 3-way bf16 splits on the bf16 MFMA, the packed kernel then running
// 4 / 8 waves per workgroup; the row-major ring kernel uses the split for any non-zero mode.  Defined in
// gemm_skinny.hip.
extern int cvc_gemm_split_mode;

namespace {

// ---- fp32 products on the bf16 matrix pipe (16x the fp32 MFMA rate), without giving up fp32 accuracy:
// every fp32 operand is split EXACTLY into three bf16 terms, v = hi + mid + lo (8 + 8 + 8 mantissa bits, by
// truncation, so both remainders are exact fp32 subtractions), and a product w*x is taken as the six leading
// cross terms  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid  (each exact in the fp32 accumulator's
// product width); the three dropped terms are below 2^-23 |w x|, i.e. at the level of one fp32 rounding.
// 6 bf16 MFMAs (32 cycles each, K = 16) replace 8 fp32 MFMAs (64 cycles each, K = 2): 0.375x the matrix time,
// paid for with ~4.5 VALU ops per operand element for the split.  Finite operands only: an infinite input splits
// into (inf, nan, nan) and yields NaN where the fp32 MFMA would return inf.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
struct Split3 { u32x4 hi, mid, lo; };       // 8 bf16 each: element e of quad a -> slot e, of quad b -> slot 4 + e

__device__ __forceinline__ Split3 split8(const f32x4 a, const f32x4 b) {
    Split3 r;
    const f32x2 v[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const u32x2 u0 = __builtin_bit_cast(u32x2, v[p]);
        const f32x2 r1 = v[p] - __builtin_bit_cast(f32x2, u0 & 0xffff0000u);
        const u32x2 u1 = __builtin_bit_cast(u32x2, r1);
        const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, u1 & 0xffff0000u);
        const u32x2 u2 = __builtin_bit_cast(u32x2, r2);
        r.hi[p] = __builtin_amdgcn_perm(u0.y, u0.x, 0x07060302u);     // {top 16 bits of .y, top 16 bits of .x}
        r.mid[p] = __builtin_amdgcn_perm(u1.y, u1.x, 0x07060302u);
        r.lo[p] = __builtin_amdgcn_perm(u2.y, u2.x, 0x07060302u);
    }
    return r;
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace
