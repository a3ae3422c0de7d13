// Shared device helpers for the cvc_hip kernels (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cvc_hip.h"
#include "../../include/cvc_hip_blocks.h"
#include "../../include/cvc_hip_experimental.h"

#define CVC_WAVE 64
#define CVC_MIN_VALUE (-1e8f)   // model/modules.py:22,98

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

static inline int cvc_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

// ---- cross-lane reductions over the 64-lane wave on the DPP network (no LDS crossbar round trips: a __shfl_xor
// butterfly is six dependent ds_bpermute_b32, ~100 cycles each; these are six full-rate VALU ops).  Steps: quad (xor 1, xor 2),
// 8 lanes (row_half_mirror), 16 lanes (row_mirror) -- every lane of a 16-lane row then holds its row's value -- then
// row_bcast15 into rows 1 / 3 and row_bcast31 into rows 2 / 3: lane 63 holds the wave's value, read back as a scalar.
// The order of the additions is fixed, so sums are deterministic run to run.
#define CVC_DPP(v, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xF, false))
#define CVC_DPP_KEEP(v, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xF, false))
__device__ __forceinline__ float wave_sum(float v) {
    v += CVC_DPP(v, 0xB1, 0xF);        // quad_perm [1,0,3,2]
    v += CVC_DPP(v, 0x4E, 0xF);        // quad_perm [2,3,0,1]
    v += CVC_DPP(v, 0x141, 0xF);       // row_half_mirror
    v += CVC_DPP(v, 0x140, 0xF);       // row_mirror
    v += CVC_DPP(v, 0x142, 0xA);       // row_bcast15 -> rows 1, 3 (rows 0, 2 add 0)
    v += CVC_DPP(v, 0x143, 0xC);       // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, CVC_DPP_KEEP(v, 0xB1, 0xF));
    v = fmaxf(v, CVC_DPP_KEEP(v, 0x4E, 0xF));
    v = fmaxf(v, CVC_DPP_KEEP(v, 0x141, 0xF));
    v = fmaxf(v, CVC_DPP_KEEP(v, 0x140, 0xF));
    v = fmaxf(v, CVC_DPP_KEEP(v, 0x142, 0xA));   // lanes outside the row mask keep their own value (max with itself)
    v = fmaxf(v, CVC_DPP_KEEP(v, 0x143, 0xC));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- transcendental helpers on the hardware v_exp_f32 / v_rcp_f32 (1 ulp each)
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// tanh(x) = 1 - 2 / (1 + e^{2x}); saturates correctly at +-inf, abs err ~1e-7
__device__ __forceinline__ float fast_tanh(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f * fast_rcp(1.0f + e);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
