// Backward-data GEMM of the skinny layers:  dX[M<=64, N] = dY[M, K] * W[K, N]   (fp32, MFMA 32x32x2).
//
// Replaces `torch.mm(d_gates, w_hh)` / `torch.mm(d_gates, w_ih[:, k0:k1])` in the backward of the two
// LSTM cells (autograd of nn.LSTMCell, reference model/decoder_core.py:45-50, 59-61, 99-108): one launch
// per cell for every input segment that needs a gradient, the weights read once in their checkpoint
// layout.  W is row-major [K, ldw] (K = 4R gate rows), so for this product its rows ARE coalesced along
// N: a lane's 16-byte load W[k][n0+4i .. n0+4i+3] feeds four MFMA column tiles (tile c holds columns
// n0 + 4i + c), and dY arrives in the quad layout [K/4][64][4] (written by cvc_lstm_pointwise_bwd), the
// same fragment-native activation layout the packed forward kernel uses.
//
//   workgroup = 128 output columns x one K slice, 4 waves interleaved over the slice's 8-row groups;
//   per group and wave: 4 weight loads + MT activation loads (16 B / lane) -> 16*MT MFMAs;
//   register ring of DEPTH groups, steady-state loads unconditional (counted vmcnt, see gemm_skinny.hip);
//   ordered cross-wave sum in LDS, K slices written as partial planes, summed in a fixed order by
//   nn_reduce_kernel (bitwise reproducible).
#include "cvc_common.h"
#include "gemm_split.h"

namespace {

constexpr int NN_MAX_SEG = 6;
#ifndef CVC_NN_DEPTH
#define CVC_NN_DEPTH 3
#endif
constexpr int NN_DEPTH = CVC_NN_DEPTH;
#ifndef CVC_NN_WGS
#define CVC_NN_WGS 2     // workgroups per CU the register budget is held to
#endif

struct NNSeg {
    const float* w;     // column 0 of this range inside a row-major [K, ldw] matrix
    float* dst;         // [M, ld_dst]
    int ldw, ncols, ld_dst, slab0;   // slab0: first 128-column slab of the segment in the launch
};

struct NNArgs {
    const float* xq;    // dY, quad layout [K/4][64][4]
    float* part;        // [ksplit][M][ntot] partial planes (unused when ksplit == 1)
    int K, M, nseg, ksplit, ntot, nslab;
    NNSeg seg[NN_MAX_SEG];
};

template <int MT>
struct NNFrag {
    f32x4 w[4];
    f32x4 x[MT];
};

template <int MT>
__global__ __launch_bounds__(256, CVC_NN_WGS) void skinny_gemm_nn_kernel(NNArgs a) {
    constexpr int NW = 4;
    constexpr int LDM = MT * 32 + 1;
    __shared__ float red[2 * 128 * LDM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;

    // 1-D grid, K slice fastest (measured: no difference to slab-fastest on MI355X; kept because the
    // workgroups that share a dY slice then start together)
    const int kslice = (int)blockIdx.x % a.ksplit, slab = (int)blockIdx.x / a.ksplit;
    // which segment does this slab belong to (uniform; a loop, not a select chain, so it stays in SGPRs)
    int s = 0;
    for (int t = 1; t < a.nseg; ++t)
        if (slab >= a.seg[t].slab0) s = t;
    const NNSeg sg = a.seg[s];
    const int n0 = (slab - sg.slab0) * 128;
    // lanes past the segment's last column re-read its last quad (never stored)
    int col = n0 + 4 * i;
    col = col + 4 <= sg.ncols ? col : sg.ncols - 4;

    const int ngroup = a.K >> 3;
    const int g_lo = ngroup * kslice / a.ksplit, g_hi = ngroup * (kslice + 1) / a.ksplit;
    const int ng = g_hi - g_lo;
    const int n_my = ng > wave ? (ng - wave + NW - 1) / NW : 0;      // groups g_lo + wave + 4*j
    const size_t ldw = (size_t)sg.ldw;
    // running per-lane pointers (4 weight rows + the activation quad), advanced by one wave step per load:
    // no per-load index arithmetic, which keeps the kernel under 256 registers -> two workgroups per CU
    const float* wp[4];
    wp[0] = sg.w + (size_t)((g_lo + wave) * 8 + kh * 4) * ldw + col;
#pragma unroll
    for (int e = 1; e < 4; ++e) wp[e] = wp[e - 1] + ldw;
    const float* xp = a.xq + ((size_t)((g_lo + wave) * 2 + kh) * 64 + i) * 4;
    const size_t WSTEP = (size_t)NW * 8 * ldw;
    constexpr size_t XSTEP = (size_t)NW * 2 * 256;

    auto load = [&](NNFrag<MT>& f) __attribute__((always_inline)) {
#if defined(CVC_NN_ABL) && CVC_NN_ABL == 1
        if (xp != a.xq + ((size_t)((g_lo + wave) * 2 + kh) * 64 + i) * 4) { xp += XSTEP; asm volatile("" : "+v"(f.w[0])); return; }   // ablation: MFMA side only
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f.w[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp[e]));
            wp[e] += WSTEP;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) f.x[mt] = ld4(xp + mt * 128);
        xp += XSTEP;
    };

    f32x16 acc[4][MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][mt][r] = 0.f;

    auto mma = [&](const NNFrag<MT>& f) __attribute__((always_inline)) {
#if defined(CVC_NN_ABL) && CVC_NN_ABL == 2
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(f.w[e]));                    // ablation: memory side only
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(f.x[mt]));
        return;
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[c][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[e][c], f.x[mt][e], acc[c][mt], 0, 0, 0);
    };

    NNFrag<MT> ring[NN_DEPTH];
    if (n_my >= NN_DEPTH) {
#pragma unroll
        for (int t = 0; t < NN_DEPTH - 1; ++t) load(ring[t]);
        int j = 0;
        for (; j + 2 * NN_DEPTH - 1 <= n_my; j += NN_DEPTH) {
#pragma unroll
            for (int t = 0; t < NN_DEPTH; ++t) {
                load(ring[(t + NN_DEPTH - 1) % NN_DEPTH]);
                mma(ring[t]);
#pragma unroll
                for (int g = 0; g < 4 + MT; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 16 * MT / (4 + MT), 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    // VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                    // VALU (addresses)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int t = 0; t < NN_DEPTH; ++t) {
            if (j + t + NN_DEPTH - 1 < n_my) load(ring[(t + NN_DEPTH - 1) % NN_DEPTH]);
            if (j + t < n_my) mma(ring[t]);
        }
#pragma unroll
        for (int t = 0; t < NN_DEPTH - 1; ++t)
            if (j + NN_DEPTH + t < n_my) mma(ring[t]);
    } else {
        for (int j = 0; j < n_my; ++j) {
            load(ring[0]);
            mma(ring[0]);
        }
    }

    // ---- cross-wave sum, fixed order (w0 + w2) + (w1 + w3), through two LDS planes A / B
    // (acc reg r -> column-tile row (r&3) + 8*(r>>2) + 4*kh, i.e. output column 4*row + c; lane&31 -> batch row).
    // Reads are issued 16 at a time (one accumulator vector) so that LDS latency is paid once per vector.
    float* const planeA = red;
    float* const planeB = red + 128 * LDM;
    auto put = [&](float* plane) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    plane[(4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + c) * LDM + mt * 32 + i] = acc[c][mt][r];
    };
    auto add = [&](const float* plane) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = plane[(4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + c) * LDM + mt * 32 + i];
                acc[c][mt] += t;
            }
    };
    if (wave == 2) put(planeA);
    if (wave == 3) put(planeB);
    __syncthreads();
    if (wave == 0) add(planeA);
    if (wave == 1) { add(planeB); put(planeB); }
    __syncthreads();
    if (wave == 0) { add(planeB); put(planeA); }
    __syncthreads();

    const int M = a.M;
    float* out;
    size_t ld;
    int cbase;
    if (a.ksplit > 1) {
        out = a.part + (size_t)kslice * M * a.ntot;
        ld = (size_t)a.ntot;
        cbase = sg.slab0 * 128 + n0;          // plane columns are slab-padded: segment s starts at slab0 * 128
    } else {
        out = sg.dst;
        ld = (size_t)sg.ld_dst;
        cbase = n0;
    }
    const int nvalid = sg.ncols - n0 < 128 ? sg.ncols - n0 : 128;
    for (int u = tid; u < 128 * MT * 32; u += NW * 64) {
        const int nl = u & 127, m = u >> 7;
        if (m >= M || nl >= nvalid) continue;
        out[(size_t)m * ld + cbase + nl] = red[nl * LDM + m];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Split-product variant (cvc_gemm_packed_split != 0, the default): the same tiling, but two consecutive 8-row groups of a wave
// are taken together as one K = 16 step of v_mfma_f32_32x32x16_bf16 -- each fp32 operand split exactly into three bf16 terms,
// six cross terms per product (gemm_split.h) -- 48 bf16 MFMAs (1536 matrix cycles) instead of 128 fp32 MFMAs (8192) per 16 rows
// at MT = 2.  The fp32 kernel above is matrix-bound (MFMA-only 66 us of its 75 us at the cfg3 lang cell); this one is bound by
// the weight stream.  Register budget: 128 accumulators + a ring of 16-row groups (48 registers each) -> one workgroup per CU
// (4 waves, up to 512 registers each), ring depth 4 to cover HBM latency with a single wave per SIMD.
// k-slot map of a double group (groups ga, gb of this wave): slot s < 4 <-> row 8 ga + 4 kh + s, slot 4 + s <-> row 8 gb + 4 kh + s,
// the same for the weight rows and for the dY quads, so any permutation of k is harmless.
#ifndef CVC_NNS_DEPTH
#define CVC_NNS_DEPTH 4
#endif

template <int MT>
struct NNFrag2 {
    f32x4 w[8];
    f32x4 x[MT][2];
};

template <int MT>
__global__ __launch_bounds__(256, 1) void skinny_gemm_nn_split_kernel(NNArgs a) {
    constexpr int NW = 4;
    constexpr int LDM = MT * 32 + 1;
    constexpr int D = CVC_NNS_DEPTH;
    __shared__ float red[2 * 128 * LDM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int kslice = (int)blockIdx.x % a.ksplit, slab = (int)blockIdx.x / a.ksplit;
    int s = 0;
    for (int t = 1; t < a.nseg; ++t)
        if (slab >= a.seg[t].slab0) s = t;
    const NNSeg sg = a.seg[s];
    const int n0 = (slab - sg.slab0) * 128;
    int col = n0 + 4 * i;
    col = col + 4 <= sg.ncols ? col : sg.ncols - 4;

    const int ngroup = a.K >> 3;
    const int g_lo = ngroup * kslice / a.ksplit, g_hi = ngroup * (kslice + 1) / a.ksplit;
    const int ng = g_hi - g_lo;
    const int n_my = ng > wave ? (ng - wave + NW - 1) / NW : 0;      // groups g_lo + wave + 4*j
    const int n2 = n_my >> 1;                                         // double groups; an odd last group runs on the fp32 MFMA
    const size_t ldw = (size_t)sg.ldw;
    const float* wp[4];
    wp[0] = sg.w + (size_t)((g_lo + wave) * 8 + kh * 4) * ldw + col;
#pragma unroll
    for (int e = 1; e < 4; ++e) wp[e] = wp[e - 1] + ldw;
    const float* xp = a.xq + ((size_t)((g_lo + wave) * 2 + kh) * 64 + i) * 4;
    const size_t WSTEP = (size_t)NW * 8 * ldw;
    constexpr size_t XSTEP = (size_t)NW * 2 * 256;

    auto load2 = [&](NNFrag2<MT>& f) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f.w[h * 4 + e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp[e]));
                wp[e] += WSTEP;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) f.x[mt][h] = ld4(xp + mt * 128);
            xp += XSTEP;
        }
    };

    f32x16 acc[4][MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][mt][r] = 0.f;

    auto mma2 = [&](const NNFrag2<MT>& f) __attribute__((always_inline)) {
        Split3 X[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) X[mt] = split8(f.x[mt][0], f.x[mt][1]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 wa = {f.w[0][c], f.w[1][c], f.w[2][c], f.w[3][c]};
            const f32x4 wb = {f.w[4][c], f.w[5][c], f.w[6][c], f.w[7][c]};
            const Split3 W = split8(wa, wb);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[c][mt] = mfma_bf16(W.mid, X[mt].mid, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.lo, X[mt].hi, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].lo, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.mid, X[mt].hi, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].mid, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].hi, acc[c][mt]);
            }
        }
    };

    NNFrag2<MT> ring[D];
    if (n2 >= D) {
#pragma unroll
        for (int t = 0; t < D - 1; ++t) load2(ring[t]);
        int j = 0;
        for (; j + 2 * D - 1 <= n2; j += D) {
#pragma unroll
            for (int t = 0; t < D; ++t) {
                load2(ring[(t + D - 1) % D]);
                __builtin_amdgcn_sched_barrier(0);            // requests first (the scheduler would sink them behind the MFMAs)
                mma2(ring[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < D; ++t) {
            if (j + t + D - 1 < n2) load2(ring[(t + D - 1) % D]);
            if (j + t < n2) mma2(ring[t]);
        }
#pragma unroll
        for (int t = 0; t < D - 1; ++t)
            if (j + D + t < n2) mma2(ring[t]);
    } else {
        for (int j = 0; j < n2; ++j) {
            load2(ring[0]);
            mma2(ring[0]);
        }
    }
    if (n_my & 1) {                                           // odd last group: exact fp32 products on the fp32 MFMA
        f32x4 w[4], x[MT];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = ld4(wp[e]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) x[mt] = ld4(xp + mt * 128);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[c][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e][c], x[mt][e], acc[c][mt], 0, 0, 0);
    }

    // ---- cross-wave sum and store: identical to skinny_gemm_nn_kernel
    float* const planeA = red;
    float* const planeB = red + 128 * LDM;
    auto put = [&](float* plane) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    plane[(4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + c) * LDM + mt * 32 + i] = acc[c][mt][r];
    };
    auto add = [&](const float* plane) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = plane[(4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + c) * LDM + mt * 32 + i];
                acc[c][mt] += t;
            }
    };
    if (wave == 2) put(planeA);
    if (wave == 3) put(planeB);
    __syncthreads();
    if (wave == 0) add(planeA);
    if (wave == 1) { add(planeB); put(planeB); }
    __syncthreads();
    if (wave == 0) { add(planeB); put(planeA); }
    __syncthreads();

    const int M = a.M;
    float* out;
    size_t ld;
    int cbase;
    if (a.ksplit > 1) {
        out = a.part + (size_t)kslice * M * a.ntot;
        ld = (size_t)a.ntot;
        cbase = sg.slab0 * 128 + n0;
    } else {
        out = sg.dst;
        ld = (size_t)sg.ld_dst;
        cbase = n0;
    }
    const int nvalid = sg.ncols - n0 < 128 ? sg.ncols - n0 : 128;
    for (int u = tid; u < 128 * MT * 32; u += NW * 64) {
        const int nl = u & 127, m = u >> 7;
        if (m >= M || nl >= nvalid) continue;
        out[(size_t)m * ld + cbase + nl] = red[nl * LDM + m];
    }
}

// dst[seg][m][n] = sum over planes, fixed order
__global__ __launch_bounds__(256) void nn_reduce_kernel(NNArgs a) {
    const int m = blockIdx.y;
    const int cq = blockIdx.x * 256 + threadIdx.x;       // float4 column index inside the padded plane
    const int c = cq * 4;
    if (c >= a.ntot) return;
    int s = 0;
    for (int t = 1; t < a.nseg; ++t)
        if (c >= a.seg[t].slab0 * 128) s = t;
    const NNSeg sg = a.seg[s];
    const int n = c - sg.slab0 * 128;
    if (n >= sg.ncols) return;
    const float* p = a.part + (size_t)m * a.ntot + c;
    f32x4 v = ld4(p);
    for (int k = 1; k < a.ksplit; ++k) v += ld4(p + (size_t)k * a.M * a.ntot);
    st4(sg.dst + (size_t)m * sg.ld_dst + n, v);
}

// row-major [M <= 64, K] -> quad layout [K/4][64][4] (rows beyond M zero): the dY operand of cvc_linear_nn_fwd for layers whose
// upstream gradient arrives row-major (nn.Linear backward; the LSTM cells get theirs from cvc_lstm_pointwise_bwd)
__global__ __launch_bounds__(256) void pack_quad_kernel(const float* x, long long ldx, int M, int K, float* xq) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = t & 63, q = t >> 6;
    if (q >= (K >> 2)) return;
    f32x4 v = {0, 0, 0, 0};
    if (m < M) v = ld4(x + (size_t)m * ldx + q * 4);
    st4(xq + ((size_t)q * 64 + m) * 4, v);
}

}  // namespace

extern "C" int cvc_pack_quad(const float* x, long long ldx, int M, int K, float* xq, cvc_stream_t stream) {
    if (!x || !xq || M < 1 || M > 64 || K < 4 || (K & 3) || (ldx & 3) || ((uintptr_t)x & 15)) return CVC_E_BADARG;
    hipLaunchKernelGGL(pack_quad_kernel, dim3((K / 4 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, M, K, xq);
    return cvc_launch_status();
}

static int linear_nn_impl(const float* dy_q, int K, int M, const cvc_nn_seg* segs, int nsegs, int ksplit, float* workspace,
                          bool reduce, cvc_stream_t stream);

extern "C" int cvc_linear_nn_fwd(const float* dy_q, int K, int M, const cvc_nn_seg* segs, int nsegs, int ksplit,
                                 float* workspace, cvc_stream_t stream) {
    return linear_nn_impl(dy_q, K, M, segs, nsegs, ksplit, workspace, true, stream);
}

// The same without the summing launch: for ksplit > 1 the K-slice partial products stay in `workspace` as planes
// [ksplit][M][ntot] (ntot = sum over segments of ceil(ncols / 128) * 128, segment s starting at column 128 * (slabs before it))
// for a consumer that sums them itself (cvc_gru_seq_bwd folds the sum into its next gate-gradient kernel).
extern "C" int cvc_linear_nn_planes_fwd(const float* dy_q, int K, int M, const cvc_nn_seg* segs, int nsegs, int ksplit,
                                        float* workspace, cvc_stream_t stream) {
    return linear_nn_impl(dy_q, K, M, segs, nsegs, ksplit, workspace, false, stream);
}

static int linear_nn_impl(const float* dy_q, int K, int M, const cvc_nn_seg* segs, int nsegs, int ksplit, float* workspace,
                          bool reduce, cvc_stream_t stream) {
    if (!dy_q || !segs || nsegs < 1 || nsegs > NN_MAX_SEG || M < 1 || M > 64 || K < 8 || (K & 7) || ksplit < 1)
        return CVC_E_BADARG;
    if (ksplit > K / 8) ksplit = K / 8;
    NNArgs a{};
    a.xq = dy_q; a.K = K; a.M = M; a.nseg = nsegs; a.ksplit = ksplit; a.part = workspace;
    int slab = 0;
    for (int s = 0; s < nsegs; ++s) {
        const cvc_nn_seg& g = segs[s];
        if (!g.w || !g.dst || g.ncols < 4 || (g.ncols & 3) || (g.ldw & 3) || (g.ld_dst & 3) || g.ldw < g.ncols ||
            g.ld_dst < g.ncols || ((uintptr_t)g.w & 15) || ((uintptr_t)g.dst & 15))
            return CVC_E_BADARG;
        a.seg[s].w = g.w; a.seg[s].dst = g.dst; a.seg[s].ldw = g.ldw; a.seg[s].ncols = g.ncols; a.seg[s].ld_dst = g.ld_dst;
        a.seg[s].slab0 = slab;
        slab += (g.ncols + 127) / 128;
    }
    a.nslab = slab;
    a.ntot = slab * 128;
    if (ksplit > 1 && !workspace) return CVC_E_BADARG;
    const dim3 grid(slab * ksplit);
    if (cvc_gemm_split_mode != 0) {
        if (M <= 32) hipLaunchKernelGGL((skinny_gemm_nn_split_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((skinny_gemm_nn_split_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else if (M <= 32) hipLaunchKernelGGL((skinny_gemm_nn_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((skinny_gemm_nn_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, a);
    if (ksplit > 1 && reduce)
        hipLaunchKernelGGL(nn_reduce_kernel, dim3((a.ntot / 4 + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}
