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
    const float* xq2;   // 128-row form only: the second 64-row group's dY (rows 64 .. 64 + M2 - 1 of the product)
    float* part;        // [ksplit][M][ntot] partial planes (unused when ksplit == 1); 128-row form: [ksplit][128][ntot]
    int K, M, M2, nseg, ksplit, ntot, nslab;
    NNSeg seg[NN_MAX_SEG];
};

// ---- cross-wave sum + store, shared by the three kernels below.  The four waves of a workgroup hold partial accumulators of the
// same 128-column x (32 MT)-row tile (they split K).  Round k makes wave k the owner of a quarter of the tile -- the MT (row block,
// column quad group) pairs p = k MT .. k MT + MT - 1, pair p = (mt = p / 4, q = p % 4) -- : every wave writes its registers of that
// quarter to LDS as 16-byte vectors (lane-linear: conflict-free ds_write_b128), one barrier, the owner reads the four copies and
// sums them in the fixed order (w0 + w2) + (w1 + w3).  Rounds alternate between two buffers, so one barrier per round is enough
// (a buffer is rewritten two rounds later, behind the barrier its reader has passed after reading).  Every wave then stores its
// quarter straight from registers: accumulator register 4 q + rr of column tile c is output column 32 q + 16 kh + 4 rr + c of row
// 32 mt + (lane & 31), so the four column tiles give 16 contiguous bytes per (q, rr).
// (Round 4 summed through two [128][rows + 1] planes with 32-bit LDS accesses, three dependent rounds with two or three waves
// idle, and stored one dword per thread and iteration: 13.7 + 9.1 us of an 87 us launch by in-kernel timestamps.)
template <int MT>
__device__ __forceinline__ void nn_cross_wave_store(f32x16 (&acc)[4][MT], char* lds, int wave, int lane, float* out, size_t ld, int cbase,
                                                    int nvalid, int M, int M2) {
    constexpr int BUF = 4 * 4 * MT * 1024;                  // bytes per buffer: 4 waves x (4 MT vectors) x 1 KiB
    const int i = lane & 31, kh = lane >> 5;
    f32x4 fin[MT][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        char* const buf = lds + (k & 1) * BUF;
        char* const mine = buf + wave * (4 * MT * 1024) + lane * 16;
#pragma unroll
        for (int pl = 0; pl < MT; ++pl) {
            const int p = k * MT + pl, mt = p >> 2, q = p & 3;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 v = {acc[c][mt][4 * q], acc[c][mt][4 * q + 1], acc[c][mt][4 * q + 2], acc[c][mt][4 * q + 3]};
                *reinterpret_cast<f32x4*>(mine + (pl * 4 + c) * 1024) = v;
            }
        }
        __syncthreads();
        if (wave == k) {
#pragma unroll
            for (int pl = 0; pl < MT; ++pl)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const char* src = buf + (pl * 4 + c) * 1024 + lane * 16;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(src), s1 = *reinterpret_cast<const f32x4*>(src + 4 * MT * 1024);
                    const f32x4 s2 = *reinterpret_cast<const f32x4*>(src + 2 * 4 * MT * 1024);
                    const f32x4 s3 = *reinterpret_cast<const f32x4*>(src + 3 * 4 * MT * 1024);
                    fin[pl][c] = (s0 + s2) + (s1 + s3);
                }
        }
    }
#pragma unroll
    for (int pl = 0; pl < MT; ++pl) {
        const int p = wave * MT + pl, mt = p >> 2, q = p & 3;
        const int m = mt * 32 + i;
        const bool row_ok = M2 < 0 ? m < M : (m < 64 ? m < M : m - 64 < M2);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int nl = 32 * q + 16 * kh + 4 * rr;
            if (row_ok && nl < nvalid) {
                const f32x4 v = {fin[pl][0][rr], fin[pl][1][rr], fin[pl][2][rr], fin[pl][3][rr]};
                st4(out + (size_t)m * ld + cbase + nl, v);
            }
        }
    }
}

template <int MT>
struct NNFrag {
    f32x4 w[4];
    f32x4 x[MT];
};

template <int MT>
__global__ __launch_bounds__(256, CVC_NN_WGS) void skinny_gemm_nn_kernel(NNArgs a) {
    constexpr int NW = 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * 16 * MT * 1024];       // the two exchange buffers of nn_cross_wave_store
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

    // ---- cross-wave sum in the fixed order (w0 + w2) + (w1 + w3) and store (nn_cross_wave_store above)
    float* out;
    size_t ld;
    int cbase;
    if (a.ksplit > 1) {
        out = a.part + (size_t)kslice * a.M * a.ntot;
        ld = (size_t)a.ntot;
        cbase = sg.slab0 * 128 + n0;          // plane columns are slab-padded: segment s starts at slab0 * 128
    } else {
        out = sg.dst;
        ld = (size_t)sg.ld_dst;
        cbase = n0;
    }
    nn_cross_wave_store<MT>(acc, lds, wave, lane, out, ld, cbase, sg.ncols - n0 < 128 ? sg.ncols - n0 : 128, a.M, -1);
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
#ifndef CVC_NNS_SCHED
#define CVC_NNS_SCHED 1
#endif

template <int MT>
struct NNFrag2 {
    f32x4 w[8];
    f32x4 x[MT][2];
};

template <int MT>
__global__ __launch_bounds__(256, 1) void skinny_gemm_nn_split_kernel(NNArgs a) {
    constexpr int NW = 4;
    constexpr int D = CVC_NNS_DEPTH;
    __shared__ __attribute__((aligned(16))) char lds[2 * 16 * MT * 1024];       // the two exchange buffers of nn_cross_wave_store
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
        // the split of column tile c + 1's weights is issued between the MFMAs of column tile c (a wave issues in order: VALU work
        // behind an MFMA that waits for the matrix pipe waits with it; between two MFMAs it runs in the first one's shadow)
        auto wsplit = [&](int c) __attribute__((always_inline)) {
            const f32x4 wa = {f.w[0][c], f.w[1][c], f.w[2][c], f.w[3][c]};
            const f32x4 wb = {f.w[4][c], f.w[5][c], f.w[6][c], f.w[7][c]};
            return split8(wa, wb);
        };
        Split3 Wn = wsplit(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const Split3 W = Wn;
#if CVC_NNS_SCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (c < 3) Wn = wsplit(c + 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[c][mt] = mfma_bf16(W.mid, X[mt].mid, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.lo, X[mt].hi, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].lo, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.mid, X[mt].hi, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].mid, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].hi, acc[c][mt]);
            }
#if CVC_NNS_SCHED
            if (c < 3) {
#pragma unroll
                for (int k = 0; k < 6 * MT; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, MT == 1 ? 6 : 3, 0);   // its share of the next tile's split
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
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

    // ---- cross-wave sum in the fixed order (w0 + w2) + (w1 + w3) and store (nn_cross_wave_store above)
    float* out;
    size_t ld;
    int cbase;
    if (a.ksplit > 1) {
        out = a.part + (size_t)kslice * a.M * a.ntot;
        ld = (size_t)a.ntot;
        cbase = sg.slab0 * 128 + n0;          // plane columns are slab-padded: segment s starts at slab0 * 128
    } else {
        out = sg.dst;
        ld = (size_t)sg.ld_dst;
        cbase = n0;
    }
    nn_cross_wave_store<MT>(acc, lds, wave, lane, out, ld, cbase, sg.ncols - n0 < 128 ? sg.ncols - n0 : 128, a.M, -1);
}

// ------------------------------------------------------------------------------------------------------------------
// 128-row form of the split-product kernel (round 5): TWO 64-row operand groups (the two loops of the cyclical pass at B = 64 each,
// captioner.py:242-270 and :348-362, which share the LSTM cells) against ONE stream of the weights -- config 3's back-propagation
// then runs 3 backward-data products per step instead of 5.  Same tiling (128 output columns x one K slice per workgroup, 4 waves
// interleaved over the slice's 8-row groups, two groups per K = 16 step, the six cross terms in the same order, the same cross-wave
// sum), so a row's result has the bits the 64-row kernel gives it under the same K split.
// Registers: 4 x 4 accumulator tiles = 256 registers per lane; what is left of the 512 holds a ring of WEIGHT fragments only
// (32 registers per K = 16 step) -- the dY operand of both groups travels through LDS instead: per step and wave 8 KiB =
// [k half of the step][quad row pair][group] x 1 KiB, copied by LDS-DMA (global_load_lds_dwordx4: a quad row of 64 batch rows IS
// 1 KiB contiguous in the quad layout) into a wave-private ring, read back as ds_read_b128.  A step's copies are issued before
// its weight loads, loads return in order, so ONE counted vmcnt covers both.  The ring shares its LDS with the cross-wave sum.
typedef __attribute__((address_space(3))) void* nn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* nn_glb_ptr_t;
#ifndef CVC_NN128_DEPTH
#define CVC_NN128_DEPTH 3
#endif
#ifndef CVC_NN128_SCHED
#define CVC_NN128_SCHED 1
#endif


struct NNW8 { f32x4 w[8]; };

__global__ __launch_bounds__(256, 1) void skinny_gemm_nn_split128_kernel(NNArgs a) {
    constexpr int NW = 4, MT = 4, D = CVC_NN128_DEPTH;
    constexpr int XSLOT = 8 * 1024;                         // bytes per wave and ring slot
    constexpr int RED_BYTES = 2 * 16 * MT * 1024, RING_BYTES = NW * D * XSLOT;      // exchange buffers of nn_cross_wave_store / dY ring
    __shared__ __attribute__((aligned(16))) char lds[RED_BYTES > RING_BYTES ? RED_BYTES : RING_BYTES];
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
    const size_t WSTEP = (size_t)NW * 8 * ldw;
    // LDS-DMA sources: lane l copies batch row l of a quad row; per double group [h][kh][group]
    const float* xs[2] = {a.xq + ((size_t)(g_lo + wave) * 2 * 64 + lane) * 4, a.xq2 + ((size_t)(g_lo + wave) * 2 * 64 + lane) * 4};
    constexpr size_t XSTEP = (size_t)NW * 2 * 256;          // floats between a wave's consecutive groups
    char* const ring = lds + wave * D * XSLOT;

    // Every load of the K loop is issued from inline assembly, and waited for by the counted s_waitcnt statements below: hipcc's own
    // wait-count pass is exact for register loads alone, but with LDS-DMA copies in the loop it falls back to vmcnt(0) at the loop
    // head (one step in D fully drained).  Loads it cannot see only make the waits it computes for its own accesses stricter.
    const unsigned ring_lds = (unsigned)(uintptr_t)(nn_lds_ptr_t)ring;
    auto issue = [&](NNW8& f, int pos) __attribute__((always_inline)) {
        const unsigned dst = ring_lds + pos * XSLOT;
#if defined(CVC_NN128_ABL) && CVC_NN128_ABL == 3          // ablation: no loads (compute side only)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "=v"(f.w[k]) : "s"(dst));
        return;
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                                 :
                                 : "v"(xs[g] + h * XSTEP + q * 256), "s"(dst + ((h * 2 + q) * 2 + g) * 1024)
                                 : "memory", "m0");      // (m0 is declared clobbered: the compiler re-materialises it before any
                                                         // use of its own)
        xs[0] += 2 * XSTEP;
        xs[1] += 2 * XSTEP;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (early-clobber output: the destination never shares registers with the address; the value becomes valid only at
                // the CVC_NN128_WAIT statement it passes through ("+v") -- every read of it sits behind that statement.  What the
                // compiler could still do between the two is MOVE the register: the build refuses scratch / spills for this kernel
                // (build_hip.NO_SCRATCH_KERNELS) and the bit-equality test against the 64-row kernel runs in the default GPU suite)
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(f.w[h * 4 + e]) : "v"(wp[e]));
                wp[e] += WSTEP;
            }
    };

    f32x16 acc[4][MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][mt][r] = 0.f;
#ifdef CVC_NN128_TS            // diagnostic build: s_memrealtime (100 MHz) at entry / K loop done / sums done / stores done, wave 0 of every workgroup
    unsigned long long ts0 = __builtin_readcyclecounter(), ts1 = 0, ts2 = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // byte offset of this lane's first fragment inside a ring slot; it passes through every wait statement below, so the ring reads
    // of a step (addressed from it) cannot be scheduled above the step's wait
    unsigned lane_off = (unsigned)(wave * D * XSLOT + kh * 2048 + i * 16);
    auto compute = [&](const NNW8& f, int pos) __attribute__((always_inline)) {
        const char* src = lds + lane_off + pos * XSLOT;
        Split3 X[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const char* p = src + (mt >> 1) * 1024 + (mt & 1) * 512;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(p), x1 = *reinterpret_cast<const f32x4*>(p + 4096);
#if defined(CVC_NN128_ABL) && CVC_NN128_ABL == 2          // ablation: no split of the dY operand (wrong numbers, same instruction mix otherwise)
            X[mt].hi = __builtin_bit_cast(u32x4, x0); X[mt].mid = __builtin_bit_cast(u32x4, x1); X[mt].lo = X[mt].hi ^ X[mt].mid;
#else
            X[mt] = split8(x0, x1);
#endif
        }
        // The split of column tile c + 1's weights (36 VALU instructions) is issued between the 24 MFMAs of column tile c: a wave
        // issues in order, and an MFMA that finds the matrix pipe busy (32 cycles per MFMA) holds back everything behind it -- VALU
        // work placed between two MFMAs runs in the shadow of the first.  (Carrying the NEXT step's dY split into the last column
        // tile's shadow as well needs a second set of split registers: with it the kernel spills an accumulator tile inside the K
        // loop and a step takes 5 400 clocks instead of 4 960 -- measured, not kept.)
        auto wsplit = [&](int c) __attribute__((always_inline)) {
            const f32x4 wa = {f.w[0][c], f.w[1][c], f.w[2][c], f.w[3][c]};
            const f32x4 wb = {f.w[4][c], f.w[5][c], f.w[6][c], f.w[7][c]};
            return split8(wa, wb);
        };
        Split3 Wn = wsplit(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const Split3 W = Wn;
#if CVC_NN128_SCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (c < 3) Wn = wsplit(c + 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#if defined(CVC_NN128_ABL) && CVC_NN128_ABL == 1          // ablation: no MFMAs (memory + split work only)
                asm volatile("" ::"v"(W.hi), "v"(W.mid), "v"(W.lo), "v"(X[mt].hi), "v"(X[mt].mid), "v"(X[mt].lo));
#else
                acc[c][mt] = mfma_bf16(W.mid, X[mt].mid, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.lo, X[mt].hi, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].lo, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.mid, X[mt].hi, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].mid, acc[c][mt]);
                acc[c][mt] = mfma_bf16(W.hi, X[mt].hi, acc[c][mt]);
#endif
            }
#if CVC_NN128_SCHED && !(defined(CVC_NN128_ABL) && CVC_NN128_ABL == 1)
            if (c < 3) {
#pragma unroll
                for (int k = 0; k < 24; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two VALU instructions of the next tile's split
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    // wait until at most `cnt` younger loads (16 per step) are outstanding; the step's weight registers and the ring offset pass
    // through the statement, so nothing that reads them is scheduled above it
#define CVC_NN128_WAIT(f, cnt)                                                                                                   \
    asm volatile("s_waitcnt vmcnt(%9)"                                                                                             \
                 : "+v"((f).w[0]), "+v"((f).w[1]), "+v"((f).w[2]), "+v"((f).w[3]), "+v"((f).w[4]), "+v"((f).w[5]), "+v"((f).w[6]), \
                   "+v"((f).w[7]), "+v"(lane_off)                                                                                  \
                 : "n"(cnt))
    auto wait_for = [&](NNW8& f, int later) __attribute__((always_inline)) {
        if (later >= 3) CVC_NN128_WAIT(f, 48);
        else if (later == 2) CVC_NN128_WAIT(f, 32);
        else if (later == 1) CVC_NN128_WAIT(f, 16);
        else CVC_NN128_WAIT(f, 0);
    };

    NNW8 wr[D];
    if (n2 >= D) {
        // (the steady state's loads are unconditional from the prologue on: a conditionally issued load in front of the loop
        // degrades hipcc's counted vmcnt inside it to 0)
#pragma unroll
        for (int t = 0; t < D - 1; ++t) issue(wr[t], t);
        int j = 0;
        for (; j + 2 * D - 1 <= n2; j += D) {
#pragma unroll
            for (int t = 0; t < D; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                issue(wr[(t + D - 1) % D], (t + D - 1) % D);
                __builtin_amdgcn_sched_barrier(0);            // requests first (the scheduler would sink them behind the MFMAs)
                CVC_NN128_WAIT(wr[t], 16 * (D - 1));
                compute(wr[t], t);
            }
        }
        // tail: D .. 2 D - 2 steps are left, D - 1 of them already requested; j % D == 0
#pragma unroll
        for (int t = 0; t < 2 * D - 2; ++t) {
            const int st = j + t;
            if (st < n2) {
                __builtin_amdgcn_sched_barrier(0);
                if (st + D - 1 < n2) issue(wr[(t + D - 1) % D], (t + D - 1) % D);
                __builtin_amdgcn_sched_barrier(0);
                const int later = n2 - 1 - st;
                wait_for(wr[t % D], later < D - 1 ? later : D - 1);
                compute(wr[t % D], t % D);
            }
        }
    } else {
        for (int j = 0; j < n2; ++j) {
            issue(wr[0], 0);
            CVC_NN128_WAIT(wr[0], 0);
            compute(wr[0], 0);
        }
    }
#undef CVC_NN128_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (nothing the compiler can see moves above the K loop's last wait)
    if (n_my & 1) {                                           // odd last group: exact fp32 products on the fp32 MFMA
        f32x4 w[4], x[MT];
        const float* xo[2] = {xs[0] - lane * 4 + ((size_t)kh * 64 + i) * 4, xs[1] - lane * 4 + ((size_t)kh * 64 + i) * 4};
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = ld4(wp[e]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) x[mt] = ld4(xo[mt >> 1] + (mt & 1) * 128);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[c][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e][c], x[mt][e], acc[c][mt], 0, 0, 0);
    }

#ifdef CVC_NN128_TS
    ts1 = __builtin_readcyclecounter();
#endif
    // ---- cross-wave sum and store (nn_cross_wave_store; its buffers take over the ring's LDS: every wave is through with it)
    __syncthreads();
    float* out;
    size_t ld;
    int cbase;
    if (a.ksplit > 1) {
        out = a.part + (size_t)kslice * 128 * a.ntot;
        ld = (size_t)a.ntot;
        cbase = sg.slab0 * 128 + n0;
    } else {
        out = sg.dst;
        ld = (size_t)sg.ld_dst;
        cbase = n0;
    }
#ifdef CVC_NN128_TS
    ts2 = __builtin_readcyclecounter();
#endif
    nn_cross_wave_store<MT>(acc, lds, wave, lane, out, ld, cbase, sg.ncols - n0 < 128 ? sg.ncols - n0 : 128, a.M, a.M2);
#ifdef CVC_NN128_TS
    if (lane == 0) {      // the record lands behind the planes (the measurement script allocates room): [wg][wave][6]
        unsigned long long* rec = reinterpret_cast<unsigned long long*>(a.part + (size_t)a.ksplit * 128 * a.ntot) + ((size_t)blockIdx.x * 4 + wave) * 6;
        rec[0] = rt0; rec[1] = ts1 - ts0; rec[2] = ts2 - ts1; rec[3] = __builtin_readcyclecounter() - ts2; rec[4] = __builtin_amdgcn_s_memrealtime();
        rec[5] = (unsigned long long)n2;
    }
#endif
}

// dst[seg][m][n] = sum over planes of the 128-row form (rows 64 .. of a plane / of dst = the second group), fixed order
__global__ __launch_bounds__(256) void nn_reduce128_kernel(NNArgs a) {
    const int m = blockIdx.y;
    if (m < 64 ? m >= a.M : m - 64 >= a.M2) return;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= a.ntot) return;
    int s = 0;
    for (int t = 1; t < a.nseg; ++t)
        if (c >= a.seg[t].slab0 * 128) s = t;
    const NNSeg sg = a.seg[s];
    const int n = c - sg.slab0 * 128;
    if (n >= sg.ncols) return;
    const float* p = a.part + (size_t)m * a.ntot + c;
    f32x4 v = ld4(p);
    for (int k = 1; k < a.ksplit; ++k) v += ld4(p + (size_t)k * 128 * a.ntot);
    st4(sg.dst + (size_t)m * sg.ld_dst + n, v);
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

// 128-row form: rows 0 .. M - 1 from dy_q, rows 64 .. 64 + M2 - 1 from dy_q2 (both in the 64-row quad layout, rows beyond M / M2
// zero), ONE stream of the weights.  Results: K-slice planes [ksplit][128][ntot] in `workspace` (reduce = 0, ksplit > 1) or the
// segments' dst [128 rows, ld_dst] (row 64 + m = row m of the second group).  Split-product arithmetic only.
extern "C" int cvc_linear_nn_planes2_fwd(const float* dy_q, const float* dy_q2, int K, int M, int M2, const cvc_nn_seg* segs, int nsegs,
                                         int ksplit, float* workspace, int reduce, cvc_stream_t stream) {
    if (!dy_q || !dy_q2 || !segs || nsegs < 1 || nsegs > NN_MAX_SEG || M < 1 || M > 64 || M2 < 1 || M2 > 64 || K < 8 || (K & 7) ||
        ksplit < 1 || ((uintptr_t)dy_q & 15) || ((uintptr_t)dy_q2 & 15))
        return CVC_E_BADARG;
    if (cvc_gemm_split_mode == 0) return CVC_E_BADARG;
    if (ksplit > K / 8) ksplit = K / 8;
    NNArgs a{};
    a.xq = dy_q; a.xq2 = dy_q2; a.K = K; a.M = M; a.M2 = M2; a.nseg = nsegs; a.ksplit = ksplit; a.part = workspace;
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
    hipLaunchKernelGGL(skinny_gemm_nn_split128_kernel, dim3(slab * ksplit), dim3(256), 0, (hipStream_t)stream, a);
    if (ksplit > 1 && reduce)
        hipLaunchKernelGGL(nn_reduce128_kernel, dim3((a.ntot / 4 + 255) / 256, 128), dim3(256), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}
