// Tile GEMM of the decode engine for MORE than 64 live rows (beam search: rows = clips x beam; big greedy batches),
// its finishing kernels, and the beam-state reorder that feeds it.
//
// Why a second GEMM family: with 64 rows the gate GEMMs are weight-streaming problems (gemm_packed.hip: one workgroup = 32
// weight rows x all rows x full K, operands straight into registers).  With 320 rows (cfg3 / cfg5, beam 5) the same tiling needs
// (32 + 320) operand rows per 32 x 320 products: 70 B of L2 ingress per matrix-pipe cycle and CU, three times what a CU can take
// in, and the row-major ring kernel it used to fall back to re-streamed the weights once per 64-row slab.  Here:
//
//   * one workgroup = 128 weight rows x up to 320 rows x K / ksplit; 8 waves = 4 weight blocks x 2 row halves, every wave
//     owning 32 x (32 * MH) outputs (MH <= 5 accumulator tiles, 80 registers);
//   * BOTH operands are stored in HBM the way the matrix pipe wants them: split once into the three bf16 terms of the
//     split-product arithmetic (gemm_split.h: v = hi + mid + lo exactly) and cut into 1 KiB fragments = 32 rows x 16 k of one
//     term, ordered [k half][row][8 k] = the lane order of v_mfma_f32_32x32x16_bf16's A / B operand.  Weights are packed once
//     per checkpoint (cvc/decode.py::pack_weights_tile), activations are written in this form by their producers (the finishing
//     kernels below, attn_wsum, beam_reorder_pack), so the K loop has NO operand-preparation VALU work at all;
//   * a K step (16 k) of the tile is (4 + 2 MH) x 3 fragments = 42 KiB at MH = 5, copied by LDS-DMA (global_load_lds_dwordx4, one
//     fragment per wave instruction, lane-linear = the fragment itself, so fragment reads are conflict-free ds_read_b128 with no
//     swizzle) into a 3-stage ring: two stages in flight, ONE raw s_barrier per stage, counted s_waitcnt vmcnt;
//   * per stage a wave issues 3 + 3 MH ds_read_b128 and 6 MH MFMAs (the six leading cross terms, small terms first);
//   * K is split over workgroups so that 256 of them exist (N / 128 tiles x ksplit); the partial tiles go to fp32 row-major
//     slabs [ksplit][rows, ld] and are summed IN A FIXED ORDER by the consumer that exists anyway: the LSTM finishing kernel
//     (bias + gate non-linearities + cell update), the attention score kernel (query partials, attn_scores.h), or
//     tile_linear_finish (vocabulary logits).
//
// Arithmetic is the same as the packed path's split mode (6 bf16 MFMAs per 16 k, fp32 accumulate): fp32-grade error
// (tests/test_gpu_parity.py::test_tile_gemm_vs_fp64).
#include "cvc_common.h"
#include "gemm_split.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#ifndef CVC_TILE_VARIANT
#define CVC_TILE_VARIANT 1
#endif
constexpr int FRAG = 512;              // bf16 elements per fragment (32 rows x 16 k) = 1 KiB
constexpr int KSTEP = 3 * FRAG;        // bf16 elements per (block, k step): three split terms

struct TileArgs {
    const uint16_t* wb;        // [nblk_pad][w_ksteps][3][FRAG]   packed weights, nblk_pad a multiple of 4
    const uint16_t* xb;        // [mblk][...][3][FRAG]            packed activations, first k step of this GEMM's K range
    long long x_mblk_stride;   // bf16 elements between consecutive 32-row blocks of xb
    int ksteps;                // K / 16
    int M, N;                  // live rows / output features (store guards)
    int ntile;                 // ceil(N / 128)
    int ksplit;
    float* parts;              // [ksplit] slabs, row-major [M, ld]
    int ld;
    long long part_stride;
};

template <int MH>
__global__ __launch_bounds__(512) void tile_gemm_kernel(TileArgs a) {
    constexpr int NX = 2 * MH;                  // 32-row activation blocks per workgroup
    constexpr int NF = (4 + NX) * 3;            // fragments per stage
    constexpr int ND = (NF + 7) / 8;            // LDS-DMA instructions per wave and stage
    constexpr int STAGE = NF * 1024;
    constexpr int NSTAGE = 3;
    __shared__ __attribute__((aligned(16))) char lds[NSTAGE * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wbk = wave & 3, mh = wave >> 2;
    // workgroup -> (weight tile, K slice).  Speed only: workgroups are dispatched round-robin over the 8 XCDs (private L2s), so
    // when ksplit divides 8 every XCD is given ONE K slice: its L2 then pulls 1 / ksplit of the activations instead of all of
    // them (at 320 x 6144 the activations are 11.8 MB of fragments against 4 MB of L2)
    int tile = (int)blockIdx.x % a.ntile, ks = (int)blockIdx.x / a.ntile;
#ifndef CVC_TILE_NO_XCD
    if (a.ksplit > 1 && a.ksplit <= 8 && 8 % a.ksplit == 0 && a.ntile % (8 / a.ksplit) == 0) {
        const int g = 8 / a.ksplit, xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        ks = xcd / g;
        tile = j * g + xcd % g;
    }
#endif
    const int mb0 = (int)blockIdx.y * NX;
    const int s_lo = (int)((long long)a.ksteps * ks / a.ksplit), s_hi = (int)((long long)a.ksteps * (ks + 1) / a.ksplit);
    const int nst = s_hi - s_lo;

    // this wave's share of a stage's fragments: f = wave + 8 j.  NF is not a multiple of 8: the first NF % 8 waves copy one
    // fragment more than the others (their counted vmcnt is one higher; the wave index is uniform, so that is a scalar branch)
    constexpr int NDLO = NF / 8, NEXTRA = NF % 8;
    const bool extra = NEXTRA != 0 && wave < NEXTRA;
    const uint16_t* src[ND];
    int dst[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        int f = wave + 8 * j;
        if (f >= NF) f -= NF;                    // (never copied: j == NDLO is issued by waves < NEXTRA only)
        const int g = f < 12 ? f : f - 12;
        const int blk = g / 3, pl = g - blk * 3;
        if (f < 12) src[j] = a.wb + ((size_t)(tile * 4 + blk) * a.ksteps + s_lo) * KSTEP + pl * FRAG + lane * 8;
        else src[j] = a.xb + (size_t)(mb0 + blk) * a.x_mblk_stride + (size_t)s_lo * KSTEP + pl * FRAG + lane * 8;
        dst[j] = f * 1024;
    }
    auto issue = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NDLO; ++j)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + (size_t)s * KSTEP), (lds_ptr_t)(lds + buf * STAGE + dst[j]), 16, 0, 0);
        if constexpr (NEXTRA != 0)
            if (extra)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[NDLO] + (size_t)s * KSTEP), (lds_ptr_t)(lds + buf * STAGE + dst[NDLO]), 16, 0, 0);
    };

    f32x16 acc[MH];
#pragma unroll
    for (int t = 0; t < MH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (nst > 0) issue(0, 0);
    if (nst > 1) issue(1, 1);
    int buf = 0;
    for (int s = 0; s < nst; ++s) {
        // stage s has landed once at most the copies of stage s + 1 are still outstanding
        if (s + 1 < nst) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDLO + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDLO) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#if !defined(CVC_TILE_ABL) || (CVC_TILE_ABL != 5 && CVC_TILE_ABL < 6)
        __builtin_amdgcn_s_barrier();           // every wave's copies of stage s are in LDS; stage s - 1 has been read by all
#endif
#if !defined(CVC_TILE_ABL) || CVC_TILE_ABL < 2 || CVC_TILE_ABL == 3
        if (s + 2 < nst) issue(s + 2, buf == 0 ? 2 : buf - 1);
#else
        if (s + 2 < nst && extra) asm volatile("s_nop 0");    // ablation 2: no copies after the prologue (compute side only)
#endif
        const char* base = lds + buf * STAGE + lane * 16;
        auto frag = [&](int f) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(base + f * 1024); };
        // fragment reads of tile t + 1 are issued before the six MFMAs of tile t, so that LDS latency runs under the matrix pipe
        u32x4 w[3], x[2][3];
#if defined(CVC_TILE_ABL) && (CVC_TILE_ABL == 4 || CVC_TILE_ABL >= 6)
        // ablation 4 / 6: fragment reads only in the first stage (matrix pipe + barrier only)
        const char* base0 = lds + lane * 16;
        auto frag0 = [&](int f) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(base0 + (s == 0 ? f : 0) * 1024); };
        if (s == 0) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { w[pl] = frag0(wbk * 3 + pl); x[0][pl] = frag0(12 + pl); x[1][pl] = frag0(15 + pl); }
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { asm volatile("" : "+v"(w[pl]), "+v"(x[0][pl]), "+v"(x[1][pl])); }
#if CVC_TILE_ABL == 8
        // ablation 8: the six terms outermost -> consecutive MFMAs hit different accumulators
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const u32x4* xt = x[t & 1];
                const int xi = term == 0 ? 1 : (term == 2 ? 2 : (term == 4 ? 1 : 0));
                const int wi = term == 0 ? 1 : (term == 1 ? 2 : (term == 3 ? 1 : 0));
                acc[t] = mfma_bf16(xt[xi], w[wi], acc[t]);
            }
#else
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            const u32x4* xt = x[t & 1];
            acc[t] = mfma_bf16(xt[1], w[1], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[2], acc[t]);
            acc[t] = mfma_bf16(xt[2], w[0], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[1], acc[t]);
            acc[t] = mfma_bf16(xt[1], w[0], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[0], acc[t]);
        }
#endif
        buf = buf == 2 ? 0 : buf + 1;
        continue;
#endif
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { w[pl] = frag(wbk * 3 + pl); x[0][pl] = frag(12 + (mh * MH) * 3 + pl); }
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            if (t + 1 < MH) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) x[(t + 1) & 1][pl] = frag(12 + (mh * MH + t + 1) * 3 + pl);
            }
            const u32x4* xt = x[t & 1];
#if CVC_TILE_VARIANT >= 1
            __builtin_amdgcn_sched_barrier(0);      // the next tile's reads are issued BEFORE this tile's MFMAs
#endif
#if CVC_TILE_VARIANT == 2
            __builtin_amdgcn_s_setprio(1);
#endif
#if defined(CVC_TILE_ABL) && CVC_TILE_ABL == 1
            asm volatile("" ::"v"(xt[0]), "v"(xt[1]), "v"(xt[2]), "v"(w[0]), "v"(w[1]), "v"(w[2]));   // ablation 1: memory side only
            continue;
#endif
            // D[i = activation row][j = weight row] += X[i][k] W[j][k]; terms 0 / 1 / 2 = hi / mid / lo, small products first
            acc[t] = mfma_bf16(xt[1], w[1], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[2], acc[t]);
            acc[t] = mfma_bf16(xt[2], w[0], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[1], acc[t]);
            acc[t] = mfma_bf16(xt[1], w[0], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[0], acc[t]);
            // pin the order: [3 reads of the next tile] then [6 MFMAs]
#if CVC_TILE_VARIANT == 2
            __builtin_amdgcn_s_setprio(0);
#endif
#if CVC_TILE_VARIANT >= 1
            __builtin_amdgcn_sched_barrier(0);
#else
            if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, MH > 1 ? 9 : 6, 0);
            else if (t + 1 < MH) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#endif
        }
        buf = buf == 2 ? 0 : buf + 1;
    }

    // partial tile -> slab ks, row-major: lane = output feature (32 consecutive floats per half wave), register = row
    float* out = a.parts + (size_t)ks * a.part_stride;
    const int n = (tile * 4 + wbk) * 32 + (lane & 31);
    const int kh = lane >> 5;
#if defined(CVC_TILE_ABL) && (CVC_TILE_ABL == 3 || CVC_TILE_ABL == 7)
    if (acc[0][0] == 12345.678f)                              // ablation 3 / 7: no epilogue stores
#endif
    if (n < a.N) {
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            const int mbase = (mb0 + mh * MH + t) * 32 + 4 * kh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < a.M) out[(size_t)m * a.ld + n] = acc[t][r];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Loader / consumer variant: the LDS-DMA copies are issued by NL extra waves that do nothing else, the eight computing waves
// only read fragments and issue MFMAs.  Why: ablations of tile_gemm_kernel at 320 x 6144 x 8192 -- compute side alone (no
// copies) 143 us, memory side alone 92 us, together 190 us: the sides do not overlap because a wave that issues its share of
// the copies (6 x global_load_lds, ~100-180 issue cycles each next to LDS reads) cannot issue MFMAs meanwhile, and that is as
// long as the stage's matrix work (960 cycles per wave).  Same ring, same one barrier per k step, counted vmcnt in the loaders.
template <int MH, int NL>
__global__ __launch_bounds__((8 + NL) * 64) void tile_gemm_ld_kernel(TileArgs a) {
    constexpr int NX = 2 * MH;
    constexpr int NF = (4 + NX) * 3;
    constexpr int STAGE = NF * 1024;
    constexpr int NSTAGE = 3;
    __shared__ __attribute__((aligned(16))) char lds[NSTAGE * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = (int)blockIdx.x % a.ntile, ks = (int)blockIdx.x / a.ntile;
    if (a.ksplit > 1 && a.ksplit <= 8 && 8 % a.ksplit == 0 && a.ntile % (8 / a.ksplit) == 0) {
        const int g = 8 / a.ksplit, xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        ks = xcd / g;
        tile = j * g + xcd % g;
    }
    const int mb0 = (int)blockIdx.y * NX;
    const int s_lo = (int)((long long)a.ksteps * ks / a.ksplit), s_hi = (int)((long long)a.ksteps * (ks + 1) / a.ksplit);
    const int nst = s_hi - s_lo;

    if (wave >= 8) {
        // ---------------- loader wave: fragments f = lw + NL j of every stage
        const int lw = wave - 8;
        constexpr int NDLO = NF / NL, NEXTRA = NF % NL, ND = NDLO + (NEXTRA ? 1 : 0);
        const bool extra = NEXTRA != 0 && lw < NEXTRA;
        const uint16_t* src[ND];
        int dst[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            int f = lw + NL * j;
            if (f >= NF) f -= NF;
            const int g = f < 12 ? f : f - 12;
            const int blk = g / 3, pl = g - blk * 3;
            if (f < 12) src[j] = a.wb + ((size_t)(tile * 4 + blk) * a.ksteps + s_lo) * KSTEP + pl * FRAG + lane * 8;
            else src[j] = a.xb + (size_t)(mb0 + blk) * a.x_mblk_stride + (size_t)s_lo * KSTEP + pl * FRAG + lane * 8;
            dst[j] = f * 1024;
        }
        auto issue = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NDLO; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + (size_t)s * KSTEP), (lds_ptr_t)(lds + buf * STAGE + dst[j]), 16, 0, 0);
            if constexpr (NEXTRA != 0)
                if (extra)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[NDLO] + (size_t)s * KSTEP), (lds_ptr_t)(lds + buf * STAGE + dst[NDLO]), 16, 0, 0);
        };
        if (nst > 0) issue(0, 0);
        if (nst > 1) issue(1, 1);
        int buf = 0;
        for (int s = 0; s < nst; ++s) {
            if (s + 1 < nst) {
                if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDLO + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDLO) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();       // stage s is in LDS (all loaders waited); stage s - 1 has been read by all
            if (s + 2 < nst) issue(s + 2, buf == 0 ? 2 : buf - 1);
            buf = buf == 2 ? 0 : buf + 1;
        }
        return;
    }

    // ---------------- computing wave
    const int wbk = wave & 3, mh = wave >> 2;
    f32x16 acc[MH];
#pragma unroll
    for (int t = 0; t < MH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int buf = 0;
    for (int s = 0; s < nst; ++s) {
        __builtin_amdgcn_s_barrier();
        const char* base = lds + buf * STAGE + lane * 16;
        auto frag = [&](int f) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(base + f * 1024); };
        u32x4 w[3], x[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { w[pl] = frag(wbk * 3 + pl); x[0][pl] = frag(12 + (mh * MH) * 3 + pl); }
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            if (t + 1 < MH) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) x[(t + 1) & 1][pl] = frag(12 + (mh * MH + t + 1) * 3 + pl);
            }
            const u32x4* xt = x[t & 1];
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = mfma_bf16(xt[1], w[1], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[2], acc[t]);
            acc[t] = mfma_bf16(xt[2], w[0], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[1], acc[t]);
            acc[t] = mfma_bf16(xt[1], w[0], acc[t]);
            acc[t] = mfma_bf16(xt[0], w[0], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
    float* out = a.parts + (size_t)ks * a.part_stride;
    const int n = (tile * 4 + wbk) * 32 + (lane & 31);
    const int kh = lane >> 5;
    if (n < a.N) {
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            const int mbase = (mb0 + mh * MH + t) * 32 + 4 * kh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < a.M) out[(size_t)m * a.ld + n] = acc[t][r];
            }
        }
    }
}

// ---- split one fp32 into its three bf16 terms (same truncation as gemm_split.h::split8)
__device__ __forceinline__ void split3(float v, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
    const unsigned u0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    hi = (uint16_t)(u0 >> 16); mid = (uint16_t)(u1 >> 16); lo = (uint16_t)(__float_as_uint(r2) >> 16);
}

using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;

// 4 consecutive k of activation row m, starting at k (k % 4 == 0), into the three term fragments of its block
__device__ __forceinline__ void store_frag4(uint16_t* xb, long long mblk_stride, int m, int k, const f32x4 v) {
    u16x4 p[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        uint16_t h, mi, l;
        split3(v[e], h, mi, l);
        p[0][e] = h; p[1][e] = mi; p[2][e] = l;
    }
    uint16_t* base = xb + (size_t)(m >> 5) * mblk_stride + (size_t)(k >> 4) * KSTEP + (((k >> 3) & 1) * 32 + (m & 31)) * 8 + (k & 7);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u16x4*>(base + pl * FRAG) = p[pl];
}

struct LstmFinishArgs {
    const float* parts; int nparts; long long part_stride;   // slabs [M, 4R], feature order = packed: blk * 32 + gate * 8 + unit
    const float* b_ih; const float* b_hh;                    // [4R] checkpoint order (gate * R + hidden), nullable
    const float* gate_bias; int gb_div;                      // [ceil(M / gb_div), 4R] checkpoint order, row m / gb_div; nullable
    const float* c_prev;                                     // [M, R]
    float* c_out; float* h_out;                              // [M, R]; h_out nullable
    uint16_t* frag1; long long frag1_stride;                 // h' as activation fragments (pointer at its first k step), nullable
    uint16_t* frag2; long long frag2_stride;
    int M, R;
    const float* emb_gate; const int64_t* word; int V;       // embedding-gate table [V, 4R] checkpoint order + the row's word; nullable
};

// Wide-wave form of tile_gemm_ld_kernel: 4 computing waves, each 2 weight blocks x MH row tiles (2 x MH accumulator tiles),
// NL loader waves.  Against the 8-wave form: a row-tile fragment read from LDS feeds 12 MFMAs instead of 6 (LDS read traffic
// per k step 144 -> 84 KB at MH = 5; the LDS was busy ~78 % of the MFMA time), and the k step's barrier sits BEFORE the last row
// tile's MFMAs -- its fragments are in registers by then -- so that the next stage's first fragments are read under those
// MFMAs instead of in a bubble at the top of every stage.
#ifdef CVC_TILE_TS
// diagnostic build (-DCVC_TILE_TS, tools/runs/r06_tile_ts.py): who waits for whom at the per-k-step barrier?  Workgroup (0, 0), k steps
// 8 .. 71: [0][s] = computing wave 0: {arrival at barrier s + 1, release}; [1][s] = loader wave 0: {its stage's data complete (vmcnt),
// release of barrier s, copies of stage s + 2 issued}.  100 MHz constant clock (s_memrealtime).
__device__ unsigned long long cvc_tile_ts_buf[2][64][3];
__device__ unsigned long long cvc_tile_clk_buf[4];        // computing wave 0 of workgroup (0, 0): {realtime, shader clock} at k step 8 and 71
extern "C" __attribute__((visibility("default"))) int cvc_tile_ts_read(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cvc_tile_ts_buf), sizeof(cvc_tile_ts_buf)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int cvc_tile_clk_read(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cvc_tile_clk_buf), sizeof(cvc_tile_clk_buf)) == hipSuccess ? 0 : -1;
}
#define CVC_TILE_TS_REC(who, s, k) do { if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (s) >= 8 && (s) < 72) \
        cvc_tile_ts_buf[who][(s) - 8][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CVC_TILE_TS_REC(who, s, k) do {} while (0)
#endif

// REGLOAD (round 6): the loader waves copy a stage with ordinary register loads + ds_write_b128 instead of LDS-DMA.  In-kernel
// timestamps (tools/runs/r06_tile_ts.py) showed where the k step's 1.86 us (pure MFMA: 0.8) goes: a loader wave spends 1.55 us ISSUING
// its ~10 global_load_lds instructions of a stage (~150 ns each: the instruction holds the wave until the texture path has taken it)
// and the computing waves wait 0.4 us per step at the barrier -- the copy is bound by the issue rate of four waves, not by the
// memory system (every wave copying: 0.96 us per stage).  A register load issues in a few cycles and is fully asynchronous: two
// register sets per loader wave, a stage requested two k steps before it is written to LDS, three LDS stages as before.
template <int MH, int NL, bool REGLOAD = false>
__global__ __launch_bounds__((4 + NL) * 64) void tile_gemm_ld2_kernel(TileArgs a) {
    constexpr int NX = 2 * MH;
    constexpr int NF = (4 + NX) * 3;
    constexpr int STAGE = NF * 1024;
    constexpr int NSTAGE = 3;
    __shared__ __attribute__((aligned(16))) char lds[NSTAGE * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = (int)blockIdx.x % a.ntile, ks = (int)blockIdx.x / a.ntile;
    if (a.ksplit > 1 && a.ksplit <= 8 && 8 % a.ksplit == 0 && a.ntile % (8 / a.ksplit) == 0) {
        const int g = 8 / a.ksplit, xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        ks = xcd / g;
        tile = j * g + xcd % g;
    }
    const int mb0 = (int)blockIdx.y * NX;
    const int s_lo = (int)((long long)a.ksteps * ks / a.ksplit), s_hi = (int)((long long)a.ksteps * (ks + 1) / a.ksplit);
    const int nst = s_hi - s_lo;

    if constexpr (REGLOAD) {
      if (wave >= 4) {
        const int lw = wave - 4;
        constexpr int ND = (NF + NL - 1) / NL;            // pieces per loader wave and stage (a wrapped duplicate where NF % NL != 0)
        const uint16_t* src[ND];
        int dst[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            int f = lw + NL * j;
            if (f >= NF) f -= NF;
            const int g = f < 12 ? f : f - 12;
            const int blk = g / 3, pl = g - blk * 3;
            if (f < 12) src[j] = a.wb + ((size_t)(tile * 4 + blk) * a.ksteps + s_lo) * KSTEP + pl * FRAG + lane * 8;
            else src[j] = a.xb + (size_t)(mb0 + blk) * a.x_mblk_stride + (size_t)s_lo * KSTEP + pl * FRAG + lane * 8;
            dst[j] = f * 1024 + lane * 16;
        }
        u32x4 ra[ND], rb[ND];
        // every load is issued unconditionally (a stage index past the end re-reads the last stage): a conditional load inside the
        // pipelined loop would degrade the compiler's counted vmcnt waits to vmcnt(0)
        auto ld = [&](u32x4 (&r)[ND], int st) __attribute__((always_inline)) {
            const int sc = st < nst ? st : nst - 1;
#pragma unroll
            for (int j = 0; j < ND; ++j) r[j] = *reinterpret_cast<const u32x4*>(src[j] + (size_t)sc * KSTEP);
        };
        auto stg = [&](const u32x4 (&r)[ND], int st) __attribute__((always_inline)) {
            char* base = lds + (st % 3) * STAGE;
#pragma unroll
            for (int j = 0; j < ND; ++j) *reinterpret_cast<u32x4*>(base + dst[j]) = r[j];
        };
        if (nst <= 0) return;
        ld(ra, 0); ld(rb, 1);
        stg(ra, 0); ld(ra, 2);
        stg(rb, 1); ld(rb, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // barrier 0: stages 0 and 1 are in LDS
        // iteration s (after barrier s): stage s + 2 (requested two iterations ago) -> LDS buffer (s + 2) % 3, whose stage s - 1 every
        // computing wave has finished; the set then takes stage s + 4; barrier s + 1 publishes stage s + 1 (written one iteration ago)
        for (int s = 0; s + 1 < nst; s += 2) {
            if (s + 2 < nst) stg(ra, s + 2);
            ld(ra, s + 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // barrier s + 1
            if (s + 2 >= nst) break;
            if (s + 3 < nst) stg(rb, s + 3);
            ld(rb, s + 5);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // barrier s + 2
        }
        return;
      }
    } else
    if (wave >= 4) {
        // ---------------- loader wave (as in tile_gemm_ld_kernel); one barrier per stage, in step with the computing waves
        const int lw = wave - 4;
        constexpr int NDLO = NF / NL, NEXTRA = NF % NL, ND = NDLO + (NEXTRA ? 1 : 0);
        const bool extra = NEXTRA != 0 && lw < NEXTRA;
        const uint16_t* src[ND];
        int dst[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            int f = lw + NL * j;
            if (f >= NF) f -= NF;
            const int g = f < 12 ? f : f - 12;
            const int blk = g / 3, pl = g - blk * 3;
            if (f < 12) src[j] = a.wb + ((size_t)(tile * 4 + blk) * a.ksteps + s_lo) * KSTEP + pl * FRAG + lane * 8;
            else src[j] = a.xb + (size_t)(mb0 + blk) * a.x_mblk_stride + (size_t)s_lo * KSTEP + pl * FRAG + lane * 8;
            dst[j] = f * 1024;
        }
        auto issue = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NDLO; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[j] + (size_t)s * KSTEP), (lds_ptr_t)(lds + buf * STAGE + dst[j]), 16, 0, 0);
            if constexpr (NEXTRA != 0)
                if (extra)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[NDLO] + (size_t)s * KSTEP), (lds_ptr_t)(lds + buf * STAGE + dst[NDLO]), 16, 0, 0);
        };
        if (nst > 0) issue(0, 0);
        if (nst > 1) issue(1, 1);
        int buf = 0;
        // barrier k (k = 0 .. nst - 1) publishes stage k; the computing waves pass it when every fragment of stage k - 1 they
        // will ever read is in their registers, so stage k + 2 may overwrite the buffer of stage k - 1
        for (int s = 0; s < nst; ++s) {
            if (s + 1 < nst) {
                if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDLO + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDLO) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (lw == 0) CVC_TILE_TS_REC(1, s, 0);
            __builtin_amdgcn_s_barrier();
            if (lw == 0) CVC_TILE_TS_REC(1, s, 1);
            if (s + 2 < nst) issue(s + 2, buf == 0 ? 2 : buf - 1);
            if (lw == 0) CVC_TILE_TS_REC(1, s, 2);
            buf = buf == 2 ? 0 : buf + 1;
        }
        return;
    }

    // ---------------- computing wave: weight blocks 2 wp, 2 wp + 1; row tiles mh * MH .. + MH - 1
    const int wp = wave & 1, mh = wave >> 1;
    f32x16 acc[2][MH];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < MH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
    u32x4 w[2][3], x[2][3];
    auto frag = [&](int buf, int f) __attribute__((always_inline)) {
        return *reinterpret_cast<const u32x4*>(lds + buf * STAGE + lane * 16 + f * 1024);
    };
    auto first_frags = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            w[0][pl] = frag(buf, (2 * wp) * 3 + pl);
            w[1][pl] = frag(buf, (2 * wp + 1) * 3 + pl);
            x[0][pl] = frag(buf, 12 + (mh * MH) * 3 + pl);
        }
    };
    auto mma = [&](const u32x4* xt, const int t) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc[b][t] = mfma_bf16(xt[1], w[b][1], acc[b][t]);
            acc[b][t] = mfma_bf16(xt[0], w[b][2], acc[b][t]);
            acc[b][t] = mfma_bf16(xt[2], w[b][0], acc[b][t]);
            acc[b][t] = mfma_bf16(xt[0], w[b][1], acc[b][t]);
            acc[b][t] = mfma_bf16(xt[1], w[b][0], acc[b][t]);
            acc[b][t] = mfma_bf16(xt[0], w[b][0], acc[b][t]);
        }
    };
    if (nst > 0) {
        __builtin_amdgcn_s_barrier();                               // barrier 0: stage 0 is in LDS
        first_frags(0);
    }
    int buf = 0;
    for (int s = 0; s < nst; ++s) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
        // row tiles 0 .. MH - 2 of this stage, the next tile's fragments read under the current tile's MFMAs
#pragma unroll
        for (int t = 0; t + 1 < MH; ++t) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) x[(t + 1) & 1][pl] = frag(buf, 12 + (mh * MH + t + 1) * 3 + pl);
            __builtin_amdgcn_sched_barrier(0);
            mma(x[t & 1], t);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the last tile's fragments are in registers: this wave is done with the stage's LDS.  Its MFMAs are issued FIRST (they
        // still use this stage's weight registers), then the barrier that publishes the next stage and that stage's first reads,
        // which land while those MFMAs execute
        u32x4 xl[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xl[pl] = x[(MH - 1) & 1][pl];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma(xl, MH - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) {
            if (wave == 0) CVC_TILE_TS_REC(0, s + 1, 0);
#ifdef CVC_TILE_TS
            if (wave == 0 && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (s == 8 || s == 71)) {
                cvc_tile_clk_buf[s == 8 ? 0 : 2] = __builtin_amdgcn_s_memrealtime();
                cvc_tile_clk_buf[s == 8 ? 1 : 3] = __builtin_amdgcn_s_memtime();
            }
#endif
            __builtin_amdgcn_s_barrier();                           // barrier s + 1
            if (wave == 0) CVC_TILE_TS_REC(0, s + 1, 1);
            first_frags(nbuf);
        }
        buf = nbuf;
    }
    float* out = a.parts + (size_t)ks * a.part_stride;
    const int kh = lane >> 5;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = (tile * 4 + 2 * wp + b) * 32 + (lane & 31);
        if (n < a.N) {
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const int mbase = (mb0 + mh * MH + t) * 32 + 4 * kh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m < a.M) out[(size_t)m * a.ld + n] = acc[b][t][r];
                }
            }
        }
    }
}

// ---- 256 x 256 form (round 6) ---------------------------------------------------------------------------------------------------
// For the large products (every dW of the training step, the encoder's dense layers, the GRU's input projections): a workgroup of
// four waves, one per SIMD, owns 256 weight rows x 256 activation rows; a wave a 128 x 128 quarter = 4 x 4 accumulator tiles (256
// registers).  Per k step a wave reads 24 fragments from LDS for 96 MFMAs (the 128 x 320 form: 21 for 60) and the workgroup moves
// 48 KB into LDS for 384 MFMAs (42 KB for 240): a third less data motion per MFMA -- which is what the kernel pays for twice, in
// LDS / fabric time and, the clock being power-limited inside these kernels (1.5 - 1.67 GHz), in clock.
// Round 5 built this tile with every wave issuing its share of the stage as LDS-DMA copies and measured it SLOWER (513 against
// 474 us on dW of an LSTM block): an LDS-DMA instruction holds its wave ~150 ns (tools/runs/r06_tile_ts.py), twelve of them per k
// step next to 2 us of MFMAs.  Here the copies are register loads (a few issue cycles each, fully asynchronous) written to LDS with
// ds_write_b128: one register set per wave, a stage requested one (~2 us) k step before it is written, three LDS stages, one barrier
// per k step.  Same products in the same order per output element as the other forms: bit-identical results (tested).
// Requires M % 256 == 0, N % 256 == 0 (the operands' fragment buffers then hold whole tiles); grid = (N / 256 * ksplit, M / 256).
__global__ __launch_bounds__(256, 1) void tile_gemm_big_kernel(TileArgs a) {
    constexpr int NF = 16 * 3;                  // fragments per stage: 8 weight blocks + 8 row blocks, three terms each
    constexpr int STAGE = NF * 1024;
    constexpr int ND = NF / 4;                  // pieces per wave and stage
    __shared__ __attribute__((aligned(16))) char lds[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = a.N >> 8;
    const int tile = (int)blockIdx.x % ntile, ks = (int)blockIdx.x / ntile;
    const int mb0 = (int)blockIdx.y * 8;
    const int s_lo = (int)((long long)a.ksteps * ks / a.ksplit), s_hi = (int)((long long)a.ksteps * (ks + 1) / a.ksplit);
    const int nst = s_hi - s_lo;
    if (nst <= 0) return;
    const int wn = wave & 1, wm = wave >> 1;    // this wave's weight blocks 4 wn .. + 3, row tiles 4 wm .. + 3

    // ---- copies: piece f = wave * 12 + j of a stage.  Wave-uniform base addresses (scalar registers) + ONE per-lane offset: the
    // twelve pointers must not cost 24 vector registers next to 256 accumulators
    const char* sbase[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int f = wave * ND + j;
        const int g = f < 24 ? f : f - 24;
        const int blk = g / 3, pl = g - blk * 3;
        const uint16_t* p0 = f < 24 ? a.wb + ((size_t)(tile * 8 + blk) * a.ksteps + s_lo) * KSTEP + pl * FRAG
                                     : a.xb + (size_t)(mb0 + blk) * a.x_mblk_stride + (size_t)s_lo * KSTEP + pl * FRAG;
        sbase[j] = reinterpret_cast<const char*>(p0);
    }
    const unsigned lane_off = lane * 16;
    const unsigned wdst = wave * ND * 1024 + lane * 16;           // this wave's pieces are consecutive fragments of the stage
    u32x4 ra[ND];
    auto ld = [&](u32x4 (&r)[ND], int st) __attribute__((always_inline)) {       // unconditional (past the end: the last stage again)
        const size_t so = (size_t)(st < nst ? st : nst - 1) * (KSTEP * 2);
#pragma unroll
        for (int j = 0; j < ND; ++j) r[j] = *reinterpret_cast<const u32x4*>(sbase[j] + so + lane_off);
    };
    auto stg = [&](const u32x4 (&r)[ND], int st) __attribute__((always_inline)) {
        char* base = lds + (st % 3) * STAGE + wdst;
#pragma unroll
        for (int j = 0; j < ND; ++j) *reinterpret_cast<u32x4*>(base + j * 1024) = r[j];
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
    auto frag = [&](int buf, int f) __attribute__((always_inline)) {
        return *reinterpret_cast<const u32x4*>(lds + buf * STAGE + lane * 16 + f * 1024);
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        u32x4 w[4][3], x[2][3];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[b][pl] = frag(buf, (4 * wn + b) * 3 + pl);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[0][pl] = frag(buf, 24 + (4 * wm) * 3 + pl);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t + 1 < 4) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) x[(t + 1) & 1][pl] = frag(buf, 24 + (4 * wm + t + 1) * 3 + pl);
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4* xt = x[t & 1];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                acc[b][t] = mfma_bf16(xt[1], w[b][1], acc[b][t]);
                acc[b][t] = mfma_bf16(xt[0], w[b][2], acc[b][t]);
                acc[b][t] = mfma_bf16(xt[2], w[b][0], acc[b][t]);
                acc[b][t] = mfma_bf16(xt[0], w[b][1], acc[b][t]);
                acc[b][t] = mfma_bf16(xt[1], w[b][0], acc[b][t]);
                acc[b][t] = mfma_bf16(xt[0], w[b][0], acc[b][t]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    ld(ra, 0); stg(ra, 0);
    ld(ra, 1); stg(ra, 1);
    ld(ra, 2);
    __syncthreads();                                                // stages 0 and 1 are in LDS, stage 2 is on its way
    // iteration s: stage s + 2 (requested one iteration = one ~2 us k step ago) -> LDS buffer (s + 2) % 3, which held stage s - 1 --
    // every wave left it before the barrier that ended iteration s - 1 --, the register set then takes stage s + 3; stage s is
    // multiplied; the closing barrier publishes stage s + 2 and releases buffer s % 3.  (ONE register set: with two, next to 256
    // accumulator registers, the compiler spilled the copies themselves.)
    for (int s = 0; s < nst; ++s) {
        if (s + 2 < nst) stg(ra, s + 2);
        ld(ra, s + 3);
        compute(s % 3);
        __syncthreads();
    }
    float* out = a.parts + (size_t)ks * a.part_stride;
    const int kh = lane >> 5;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = (tile * 8 + 4 * wn + b) * 32 + (lane & 31);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int mbase = (mb0 + 4 * wm + t) * 32 + 4 * kh;
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(mbase + (r & 3) + 8 * (r >> 2)) * a.ld + n] = acc[b][t][r];
        }
    }
}

// NP = number of slabs when it is 1 / 2 / 4 / 8 (their reads are then issued together), 0 = any number (one after the other)
template <int NP>
__global__ __launch_bounds__(256) void tile_lstm_finish_kernel(LstmFinishArgs a) {
    const int R = a.R;
    // A wave = 32 rows x the 2 hidden quads of ONE packed block (8 hidden units, 32 slab columns): its slab reads are 32-byte
    // runs that together use every fetched line, and its fragment writes are 512 contiguous bytes per term (row stride 16 B,
    // the two quads side by side) -- with the hidden quad fastest across a wave the fragment stores were 8-byte scatters.
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int hq = (int)(q & 1), i = (int)((q >> 1) & 31);
    const long long unit = q >> 6;                           // (row block, packed block), packed block fastest
    const int nblk = R >> 3;
    const int blk = (int)(unit % nblk), mb = (int)(unit / nblk);
    const int m = mb * 32 + i;
    if (m >= a.M) return;
    const int j = blk * 8 + hq * 4;                          // hidden units j .. j + 3
    const int col = blk * 32 + hq * 4;                       // packed feature index of gate 0
    // every read -- the slabs of the four gates, the cell state, the optional bias terms (behind uniform branches) -- is
    // requested before the first sum: with a gate's slab sum and its bias branches inside one loop body the compiler ran four
    // dependent rounds of loads (23 us per launch at 320 rows)
    f32x4 pre[4];
    const float* src = a.parts + (size_t)m * 4 * R + col;
    constexpr int NV = NP > 0 ? NP : 1;
    f32x4 v[4][NV];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int p = 0; p < NV; ++p) v[g][p] = ld4(src + g * 8 + (size_t)p * a.part_stride);
    const f32x4 cp0 = ld4(a.c_prev + (size_t)m * R + j);
    f32x4 bi[4], bh[4], gbv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bi[g] = bh[g] = gbv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.b_ih != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bi[g] = ld4(a.b_ih + g * R + j);
    }
    if (a.b_hh != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bh[g] = ld4(a.b_hh + g * R + j);
    }
    if (a.gate_bias != nullptr) {
        const float* gb = a.gate_bias + (size_t)(m / a.gb_div) * 4 * R + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) gbv[g] = ld4(gb + g * R);
    }
    f32x4 egv[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (a.emb_gate != nullptr) {                              // the embedded word's share of the gates: a table row, not a GEMM
        long long w = a.word[m];
        if (w < 0 || w >= a.V) w = 0;
        const float* eg = a.emb_gate + (size_t)w * 4 * R + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) egv[g] = ld4(eg + g * R);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        pre[g] = v[g][0];
#pragma unroll
        for (int p = 1; p < NV; ++p) pre[g] += v[g][p];
    }
    if constexpr (NP == 0) {
        for (int p = 1; p < a.nparts; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] += ld4(src + g * 8 + (size_t)p * a.part_stride);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] = (((pre[g] + bi[g]) + bh[g]) + gbv[g]) + egv[g];      // (an absent term adds an exact zero)
    const f32x4 cp = cp0;
    f32x4 hv, cv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = fast_sigmoid(pre[0][e]), fg = fast_sigmoid(pre[1][e]);
        const float gg = fast_tanh(pre[2][e]), og = fast_sigmoid(pre[3][e]);
        const float c2 = fg * cp[e] + ig * gg;
        cv[e] = c2;
        hv[e] = og * fast_tanh(c2);
    }
    st4(a.c_out + (size_t)m * R + j, cv);
    if (a.h_out != nullptr) st4(a.h_out + (size_t)m * R + j, hv);
    if (a.frag1 != nullptr) store_frag4(a.frag1, a.frag1_stride, m, j, hv);
    if (a.frag2 != nullptr) store_frag4(a.frag2, a.frag2_stride, m, j, hv);
}

// y[m, n] = sum_p parts[p][m, n] + bias[n] + bias2[n]      (vocabulary logits / gate_fc of the tile path)
template <int NP>
__global__ __launch_bounds__(256) void tile_linear_finish_kernel(const float* parts, int nparts, long long part_stride, int ld,
                                                                 const float* bias, const float* bias2, int M, int N, float* y,
                                                                 int ldy) {
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    const float* src = parts + (size_t)m * ld + n;
    float s;
    if constexpr (NP > 0) {
        float v[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = src[(size_t)p * part_stride];
        s = v[0];
#pragma unroll
        for (int p = 1; p < NP; ++p) s += v[p];
    } else {
        s = src[0];
        for (int p = 1; p < nparts; ++p) s += src[(size_t)p * part_stride];
    }
    if (bias != nullptr) s += bias[n];
    if (bias2 != nullptr) s += bias2[n];
    y[(size_t)m * ldy + n] = s;
}

// fp32 row-major [M, K] (optionally gathered rows, optional ReLU) -> activation fragments
__global__ __launch_bounds__(256) void tile_pack_rows_kernel(const float* x, int ldx, const int64_t* idx, int relu, int M, int K,
                                                             uint16_t* xb, long long mblk_stride) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int nq = K >> 2;
    if (q >= M * nq) return;
    const int m = q / nq, k = (q - m * nq) * 4;
    const int64_t r = idx != nullptr ? idx[m] : (int64_t)m;
    f32x4 v = ld4(x + (size_t)r * ldx + k);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    store_frag4(xb, mblk_stride, m, k, v);
}


// fp32 row-major [M, K], ANY K (scalar, bounds-checked loads; the k tail of the last 16-step is zero) -> activation fragments
__global__ __launch_bounds__(256) void tile_pack_rows_any_kernel(const float* x, long long ldx, int M, int K, uint16_t* xb,
                                                                 long long mblk_stride) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nq = (K + 3) >> 2;
    if (q >= (long long)M * nq) return;
    const int m = (int)(q / nq), k = (int)(q - (long long)m * nq) * 4;
    f32x4 v = {0, 0, 0, 0};
    const float* src = x + (size_t)m * ldx + k;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (k + e < K) v[e] = src[e];
    store_frag4(xb, mblk_stride, m, k, v);
}

// The same for K % 16 == 0 and 16-byte aligned rows, fragment-shaped: a wave = one 1 KiB fragment per term (32 rows x 16 k of one
// row block and k step) -- lane (k half, row) reads its 8 consecutive floats (a wave reads 64 contiguous bytes of each of 32 rows; the
// 4 waves of a workgroup take 4 consecutive k steps, i.e. 256 contiguous bytes per row) and stores 16 bytes per term, the wave 1 KiB
// contiguous.  The quad-per-thread kernels above scatter 8-byte pieces 512 bytes apart (1.4 TB/s on the 30 720 x 2048 operands of the
// encoder); rows beyond M inside the last block are written as zeros.
__global__ __launch_bounds__(256) void tile_pack_rows_blk_kernel(const float* x, long long ldx, int M, int K, uint16_t* xb,
                                                                 long long mblk_stride) {
    using u16x8 = __attribute__((ext_vector_type(8))) uint16_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mblk = blockIdx.y, ks = blockIdx.x * 4 + wave;
    if (ks >= (K >> 4)) return;
    const int row = mblk * 32 + (lane & 31), kh = lane >> 5;
    f32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
    if (row < M) {
        const float* src = x + (size_t)row * ldx + ks * 16 + kh * 8;
        v0 = ld4(src);
        v1 = ld4(src + 4);
    }
    u16x8 p[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        uint16_t h, mi, l;
        split3(e < 4 ? v0[e & 3] : v1[e & 3], h, mi, l);
        p[0][e] = h; p[1][e] = mi; p[2][e] = l;
    }
    uint16_t* base = xb + (size_t)mblk * mblk_stride + (size_t)ks * KSTEP + (kh * 32 + (lane & 31)) * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u16x8*>(base + pl * FRAG) = p[pl];
}

// fp32 row-major [S, C] read as its transpose: fragment rows = the C columns, contraction index = the S rows (dW = dY^T X and
// dX = dY W need one operand this way).  One wave per 32 columns x 16 rows: coalesced 128-byte row reads into a padded LDS
// tile, then lane (k half, column) gathers its 8 rows and writes 16 bytes per term.  Any S, C (zero fill).
__global__ __launch_bounds__(256) void tile_pack_cols_kernel(const float* x, long long ldx, int S, int C, uint16_t* xb,
                                                             long long mblk_stride) {
    __shared__ float tile[4][16][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = blockIdx.x;                                   // 32-column block
    const int ks = blockIdx.y * 4 + wave;                        // k step of 16 rows
    const int nks = (S + 15) >> 4;
    if (ks < nks) {
        const int c = lane & 31;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int r = (lane >> 5) + 2 * h;
            const int s = ks * 16 + r, col = cb * 32 + c;
            tile[wave][r][c] = (s < S && col < C) ? x[(size_t)s * ldx + col] : 0.f;
        }
    }
    __syncthreads();
    if (ks >= nks) return;
    const int kh = lane >> 5, i = lane & 31;
    using u16x8 = __attribute__((ext_vector_type(8))) uint16_t;
    u16x8 p[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        uint16_t h, mi, l;
        split3(tile[wave][kh * 8 + e][i], h, mi, l);
        p[0][e] = h; p[1][e] = mi; p[2][e] = l;
    }
    uint16_t* base = xb + (size_t)cb * mblk_stride + (size_t)ks * KSTEP + (kh * 32 + i) * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u16x8*>(base + pl * FRAG) = p[pl];
}

// After word / beam selection of step t: row r of step t + 1 continues hypothesis src = (r / beam) * beam + parent[r]
// (parent == nullptr: r itself).  Builds, in one pass over the four state tensors, everything step t + 1 reads:
//   c_att_prev / c_lang_prev [rows, R]  <- c_att / c_lang [src]
//   xa = [h_lang[src] | relu(Emb[word[r]]) | h_att[src]]   (att-LSTM input fragments, K = 2R + E)
//   xl third segment <- h_lang[src]                        (lang-LSTM input fragments, k offset 2R)
struct ReorderArgs {
    const int64_t* parent; const int64_t* word; int beam;
    const float* h_att; const float* c_att; const float* h_lang; const float* c_lang;   // [rows, R] of step t
    const float* table; int E; int V;
    float* c_att_prev; float* c_lang_prev;
    uint16_t* xa; long long xa_stride;
    uint16_t* xl_hlang; long long xl_stride;     // pointer at k step 2R / 16 of xl
    int rows, R;
};

__global__ __launch_bounds__(256) void tile_reorder_pack_kernel(ReorderArgs a) {
    // A wave = 32 rows x the two quads of one 8-k group (as in tile_lstm_finish_kernel): its fragment stores are 512 contiguous
    // bytes per term.  With k fastest across the wave they were 8-byte scatters over 96 lines per instruction (16.8 us).
    const int R = a.R, E = a.E;
    const int noct = (2 * R + E) >> 3;                               // 8-k groups per row (R, E multiples of 16)
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int hq = (int)(q & 1), i = (int)((q >> 1) & 31);
    const long long unit = q >> 6;                                    // (row block, 8-k group), group fastest
    const int oct = (int)(unit % noct), mb = (int)(unit / noct);
    const int r = mb * 32 + i, k = oct * 8 + hq * 4;
    if (r >= a.rows) return;
    const int src = a.parent != nullptr ? (r / a.beam) * a.beam + (int)a.parent[r] : r;
    if (k < R) {                                   // h_lang -> xa segment 0 and xl segment 2; c_lang rides along
        const f32x4 h = ld4(a.h_lang + (size_t)src * R + k);
        const f32x4 c = ld4(a.c_lang + (size_t)src * R + k);
        store_frag4(a.xa, a.xa_stride, r, k, h);
        store_frag4(a.xl_hlang, a.xl_stride, r, k, h);
        st4(a.c_lang_prev + (size_t)r * R + k, c);
    } else if (k < R + E) {
        int64_t w = a.word[r];
        if (w < 0 || w >= a.V) w = 0;
        f32x4 v = ld4(a.table + (size_t)w * E + (k - R));
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        store_frag4(a.xa, a.xa_stride, r, k, v);
    } else {
        const int kk = k - R - E;
        const f32x4 h = ld4(a.h_att + (size_t)src * R + kk);
        const f32x4 c = ld4(a.c_att + (size_t)src * R + kk);
        store_frag4(a.xa, a.xa_stride, r, k, h);
        st4(a.c_att_prev + (size_t)r * R + kk, c);
    }
}

}  // namespace

static int cvc_tile_loader_waves = 3;
// Rows are walked in chunks of 2 MH 32-row blocks (MH <= 5 accumulator tiles per wave).  Up to 320 rows: one chunk (the decode
// engine's beam rows).  More (the dense products of the training pass: 1 280 .. 8 192 rows): the chunk height that needs the least
// time by a two-term model -- workgroups run in rounds of one per CU (126 KB of LDS each) and a workgroup's time grows with its
// MH + a fixed part -- so that a launch does not end in a mostly empty round: dW of an LSTM weight block (8 192 x 2 048 outputs) is
// 16 x 26 = 416 workgroups at MH = 5 (two rounds, the second 62 % full) and 16 x 32 = 512 at MH = 4 (two full rounds of shorter
// workgroups): 0.37 -> see DESIGN.md section 4 of the split-product roof.
static int tile_rows_per_chunk(int mblk, int col_wgs) {
    if (mblk < 10) return (mblk + 1) / 2;
    if (mblk == 10) return 5;
    static const int cus = [] {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount < 1) return 256;
        return p.multiProcessorCount;
    }();
    int best = 5;
    double best_t = 1e30;
    for (int mh = 5; mh >= 3; --mh) {
        const long long wgs = (long long)col_wgs * ((mblk + 2 * mh - 1) / (2 * mh));
        const double t = (double)((wgs + cus - 1) / cus) * (mh + 0.6);
        if (t < best_t - 1e-9) { best_t = t; best = mh; }
    }
    return best;
}

// test / A-B hook: 0 = every wave copies its share of a stage (tile_gemm_kernel), 1 = dedicated loader waves + 8 computing waves,
// 2 = dedicated loader waves + 4 wide computing waves (tile_gemm_ld2_kernel), 3 = 2 for long K loops, 1 otherwise (default)
static int cvc_tile_big = 0, cvc_tile_big_min_wgs = 192;      // OFF by default (see below); min_wgs == 1: any grid (tests, A/B)
// test / A-B hook: 1 = the 256 x 256 form where it applies, 0 (default) = never; min_wgs > 0 sets the smallest grid it is taken for.
// Returns the previous on / off setting.
extern "C" int cvc_tile_gemm_big(int on, int min_wgs) {
    const int prev = cvc_tile_big;
    if (on >= 0) cvc_tile_big = on ? 1 : 0;
    if (min_wgs > 0) cvc_tile_big_min_wgs = min_wgs;
    return prev;
}

extern "C" int cvc_tile_gemm_loaders(int on) {
    const int prev = cvc_tile_loader_waves;
    if (on >= 0) cvc_tile_loader_waves = on > 4 ? 3 : on;
    return prev;
}

extern "C" int cvc_tile_gemm(const void* wb, const void* xb, long long x_mblk_stride, int K, int M, int N, int ksplit,
                             float* parts, int ld, long long part_stride, cvc_stream_t stream) {
    if (!wb || !xb || !parts || K < 16 || (K & 15) || M < 1 || N < 1 || ksplit < 1 || ksplit > K / 16 || ld < N) return CVC_E_BADARG;
    if (((uintptr_t)wb & 15) || ((uintptr_t)xb & 15) || (x_mblk_stride & 7)) return CVC_E_BADARG;
    TileArgs a;
    a.wb = (const uint16_t*)wb; a.xb = (const uint16_t*)xb; a.x_mblk_stride = x_mblk_stride; a.ksteps = K / 16;
    a.M = M; a.N = N; a.ntile = (N + 127) / 128; a.ksplit = ksplit; a.parts = parts; a.ld = ld; a.part_stride = part_stride;
    const int mblk = (M + 31) / 32;
    const hipStream_t st = (hipStream_t)stream;
    // the 256 x 256 form for large products made of whole tiles (see tile_gemm_big_kernel); cvc_tile_gemm_big(0) switches it off (A/B)
    // (measured, round 6: standalone with cold caches it wins on ONE nearly full round of workgroups -- dW of an LSTM block, 8192 x
    // 2048 outputs = 256 workgroups: 463 -> 413 us, a k step at 85 % of the power-limited MFMA time against 68 % -- and loses on grids
    // of many rounds -- GRU input projections 3.34 -> 3.70 ms: one workgroup per CU leaves every round's prologue and its 256 KB
    // epilogue uncovered, 11.25 rounds run as 12 -- and on grids below a round.  INSIDE the training step the six dW products it
    // applies to did not get faster (22 dense launches 3.69 -> 3.76 ms at config 3, 2.09 -> 2.25 at config 4's share): off by
    // default, cvc_tile_gemm_big(1, 0) selects it)
    const long long big_wgs = (long long)(M >> 8) * (N >> 8) * ksplit;
    if (cvc_tile_big != 0 && (M & 255) == 0 && (N & 255) == 0 && M >= 512 && big_wgs >= cvc_tile_big_min_wgs &&
        (big_wgs <= 256 || cvc_tile_big_min_wgs == 1)) {
        hipLaunchKernelGGL(tile_gemm_big_kernel, dim3((N >> 8) * ksplit, M >> 8), dim3(256), 0, st, a);
        return cvc_launch_status();
    }
    const int MH = tile_rows_per_chunk(mblk, a.ntile * ksplit);
    const int chunks = (mblk + 2 * MH - 1) / (2 * MH);
    const dim3 grid(a.ntile * ksplit, chunks);
#ifndef CVC_TILE_LOADERS
#define CVC_TILE_LOADERS 4
#endif
    // form 2 (4 wide computing waves) pays on long K loops (measured: lang / att gate GEMMs -2..3 us, the short-K vocabulary
    // head and h2attn +2 us each); the default picks per launch
    const bool wide = cvc_tile_loader_waves == 2 || (cvc_tile_loader_waves == 3 && a.ksteps / ksplit >= 64);
    if (CVC_TILE_LOADERS > 0 && cvc_tile_loader_waves == 4) {             // 4 wide computing waves + register-load loader waves (A/B)
        constexpr int NL = CVC_TILE_LOADERS > 0 ? CVC_TILE_LOADERS : 1;
        const dim3 blk((4 + NL) * 64);
        switch (MH) {
            case 1: hipLaunchKernelGGL((tile_gemm_ld2_kernel<1, NL, true>), grid, blk, 0, st, a); break;
            case 2: hipLaunchKernelGGL((tile_gemm_ld2_kernel<2, NL, true>), grid, blk, 0, st, a); break;
            case 3: hipLaunchKernelGGL((tile_gemm_ld2_kernel<3, NL, true>), grid, blk, 0, st, a); break;
            case 4: hipLaunchKernelGGL((tile_gemm_ld2_kernel<4, NL, true>), grid, blk, 0, st, a); break;
            default: hipLaunchKernelGGL((tile_gemm_ld2_kernel<5, NL, true>), grid, blk, 0, st, a); break;
        }
        return cvc_launch_status();
    }
    if (CVC_TILE_LOADERS > 0 && wide) {
        constexpr int NL = CVC_TILE_LOADERS > 0 ? CVC_TILE_LOADERS : 1;
        const dim3 blk((4 + NL) * 64);
        switch (MH) {
            case 1: hipLaunchKernelGGL((tile_gemm_ld2_kernel<1, NL>), grid, blk, 0, st, a); break;
            case 2: hipLaunchKernelGGL((tile_gemm_ld2_kernel<2, NL>), grid, blk, 0, st, a); break;
            case 3: hipLaunchKernelGGL((tile_gemm_ld2_kernel<3, NL>), grid, blk, 0, st, a); break;
            case 4: hipLaunchKernelGGL((tile_gemm_ld2_kernel<4, NL>), grid, blk, 0, st, a); break;
            default: hipLaunchKernelGGL((tile_gemm_ld2_kernel<5, NL>), grid, blk, 0, st, a); break;
        }
        return cvc_launch_status();
    }
    if (CVC_TILE_LOADERS > 0 && cvc_tile_loader_waves != 0) {
        constexpr int NL = CVC_TILE_LOADERS > 0 ? CVC_TILE_LOADERS : 1;
        const dim3 blk((8 + NL) * 64);
        switch (MH) {
            case 1: hipLaunchKernelGGL((tile_gemm_ld_kernel<1, NL>), grid, blk, 0, st, a); break;
            case 2: hipLaunchKernelGGL((tile_gemm_ld_kernel<2, NL>), grid, blk, 0, st, a); break;
            case 3: hipLaunchKernelGGL((tile_gemm_ld_kernel<3, NL>), grid, blk, 0, st, a); break;
            case 4: hipLaunchKernelGGL((tile_gemm_ld_kernel<4, NL>), grid, blk, 0, st, a); break;
            default: hipLaunchKernelGGL((tile_gemm_ld_kernel<5, NL>), grid, blk, 0, st, a); break;
        }
        return cvc_launch_status();
    }
    switch (MH) {
        case 1: hipLaunchKernelGGL(tile_gemm_kernel<1>, grid, dim3(512), 0, st, a); break;
        case 2: hipLaunchKernelGGL(tile_gemm_kernel<2>, grid, dim3(512), 0, st, a); break;
        case 3: hipLaunchKernelGGL(tile_gemm_kernel<3>, grid, dim3(512), 0, st, a); break;
        case 4: hipLaunchKernelGGL(tile_gemm_kernel<4>, grid, dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL(tile_gemm_kernel<5>, grid, dim3(512), 0, st, a); break;
    }
    return cvc_launch_status();
}

// The split of K and the grid cvc_tile_gemm launches for it, chosen HERE (callers used to derive the split from a row-chunk count of
// their own that the chunk-height choice above no longer matches): the largest split whose grid still is ONE round of workgroups
// (<= one per compute unit) with at least 8 k steps per slice; 1 when even the unsplit grid is more than a round.
extern "C" int cvc_tile_gemm_plan(int M, int N, int K, int* ksplit, int* chunk_rows, int* workgroups) {
    if (M < 1 || N < 1 || K < 16 || (K & 15)) return CVC_E_BADARG;
    const int mblk = (M + 31) / 32, ntile = (N + 127) / 128, ksteps = K / 16;
    const int cus = 256;
    int ks = ksteps / 8 < 1 ? 1 : ksteps / 8;
    if (ks > cus / ntile) ks = cus / ntile < 1 ? 1 : cus / ntile;
    int mh = 0, wgs = 0;
    for (; ks >= 1; --ks) {
        mh = tile_rows_per_chunk(mblk, ntile * ks);
        wgs = ntile * ks * ((mblk + 2 * mh - 1) / (2 * mh));
        if (wgs <= cus || ks == 1) break;
    }
    if (ksplit) *ksplit = ks;
    if (chunk_rows) *chunk_rows = 2 * mh * 32;
    if (workgroups) *workgroups = wgs;
    return 0;
}

extern "C" int cvc_tile_rows_alloc(int M) {
    // rows a fragment buffer must hold (zero beyond M) for ANY chunk height cvc_tile_gemm may pick for these rows
    const int mblk = (M + 31) / 32;
    if (mblk <= 10) {
        const int MH = mblk == 10 ? 5 : (mblk + 1) / 2;
        return 2 * MH * 32;
    }
    int rows = 0;
    for (int mh = 3; mh <= 5; ++mh) {
        const int r = (mblk + 2 * mh - 1) / (2 * mh) * 2 * mh * 32;
        rows = r > rows ? r : rows;
    }
    return rows;
}

static int tile_lstm_finish_impl(const float* parts, int nparts, long long part_stride, const float* b_ih, const float* b_hh,
                                 const float* gate_bias, int gb_div, const float* c_prev, int M, int R, float* c_out,
                                 float* h_out, void* frag1, long long frag1_stride, void* frag2, long long frag2_stride,
                                 const float* emb_gate, const int64_t* word, int V, cvc_stream_t stream);

extern "C" int cvc_tile_lstm_finish(const float* parts, int nparts, long long part_stride, const float* b_ih, const float* b_hh,
                                    const float* gate_bias, int gb_div, const float* c_prev, int M, int R, float* c_out,
                                    float* h_out, void* frag1, long long frag1_stride, void* frag2, long long frag2_stride,
                                    cvc_stream_t stream) {
    return tile_lstm_finish_impl(parts, nparts, part_stride, b_ih, b_hh, gate_bias, gb_div, c_prev, M, R, c_out, h_out, frag1, frag1_stride,
                                 frag2, frag2_stride, nullptr, nullptr, 0, stream);
}

extern "C" int cvc_tile_lstm_finish_embgate(const float* parts, int nparts, long long part_stride, const float* b_ih,
                                            const float* b_hh, const float* gate_bias, int gb_div, const float* emb_gate,
                                            const int64_t* word, int V, const float* c_prev, int M, int R, float* c_out,
                                            float* h_out, void* frag1, long long frag1_stride, void* frag2,
                                            long long frag2_stride, cvc_stream_t stream) {
    if (!emb_gate || !word || V < 1) return CVC_E_BADARG;
    return tile_lstm_finish_impl(parts, nparts, part_stride, b_ih, b_hh, gate_bias, gb_div, c_prev, M, R, c_out, h_out, frag1, frag1_stride,
                                 frag2, frag2_stride, emb_gate, word, V, stream);
}

static int tile_lstm_finish_impl(const float* parts, int nparts, long long part_stride, const float* b_ih, const float* b_hh,
                                 const float* gate_bias, int gb_div, const float* c_prev, int M, int R, float* c_out,
                                 float* h_out, void* frag1, long long frag1_stride, void* frag2, long long frag2_stride,
                                 const float* emb_gate, const int64_t* word, int V, cvc_stream_t stream) {
    if (!parts || nparts < 1 || !c_prev || !c_out || M < 1 || R < 16 || (R & 15) || gb_div < 1) return CVC_E_BADARG;
    LstmFinishArgs a;
    a.emb_gate = emb_gate; a.word = word; a.V = V;
    a.parts = parts; a.nparts = nparts; a.part_stride = part_stride; a.b_ih = b_ih; a.b_hh = b_hh; a.gate_bias = gate_bias;
    a.gb_div = gb_div; a.c_prev = c_prev; a.c_out = c_out; a.h_out = h_out; a.frag1 = (uint16_t*)frag1; a.frag1_stride = frag1_stride;
    a.frag2 = (uint16_t*)frag2; a.frag2_stride = frag2_stride; a.M = M; a.R = R;
    const long long n = (long long)((M + 31) / 32) * 32 * (R / 4);       // whole 32-row blocks (threads past M exit)
    const dim3 g((unsigned)((n + 255) / 256));
    switch (nparts) {
        case 1: hipLaunchKernelGGL(tile_lstm_finish_kernel<1>, g, dim3(256), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(tile_lstm_finish_kernel<2>, g, dim3(256), 0, (hipStream_t)stream, a); break;
        case 4: hipLaunchKernelGGL(tile_lstm_finish_kernel<4>, g, dim3(256), 0, (hipStream_t)stream, a); break;
        case 8: hipLaunchKernelGGL(tile_lstm_finish_kernel<8>, g, dim3(256), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(tile_lstm_finish_kernel<0>, g, dim3(256), 0, (hipStream_t)stream, a); break;
    }
    return cvc_launch_status();
}

extern "C" int cvc_tile_linear_finish(const float* parts, int nparts, long long part_stride, int ld, const float* bias,
                                      const float* bias2, int M, int N, float* y, int ldy, cvc_stream_t stream) {
    if (!parts || nparts < 1 || !y || M < 1 || N < 1 || ld < N || ldy < N) return CVC_E_BADARG;
    const dim3 g((N + 255) / 256, M);
#define CVC_LF(NP) hipLaunchKernelGGL(tile_linear_finish_kernel<NP>, g, dim3(256), 0, (hipStream_t)stream, parts, nparts, \
                                      part_stride, ld, bias, bias2, M, N, y, ldy)
    switch (nparts) {
        case 1: CVC_LF(1); break;
        case 2: CVC_LF(2); break;
        case 4: CVC_LF(4); break;
        case 6: CVC_LF(6); break;
        case 8: CVC_LF(8); break;
        case 16: CVC_LF(16); break;
        default: CVC_LF(0); break;
    }
#undef CVC_LF
    return cvc_launch_status();
}

extern "C" int cvc_tile_pack_rows(const float* x, int ldx, const int64_t* idx, int relu, int M, int K, void* xb,
                                  long long x_mblk_stride, cvc_stream_t stream) {
    if (!x || !xb || M < 1 || K < 16 || (K & 15) || (ldx & 3)) return CVC_E_BADARG;
    const long long n = (long long)M * (K / 4);
    hipLaunchKernelGGL(tile_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, relu,
                       M, K, (uint16_t*)xb, x_mblk_stride);
    return cvc_launch_status();
}

extern "C" int cvc_tile_pack_rows_any(const float* x, long long ldx, int M, int K, void* xb, long long x_mblk_stride,
                                      cvc_stream_t stream) {
    if (!x || !xb || M < 1 || K < 1 || ldx < K || (x_mblk_stride & 7)) return CVC_E_BADARG;
    if ((K & 15) == 0 && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0) {       // fragment-shaped form: whole 1 KiB stores
        const dim3 g((K / 16 + 3) / 4, (M + 31) / 32);
        hipLaunchKernelGGL(tile_pack_rows_blk_kernel, g, dim3(256), 0, (hipStream_t)stream, x, ldx, M, K, (uint16_t*)xb, x_mblk_stride);
        return cvc_launch_status();
    }
    const long long n = (long long)M * ((K + 3) / 4);
    hipLaunchKernelGGL(tile_pack_rows_any_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, M, K,
                       (uint16_t*)xb, x_mblk_stride);
    return cvc_launch_status();
}

extern "C" int cvc_tile_pack_cols(const float* x, long long ldx, int S, int C, void* xb, long long x_mblk_stride,
                                  cvc_stream_t stream) {
    if (!x || !xb || S < 1 || C < 1 || ldx < C || (x_mblk_stride & 7)) return CVC_E_BADARG;
    const dim3 g((C + 31) / 32, ((S + 15) / 16 + 3) / 4);
    hipLaunchKernelGGL(tile_pack_cols_kernel, g, dim3(256), 0, (hipStream_t)stream, x, ldx, S, C, (uint16_t*)xb, x_mblk_stride);
    return cvc_launch_status();
}

extern "C" int cvc_tile_reorder_pack(const int64_t* parent, const int64_t* word, int beam, const float* h_att, const float* c_att,
                                     const float* h_lang, const float* c_lang, const float* table, int E, int V,
                                     float* c_att_prev, float* c_lang_prev, void* xa, long long xa_stride, void* xl_hlang,
                                     long long xl_stride, int rows, int R, cvc_stream_t stream) {
    // E == 0 (table may then be null): xa = [h_lang | h_att] only -- the embedding-gate form, where the word enters through
    // cvc_tile_lstm_finish_embgate
    if (!word || !h_att || !c_att || !h_lang || !c_lang || (!table && E != 0) || !c_att_prev || !c_lang_prev || !xa || !xl_hlang)
        return CVC_E_BADARG;
    if (rows < 1 || beam < 1 || R < 16 || (R & 15) || E < 0 || (E & 15) || V < 1) return CVC_E_BADARG;
    ReorderArgs a;
    a.parent = parent; a.word = word; a.beam = beam; a.h_att = h_att; a.c_att = c_att; a.h_lang = h_lang; a.c_lang = c_lang;
    a.table = table; a.E = E; a.V = V; a.c_att_prev = c_att_prev; a.c_lang_prev = c_lang_prev; a.xa = (uint16_t*)xa;
    a.xa_stride = xa_stride; a.xl_hlang = (uint16_t*)xl_hlang; a.xl_stride = xl_stride; a.rows = rows; a.R = R;
    const long long n = (long long)((rows + 31) / 32) * 32 * ((2 * R + E) / 4);      // whole 32-row blocks (threads past `rows` exit)
    hipLaunchKernelGGL(tile_reorder_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}
