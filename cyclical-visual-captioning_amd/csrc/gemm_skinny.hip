// Weight-streaming skinny GEMM (M <= 64 rows) on the matrix cores (fp32 MFMA, or -- default -- fp32 products taken as
// exact bf16 splits on the bf16 MFMA, gemm_split.h), with the nn.LSTMCell pointwise
// update fused into its epilogue.  Reference call sites: model/decoder_core.py:45-50 (att-LSTM
// over cat[h_lang, fc, emb] + h_att), :59-61 (lang-LSTM over cat[ctx, h_att] + h_lang),
// model/modules.py:109 (h2attn), model/captioner.py:266 (logit).
//
// Shape regime (cfg2): M = 64 caption rows, K = 6144..9216, N = 8192 gate rows -> 201..302 MB of
// fp32 weights read once per step: HBM-bound by weight streaming, with the fp32 MFMA pipe
// (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD) just behind it.  Layout decisions:
//   * torch.cat is virtual: the K loop walks up to 4 (x, W-column-block) segments; the embedding
//     segment gathers table rows by word index and applies the ReLU on load.
//   * One workgroup owns 32 weight rows for ALL batch rows and the FULL K, so every weight byte
//     is fetched by exactly one CU exactly once; for the LSTM the 32 rows are the 4 gates of 8
//     hidden units, so c'/h' are finished in the epilogue and the [M,4R] gate matrix never
//     exists in HBM.
//   * The NW waves of a workgroup split K (chunk c -> wave c mod NW; a chunk = 32 k, i.e.
//     128 contiguous bytes per weight row: lanes 0-31 take k0..15, lanes 32-63 take k16..31
//     of the lane's row, matching the MFMA operand layout A[i=lane&31][k=lane>>5] without
//     shuffles; the k-order inside a chunk is permuted, which fp32 addition tolerates).
//     Partial 32 x M tiles are combined through LDS in a fixed order (deterministic).
#include "cvc_common.h"
#include "gemm_split.h"
#include <type_traits>

int cvc_gemm_split_mode = 2;      // see gemm_split.h

namespace {

constexpr int KC = 32;
constexpr int MAXSEG = 4;

struct GemmArgs {
    cvc_gemm_seg seg[MAXSEG];
    int prefix[MAXSEG + 1];   // cumulative chunk counts per segment
    int nsegs, total_chunks;
    int M, Nout, R;
    const float* bias;
    const float* bias2;
    const float* c_prev;
    const float* gate_bias;   // lstm: optional per-row additive gate term [M, 4R] (loop-invariant inputs hoisted)
    float* y;                 // linear: y [M, ldy]; lstm: h_out [M, R]
    float* c_out;
    float* gates_out;
    int ldy;
    int ksplit;               // linear only: blockIdx.y = K slice; slice sp writes y + sp * split_stride (bias in slice 0)
    long long split_stride;
    float* top2_part;         // linear only: per-(block,row) partial {top1 v,i, top2 v,i, max, sumexp} of the block's 32 columns
};

template <int MT>
struct Frag {
    f32x4 w[4];
    f32x4 x[MT][4];
};

template <int MT>
__device__ __forceinline__ void load_chunk(Frag<MT>& f, const GemmArgs& a, int it, const size_t (&wrow_off)[MAXSEG],
                                           const float* const (&xrow)[MT][MAXSEG], int kh) {
    int s = 0;
#pragma unroll
    for (int t = 1; t < MAXSEG; ++t)
        if (t < a.nsegs && it >= a.prefix[t]) s = t;
    const int c = it - a.prefix[s];
    const int k = a.seg[s].k;
    const float* wp = a.seg[s].w + wrow_off[s];
    const int kbase = c * KC + kh * 16;
    const bool relu = a.seg[s].relu != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = kbase + 4 * j;
        const bool ok = kk < k;
        f.w[j] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + kk)) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = ok ? ld4(xrow[mt][s] + kk) : f32x4{0, 0, 0, 0};
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            f.x[mt][j] = v;
        }
    }
}

template <int MT>
__device__ __forceinline__ void mma_chunk(const Frag<MT>& f, f32x16 (&acc)[MT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[j].x, f.x[mt][j].x, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[j].y, f.x[mt][j].y, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[j].z, f.x[mt][j].z, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[j].w, f.x[mt][j].w, acc[mt], 0, 0, 0);
        }
    }
}

// Combine the NW partial 32 x (MT*32) tiles through LDS in a fixed order (run-to-run
// deterministic) and finish: bias (+ LSTM cell update).  `red` needs min(NW,4)*32*(MT*32+1) floats,
// `scratch` 4*64*6 floats (word-selection partials).
template <int MT, int NW, bool LSTM>
__device__ __forceinline__ void combine_and_store(f32x16 (&acc)[MT], float* red, float* scratch, const GemmArgs& a) {
    constexpr int LDM = MT * 32 + 1;
    constexpr int NRED = NW > 4 ? 4 : NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const int M = a.M, R = a.R;
    // ---- combine the NW partial tiles (fixed order => run-to-run deterministic)
    // D layout of 32x32 MFMA: lane holds col = lane & 31 (batch), rows (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    auto spill = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                red[(slot * 32 + row) * LDM + mt * 32 + i] = acc[mt][r];
            }
    };
    if (NW > 4) {
        if (wave >= 4) spill(wave - 4);
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    acc[mt][r] += red[(wave * 32 + row) * LDM + mt * 32 + i];
                }
        }
        __syncthreads();
        if (wave < 4) spill(wave);
    } else {
        spill(wave);
    }
    __syncthreads();

    if (LSTM) {
        // unit u -> (batch row m, hidden jj in 0..7 fastest): 8 lanes share a 32-byte piece of a row-major state row
        // (with m fastest every lane touched its own line); LDS reads stay conflict-free (LDM is odd)
        const int j0 = blockIdx.x * 8;
        for (int u = tid; u < 8 * MT * 32; u += NW * 64) {
            const int jj = u & 7, m = u >> 3;
            if (m >= M) continue;
            const int j = j0 + jj;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NRED; ++w) v += red[(w * 32 + g * 8 + jj) * LDM + m];
                if (a.bias != nullptr) v += a.bias[g * R + j];
                if (a.bias2 != nullptr) v += a.bias2[g * R + j];
                if (a.gate_bias != nullptr) v += a.gate_bias[(size_t)m * 4 * R + g * R + j];
                pre[g] = v;
            }
            const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]);
            const float gg = fast_tanh(pre[2]), og = fast_sigmoid(pre[3]);
            const float c2 = fg * a.c_prev[(size_t)m * R + j] + ig * gg;
            a.c_out[(size_t)m * R + j] = c2;
            a.y[(size_t)m * R + j] = og * fast_tanh(c2);
            if (a.gates_out != nullptr) {
                float* go = a.gates_out + (size_t)m * 4 * R + j;
                go[0] = ig; go[R] = fg; go[2 * R] = gg; go[3 * R] = og;
            }
        }
    } else {
        // unit u -> (n local in 0..31 fastest, batch row m): coalesced stores along n
        const int n0 = blockIdx.x * 32;
        const bool lead = blockIdx.y == 0;                      // K slice 0 carries the bias
        float* y = a.y != nullptr ? a.y + (long long)blockIdx.y * a.split_stride : nullptr;
        if (y != nullptr) {
            for (int u = tid; u < 32 * MT * 32; u += NW * 64) {
                const int nl = u & 31, m = u >> 5;
                const int n = n0 + nl;
                if (m >= M || n >= a.Nout) continue;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NRED; ++w) v += red[(w * 32 + nl) * LDM + m];
                if (lead && a.bias != nullptr) v += a.bias[n];
                if (lead && a.bias2 != nullptr) v += a.bias2[n];
                y[(size_t)m * a.ldy + n] = v;
            }
        }
        if (a.top2_part != nullptr) {
            // fused word-selection partial (captioner.py:415-422 needs only the top-2 and the
            // log-sum-exp of a row).  Wave w scans columns 8w .. 8w+7 of the block for row m = lane
            // (LDS reads along m: conflict-free); the four wave records of a row are merged by wave 0.
            float v1 = -__builtin_inff(), v2 = -__builtin_inff(), mx = -__builtin_inff(), se = 0.f;
            int i1 = 0x7fffffff, i2 = 0x7fffffff;
            const int m = lane < MT * 32 ? lane : MT * 32 - 1;
            if (wave < 4) {
                for (int nl = wave * 8; nl < wave * 8 + 8; ++nl) {
                    const int n = n0 + nl;
                    if (n >= a.Nout) break;
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < NRED; ++w) v += red[(w * 32 + nl) * LDM + m];
                    if (a.bias != nullptr) v += a.bias[n];
                    if (a.bias2 != nullptr) v += a.bias2[n];
                    if (v > v1) { v2 = v1; i2 = i1; v1 = v; i1 = n; }   // n ascending: ties keep the lower index
                    else if (v > v2) { v2 = v; i2 = n; }
                    const float nm = fmaxf(mx, v);
                    se = se * __expf(mx - nm) + __expf(v - nm);
                    mx = nm;
                }
                float* r4 = scratch + ((size_t)wave * 64 + lane) * 6;
                r4[0] = v1; r4[1] = __int_as_float(i1); r4[2] = v2; r4[3] = __int_as_float(i2); r4[4] = mx; r4[5] = se;
            }
            __syncthreads();
            if (wave == 0 && lane < M) {
                for (int w = 1; w < 4; ++w) {                           // waves hold ascending column ranges
                    const float* r4 = scratch + ((size_t)w * 64 + lane) * 6;
                    const float u1 = r4[0], u2 = r4[2];
                    const int j1 = __float_as_int(r4[1]), j2 = __float_as_int(r4[3]);
                    if (u1 > v1) { if (v1 >= u2) { v2 = v1; i2 = i1; } else { v2 = u2; i2 = j2; } v1 = u1; i1 = j1; }
                    else if (u1 > v2) { v2 = u1; i2 = j1; }
                    const float nm = fmaxf(mx, r4[4]);
                    se = (nm == -__builtin_inff()) ? 0.f : se * __expf(mx - nm) + r4[5] * __expf(r4[4] - nm);
                    mx = nm;
                }
                float* rec = a.top2_part + ((size_t)blockIdx.x * 64 + lane) * 6;
                rec[0] = v1; rec[1] = __int_as_float(i1); rec[2] = v2; rec[3] = __int_as_float(i2); rec[4] = mx; rec[5] = se;
            }
        }
    }
}

// MT = number of 32-row batch tiles (M <= 32*MT); NW = waves per workgroup (K split)
template <int MT, int NW, bool LSTM>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_kernel(GemmArgs a) {
    constexpr int LDM = MT * 32 + 1;
    constexpr int NRED = NW > 4 ? 4 : NW;               // tiles resident in LDS at once
    __shared__ float red[NRED * 32 * LDM + 4 * 64 * 6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const int M = a.M, R = a.R;

    // this lane's weight row and batch rows
    int wrow;
    if (LSTM) wrow = (i >> 3) * R + blockIdx.x * 8 + (i & 7);
    else wrow = min((int)blockIdx.x * 32 + i, a.Nout - 1);
    if (!LSTM && a.ksplit > 1) {
        // K slice of this workgroup: whole chunks [lo, hi) of every segment
        a.prefix[0] = 0;
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) {
            if (s < a.nsegs) {
                const int nch = (a.seg[s].k + KC - 1) / KC;
                const int lo = nch * (int)blockIdx.y / a.ksplit, hi = nch * ((int)blockIdx.y + 1) / a.ksplit;
                const int klo = lo * KC, khi = min(hi * KC, a.seg[s].k);
                a.seg[s].w += klo;
                a.seg[s].x += klo;
                a.seg[s].k = max(khi - klo, 0);
                a.prefix[s + 1] = a.prefix[s] + (hi - lo);
            } else {
                a.prefix[s + 1] = a.prefix[s];
            }
        }
        a.total_chunks = a.prefix[a.nsegs];
    }
    size_t wrow_off[MAXSEG];
    const float* xrow[MT][MAXSEG];
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) {
        const int ss = s < a.nsegs ? s : 0;
        wrow_off[s] = (size_t)wrow * a.seg[ss].ldw;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(mt * 32 + i, M - 1);
            const int64_t r = a.seg[ss].idx != nullptr ? a.seg[ss].idx[m] : (int64_t)m;
            xrow[mt][s] = a.seg[ss].x + (size_t)r * a.seg[ss].ldx;
        }
    }

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    Frag<MT> fa, fb;
    int it = wave;
    const int total = a.total_chunks;
    if (it < total) load_chunk<MT>(fa, a, it, wrow_off, xrow, kh);
    while (it < total) {
        if (it + NW < total) load_chunk<MT>(fb, a, it + NW, wrow_off, xrow, kh);
        mma_chunk<MT>(fa, acc);
        it += NW;
        if (it >= total) break;
        if (it + NW < total) load_chunk<MT>(fa, a, it + NW, wrow_off, xrow, kh);
        mma_chunk<MT>(fb, acc);
        it += NW;
    }

    combine_and_store<MT, NW, LSTM>(acc, red, red + NRED * 32 * LDM, a);
}

// ------------------------------------------------------------------------------------------
// Fast path (every segment's k a multiple of 32): wave-private LDS rings, no workgroup barrier
// in the K loop.
//
// Why: the direct-load kernel above fetches MFMA fragments as 32 rows x 16 B per instruction and
// keeps the texture-address path busier than the matrix pipe (measured 196 us for the cfg2
// att-LSTM = 24 % of the fp32 MFMA rate).  A first LDS version with workgroup-wide 128-k stages and
// two barriers per stage reached 48 % (98 us): only one 16 KB weight stage in flight per CU and
// every wave stalling at the same barrier.  Here:
//   * 4 waves per workgroup (one per SIMD) split K by 32-k chunks (chunk c -> wave c mod 4), and
//     each wave owns a private 3-slot LDS ring: it copies ITS chunk (32 weight rows + MT*32
//     activation rows, 128 B = one full cache line per row) with LDS-DMA
//     (global_load_lds_dwordx4: 8 rows x 128 B per wave instruction), keeps two chunks in flight
//     (counted s_waitcnt vmcnt, ~32 KB of weights in flight per CU) and reads its fragments back
//     with ds_read_b128 -- wave-local ordering only, so waves drift apart and the MFMA pipe of a
//     SIMD never waits for a neighbour;
//   * LDS image is lane-linear per DMA instruction, so the bank-conflict fix sits on the SOURCE
//     address: 16-B slot s of row r holds k-quad s ^ ((r >> 1) & 7); the fragment read applies the
//     same XOR, which spreads the 16 rows of a ds_read_b128 lane group over 16 distinct slots;
//   * fragments of chunk j+1 are read while the 32 MFMAs of chunk j issue (two accumulators
//     alternate, so the 64-cycle v_mfma_f32_32x32x2_f32 issues back to back);
//   * the four 32 x M partial tiles meet in the ordered LDS reduction shared with the generic kernel.
#ifndef CVC_W_AUX
#define CVC_W_AUX 2   // weights are streamed once per launch by exactly one CU: non-temporal (aux = nt)
#endif
constexpr int RK = 32;                                     // k per chunk
constexpr int RING = 3;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int MT>
struct RFrag {
    f32x4 w[2];
    f32x4 x[MT][2];
};

template <int MT, bool LSTM, bool SPLIT>
__global__ __launch_bounds__(256) void skinny_gemm_ring_kernel(GemmArgs a) {
    constexpr int NW = 4;
    constexpr int ROWS = 32 + MT * 32;
    constexpr int STAGE = ROWS * RK * 4;                   // 8 / 12 KB
    constexpr int NLOAD = ROWS / 8;                        // DMA instructions per chunk: 8 / 12
    constexpr int RED_BYTES = NW * 32 * (MT * 32 + 1) * 4;
    constexpr int LDS_BYTES = NW * RING * STAGE > RED_BYTES + 6144 ? NW * RING * STAGE : RED_BYTES + 6144;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int M = a.M, R = a.R;
    char* ring = lds + wave * RING * STAGE;

    // ---- staging geometry: DMA instruction u of an operand covers rows 8u .. 8u+7 (128 B each);
    // this lane copies 16 B: row 8u + (lane >> 3), slot lane & 7 <- k-quad slot ^ ((row >> 1) & 7).
    const int prow = lane >> 3;
    const int kq0 = ((lane & 7) ^ (lane >> 4)) * 4;        // u even: (row >> 1) & 7 = lane >> 4
    const int kq1 = ((lane & 7) ^ (4 + (lane >> 4))) * 4;  // u odd

    // fragment addresses: row i (weights) / mt*32 + i (activations); this lane's 16 k of the chunk
    // are k-quads 4*kh .. 4*kh+3 -> physical slots q ^ ((i >> 1) & 7).  (A 32x32x2 MFMA wants
    // A[i][k = lane >> 5]: lanes 0-31 carry k 0..15 of the chunk, lanes 32-63 k 16..31.)
    const int fsw = (i >> 1) & 7;
    int fo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fo[q] = i * (RK * 4) + (((4 * kh + q) ^ fsw) & 7) * 16;

    auto read_frags = [&](RFrag<MT>(&f)[2], int slot) __attribute__((always_inline)) {
        const char* base = ring + slot * STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                f[h].w[p] = *reinterpret_cast<const f32x4*>(base + fo[2 * h + p]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    f[h].x[mt][p] = *reinterpret_cast<const f32x4*>(base + 4096 + mt * 4096 + fo[2 * h + p]);
            }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // relu (embedding segment only) is applied when the fragment is consumed
    auto mma = [&](const RFrag<MT>(&f)[2], auto relu_tag) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_tag)::value;
#if defined(CVC_ABL) && CVC_ABL == 2
        // ablation: no MFMA (memory pipeline only); keep the fragments live
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                asm volatile("" ::"v"(f[h].w[p]));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(f[h].x[mt][p]));
            }
        return;
#endif
        if constexpr (SPLIT) {
            // products as exact 3-way bf16 splits on the bf16 MFMA (see split8): quads (h, 0), (h, 1) = one K=16 step
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const Split3 W = split8(f[h].w[0], f[h].w[1]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 x0 = f[h].x[mt][0], x1 = f[h].x[mt][1];
                    if (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { x0[e] = fmaxf(x0[e], 0.f); x1[e] = fmaxf(x1[e], 0.f); }
                    }
                    const Split3 X = split8(x0, x1);
                    acc[mt] = mfma_bf16(W.mid, X.mid, acc[mt]);
                    acc[mt] = mfma_bf16(W.lo, X.hi, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, X.lo, acc[mt]);
                    acc[mt] = mfma_bf16(W.mid, X.hi, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, X.mid, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, X.hi, acc[mt]);
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[h].w[p][e], RELU ? fmaxf(f[h].x[mt][p][e], 0.f) : f[h].x[mt][p][e],
                                                                           acc[mt], 0, 0, 0);
        }
    };

    // ---- K loop: segment by segment (the decode engine passes ONE pre-concatenated segment, so the
    // pipeline below never restarts there); inside a segment everything is affine in the chunk index.
    for (int sg = 0; sg < a.nsegs; ++sg) {
        cvc_gemm_seg g = a.seg[sg];
        if (!LSTM && a.ksplit > 1) {                            // K slice of this workgroup (whole chunks)
            const int nch = g.k / RK;
            const int lo = nch * (int)blockIdx.y / a.ksplit, hi = nch * ((int)blockIdx.y + 1) / a.ksplit;
            g.w += lo * RK;
            g.x += lo * RK;
            g.k = (hi - lo) * RK;
        }
        const int nchunk = g.k / RK;
        const int n_my = nchunk > wave ? (nchunk - wave + NW - 1) / NW : 0;     // local chunks wave, wave+4, ...
        if (n_my == 0) continue;
        auto run_segment = [&](auto relu_tag) __attribute__((always_inline)) {
            // per-lane BYTE offsets of the rows this lane copies: 32-bit (checked on the host) so that the DMA
            // uses the scalar-base + vector-offset form and the K loop spends no VALU on addresses
            unsigned woff[4], xoff[4 * MT];
    #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int wrow = LSTM ? u * R + (int)blockIdx.x * 8 + prow : min((int)blockIdx.x * 32 + 8 * u + prow, a.Nout - 1);
                woff[u] = ((unsigned)wrow * (unsigned)g.ldw + ((u & 1) ? kq1 : kq0)) * 4u;
            }
    #pragma unroll
            for (int u = 0; u < 4 * MT; ++u) {
                const int m = min(8 * u + prow, M - 1);
                const unsigned xr = g.idx != nullptr ? (unsigned)g.idx[m] : (unsigned)m;
                xoff[u] = (xr * (unsigned)g.ldx + ((u & 1) ? kq1 : kq0)) * 4u;
            }
            const char* wp = reinterpret_cast<const char*>(g.w + wave * RK);   // chunk c of this wave: k = (wave + c*NW) * RK
            const char* xp = reinterpret_cast<const char*>(g.x + wave * RK);

            // issue the 4 + 4*MT DMA instructions of this wave's local chunk c into ring slot c % RING
            auto stage = [&](int c) __attribute__((always_inline)) {
    #if defined(CVC_ABL) && CVC_ABL == 1
                if (c >= 0) return;       // ablation: no DMA (MFMA + LDS only)
    #endif
                char* base = ring + (c % RING) * STAGE;
                const char* wc = wp + (size_t)c * (NW * RK * 4);
                const char* xc = xp + (size_t)c * (NW * RK * 4);
    #pragma unroll
                for (int u = 0; u < 4; ++u)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wc + woff[u]), (lds_ptr_t)(base + u * 1024), 16, 0, CVC_W_AUX);
    #pragma unroll
                for (int u = 0; u < 4 * MT; ++u)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(xc + xoff[u]), (lds_ptr_t)(base + 4096 + u * 1024), 16, 0, 0);
            };

            // one steady-state step: chunk c is in `cur`; put chunk c+2 in flight, pull chunk c+1 into `nxt`,
            // multiply chunk c.  The directives ask the scheduler to issue the DMA / address math and the
            // LDS reads in the shadow of the 64-cycle MFMAs instead of serialising them after the block.
            auto step = [&](const RFrag<MT>(&cur)[2], RFrag<MT>(&nxt)[2], int c) __attribute__((always_inline)) {
                stage(c + 2);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
                read_frags(nxt, (c + 1) % RING);
                mma(cur, relu_tag);
                if constexpr (SPLIT) {                                   // 12 MT MFMAs, ~108 MT split VALU ops, 2 NLOAD memory ops
    #pragma unroll
                    for (int q = 0; q < NLOAD; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);   // VALU (operand split)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
                    }
    #pragma unroll
                    for (int q = 0; q < NLOAD; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);   // VALU
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    }
                } else {
    #pragma unroll
                    for (int q = 0; q < NLOAD; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
                    }
    #pragma unroll
                    for (int q = 0; q < NLOAD; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                    }
                }
            };

            RFrag<MT> fa[2], fb[2];
            // the previous segment's LDS reads are complete (consumed by its MFMAs): slots are free
            stage(0);
            if (n_my > 1) stage(1);
            if (n_my > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_frags(fa, 0);
            int c = 0;
            for (; c + 3 < n_my; c += 2) {       // steady state, unrolled by two: fragment registers are statically named
                step(fa, fb, c);
                step(fb, fa, c + 1);
            }
            while (c < n_my) {                   // drain (at most 3 chunks left; chunk c is in fa)
                if (c + 2 < n_my) stage(c + 2);
                if (c + 1 < n_my) {
                    if (c + 2 < n_my) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    read_frags(fb, (c + 1) % RING);
                }
                mma(fa, relu_tag);
                ++c;
                if (c >= n_my) break;
                if (c + 2 < n_my) stage(c + 2);
                if (c + 1 < n_my) {
                    if (c + 2 < n_my) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    read_frags(fa, (c + 1) % RING);
                }
                mma(fb, relu_tag);
                ++c;
            }
        };
        if (g.relu) run_segment(std::true_type{}); else run_segment(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                       // rings are dead: LDS becomes the reduction buffer
    combine_and_store<MT, NW, LSTM>(acc, reinterpret_cast<float*>(lds), reinterpret_cast<float*>(lds + RED_BYTES), a);
}

int fill_args(GemmArgs& a, const cvc_gemm_seg* segs, int nsegs, int chunk) {
    if (nsegs < 1 || nsegs > MAXSEG) return CVC_E_BADARG;
    a.nsegs = nsegs;
    a.prefix[0] = 0;
    for (int s = 0; s < MAXSEG; ++s) {
        if (s < nsegs) {
            const cvc_gemm_seg& g = segs[s];
            if (g.k < 4 || (g.k & 3) || (g.ldx & 3) || (g.ldw & 3) || !g.x || !g.w) return CVC_E_BADARG;
            if (((uintptr_t)g.x & 15) || ((uintptr_t)g.w & 15)) return CVC_E_BADARG;
            a.seg[s] = g;
            a.prefix[s + 1] = a.prefix[s] + (g.k + chunk - 1) / chunk;
        } else {
            a.seg[s] = segs[0];
            a.prefix[s + 1] = a.prefix[s];
        }
    }
    a.total_chunks = a.prefix[nsegs];
    return 0;
}

// LDS-DMA fast path needs whole 32-k chunks (full 128-B lines per row) and at most one gather segment
bool fast_ok(const cvc_gemm_seg* segs, int nsegs, int Nout_rows) {
    int gathers = 0;
    for (int s = 0; s < nsegs; ++s) {
        if (segs[s].k % RK) return false;
        if ((segs[s].ldx & 3) || (segs[s].ldw & 3)) return false;
        if ((long long)Nout_rows * segs[s].ldw >= (1LL << 32) - 64) return false;   // 32-bit element offsets
        gathers += segs[s].idx != nullptr;
    }
    return gathers <= 1;
}

template <bool LSTM>
int launch(const GemmArgs& a, int blocks, bool fast, hipStream_t st) {
    if (a.M > 64) return CVC_E_TOOBIG;
    const dim3 grid(blocks, LSTM || a.ksplit < 1 ? 1 : a.ksplit);
    if (fast && cvc_gemm_split_mode) {
        if (a.M <= 32) hipLaunchKernelGGL((skinny_gemm_ring_kernel<1, LSTM, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_ring_kernel<2, LSTM, true>), grid, dim3(256), 0, st, a);
    } else if (fast) {
        if (a.M <= 32) hipLaunchKernelGGL((skinny_gemm_ring_kernel<1, LSTM, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_ring_kernel<2, LSTM, false>), grid, dim3(256), 0, st, a);
    } else {
        if (a.M <= 32) hipLaunchKernelGGL((skinny_gemm_kernel<1, 8, LSTM>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_kernel<2, 8, LSTM>), grid, dim3(512), 0, st, a);
    }
    return cvc_launch_status();
}

int g_force_generic = 0;   // test hook: cvc_gemm_force_generic(1) routes everything to the direct-load kernel

}  // namespace

static int linear_impl(const cvc_gemm_seg* segs, int nsegs, const float* bias, const float* bias2, int M, int Nout,
                       float* y, int ldy, int ksplit, long long split_stride, float* top2_part, hipStream_t st) {
    if (M < 1 || Nout < 1 || (y == nullptr && top2_part == nullptr) || (y != nullptr && ldy < Nout)) return CVC_E_BADARG;
    if (nsegs < 1 || nsegs > MAXSEG || ksplit < 1) return CVC_E_BADARG;
    if ((ksplit > 1 || top2_part != nullptr) && M > 64) return CVC_E_TOOBIG;
    // M > 64: walk the batch in 64-row slabs (weights are re-streamed per slab; the greedy decode
    // path never gets here, beam/training shapes do until the wide kernel lands)
    for (int m0 = 0; m0 < M; m0 += 64) {
        GemmArgs a{};
        cvc_gemm_seg tmp[MAXSEG];
        for (int s = 0; s < nsegs; ++s) {
            tmp[s] = segs[s];
            if (tmp[s].idx) tmp[s].idx += m0; else tmp[s].x += (size_t)m0 * tmp[s].ldx;
            if (ksplit > 1 && tmp[s].idx != nullptr) return CVC_E_BADARG;
        }
        const bool fast = !g_force_generic && fast_ok(tmp, nsegs, Nout);
        int rc = fill_args(a, tmp, nsegs, fast ? RK : KC);
        if (rc) return rc;
        a.M = M - m0 < 64 ? M - m0 : 64; a.Nout = Nout; a.R = 0;
        a.bias = bias; a.bias2 = bias2; a.c_prev = nullptr; a.gate_bias = nullptr;
        a.y = y ? y + (size_t)m0 * ldy : nullptr; a.c_out = nullptr; a.gates_out = nullptr; a.ldy = ldy;
        a.ksplit = ksplit; a.split_stride = split_stride; a.top2_part = top2_part;
        rc = launch<false>(a, (Nout + 31) / 32, fast, st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cvc_linear_fwd(const cvc_gemm_seg* segs, int nsegs, const float* bias, const float* bias2,
                              int M, int Nout, float* y, int ldy, cvc_stream_t stream) {
    return linear_impl(segs, nsegs, bias, bias2, M, Nout, y, ldy, 1, 0, nullptr, (hipStream_t)stream);
}

extern "C" int cvc_linear_splitk_fwd(const cvc_gemm_seg* segs, int nsegs, const float* bias, int M, int Nout,
                                     int ksplit, float* y_parts, cvc_stream_t stream) {
    if (M > 64 || ksplit < 1 || ksplit > 64) return CVC_E_BADARG;
    return linear_impl(segs, nsegs, bias, nullptr, M, Nout, y_parts, Nout, ksplit, (long long)M * Nout, nullptr,
                       (hipStream_t)stream);
}

extern "C" int cvc_linear_top2_fwd(const cvc_gemm_seg* segs, int nsegs, const float* bias, int M, int Nout,
                                   float* y_or_null, float* top2_part, cvc_stream_t stream) {
    if (top2_part == nullptr) return CVC_E_BADARG;
    return linear_impl(segs, nsegs, bias, nullptr, M, Nout, y_or_null, Nout, 1, 0, top2_part, (hipStream_t)stream);
}

extern "C" int cvc_lstm_cell_fwd(const cvc_gemm_seg* segs, int nsegs, const float* b_ih, const float* b_hh,
                                 const float* gate_bias, const float* c_prev, int M, int R, float* h_out,
                                 float* c_out, float* gates_out, cvc_stream_t stream) {
    if (M < 1 || R < 8 || (R & 7) || !c_prev || !h_out || !c_out) return CVC_E_BADARG;
    for (int m0 = 0; m0 < M; m0 += 64) {
        GemmArgs a{};
        cvc_gemm_seg tmp[MAXSEG];
        if (nsegs < 1 || nsegs > MAXSEG) return CVC_E_BADARG;
        for (int s = 0; s < nsegs; ++s) {
            tmp[s] = segs[s];
            if (tmp[s].idx) tmp[s].idx += m0; else tmp[s].x += (size_t)m0 * tmp[s].ldx;
        }
        const bool fast = !g_force_generic && fast_ok(tmp, nsegs, 4 * R);
        int rc = fill_args(a, tmp, nsegs, fast ? RK : KC);
        if (rc) return rc;
        a.M = M - m0 < 64 ? M - m0 : 64; a.Nout = 4 * R; a.R = R;
        a.bias = b_ih; a.bias2 = b_hh; a.c_prev = c_prev + (size_t)m0 * R;
        a.gate_bias = gate_bias ? gate_bias + (size_t)m0 * 4 * R : nullptr;
        a.y = h_out + (size_t)m0 * R; a.c_out = c_out + (size_t)m0 * R;
        a.gates_out = gates_out ? gates_out + (size_t)m0 * 4 * R : nullptr; a.ldy = R;
        a.ksplit = 1; a.split_stride = 0; a.top2_part = nullptr;
        rc = launch<true>(a, R / 8, fast, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cvc_gemm_packed_split(int on) {
    const int prev = cvc_gemm_split_mode;
    if (on >= 0) cvc_gemm_split_mode = on > 2 ? 2 : on;       // negative: query only
    return prev;
}

extern "C" int cvc_gemm_force_generic(int on) {
    const int prev = g_force_generic;
    g_force_generic = on ? 1 : 0;
    return prev;
}
