// Score pass shared by the attention forward, the attention backward (d_attn = d_ctx . ctx_n is
// the same streaming dot product) and the grounder (captioner.py:154-158).  See attn_fwd.hip.
#pragma once
#include "cvc_common.h"
#ifdef CVC_EXPERIMENTAL
#include "gsk.h"          // the stream-K query form (cvc_attn_scores_qslab) exists in experimental builds only
#endif
#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef CVC_SCORE_ROWS
#define CVC_SCORE_ROWS 32
#endif
constexpr int ROWS_PER_WG = CVC_SCORE_ROWS;   // score kernel: rows per workgroup (4 waves, interleaved) -- the one-query default
constexpr int ROWS_PER_WG_MAX = 128;          // 32 mask bits per wave
constexpr int SCORE_WG = 256;

struct ScoreArgs {
    cvc_attn_set set[2];
    int chunks0;                   // blockIdx.x < chunks0 -> set 0 else set 1
    const float* q;                // [q_nparts][nclip*nq, A]: partial sums of the query GEMM (split-K), summed on load
    const float* q_bias;           // [A] added once (h2attn.bias) or null
    int q_nparts;
    long long q_part_stride;
    long long q_ld;                // floats between the rows of q (A when dense; the K-slice planes of a backward-data product are wider)
    const float* w_a;              // [A] (additive)
    const float* b_a;              // device scalar (alpha_net.bias) or null
    float inv_temp;
    int nq, A;                     // nq = queries handled by this launch
    int nq_total, q0;              // row = clip * nq_total + q0 + qi
    int rows_per_wg;               // feature rows a workgroup scores (chosen per launch, see score_rows_per_wg)
#ifdef CVC_EXPERIMENTAL
    int q_from_slab;               // the query is the sum of a stream-K group's partial tiles (gsk.h) instead of q / q_nparts
    GskSegs q_slab;
#endif
};

// QG = queries per group: the group's accumulators are independent chains for the VALU and its LDS reads / wave reductions are
// issued back to back.  The query list is padded to a multiple of QG in LDS (zero rows), so the group body has NO branches:
// with a per-query `if` the compiler fences every 16-byte query read behind its own s_waitcnt (measured at beam 5: 87 us).
//
// Additive form, instruction count per (row, query, element) -- the pass is VALU-issue-bound with several queries per clip (PMC at
// beam 5: the vector ALU busy 68 of 80 us, ~4.8 cycles per instruction whatever its kind): tanh(p + q) = 1 - 2 / (1 + 2^(C (p + q))),
// C = 2 log2 e, with C q stored in LDS (scaled once per workgroup) and C p formed once per row element for all of its queries:
// add, exp2, add, rcp, fma, fma = 6 instructions (the expf(2x) form compiled to 8).
//
// Round 6 -- the FACTORED form for several queries per clip (beam search): 2^(C (p + q)) = 2^(C p) * 2^(C q).  2^(C q) is formed once
// per workgroup in LDS, 2^(C p) once per row element for all of the row's queries, so an element-query costs
//     d = fma(ep, eq, 1)   r = rcp(d)   acc = fma(-2 w, r, acc)              score = sum(w) + acc
// -- ONE quarter-rate transcendental and two packed full-rate operations instead of two and four (the pass is bound by the
// transcendental pipe: 16 of every 20 issue cycles per element-query before).  Exactness of the factoring: taken only when every
// |C q| of the workgroup is <= 30 (|q| <= 10.4; checked while the queries are staged, one __syncthreads_or); C p is clamped to
// +-62 -- a clamped p with such a q has |p + q| >= 11, where tanh is +-1 to the last fp32 bit in the direct form as well, and the
// factors stay inside [2^-92, 2^92]: no overflow, no denormal, no inf * 0.  A workgroup with a larger query runs the direct form
// below, unchanged.  (Round 2 tried the factoring with a per-element fallback inside one loop body: 186 VGPRs, slower.  Here the
// choice is workgroup-uniform and the two forms are separate loops.)
template <int NPL>
__device__ __forceinline__ f32x4 sum_planes_fixed(const float* src, long long stride) {
    f32x4 pv[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) pv[p] = ld4(src + (size_t)p * stride);
    f32x4 v = pv[0];
#pragma unroll
    for (int p = 1; p < NPL; ++p) v += pv[p];
    return v;
}
// plane 0 + plane 1 + ... (in that order, whatever the count)
__device__ __forceinline__ f32x4 sum_planes(const float* src, int n, long long stride) {
    switch (n) {
        case 1: return ld4(src);
        case 2: return sum_planes_fixed<2>(src, stride);
        case 4: return sum_planes_fixed<4>(src, stride);
        case 8: return sum_planes_fixed<8>(src, stride);
        case 16: return sum_planes_fixed<16>(src, stride);
        default: break;
    }
    f32x4 v = ld4(src);
    for (int p = 1; p < n; ++p) v += ld4(src + (size_t)p * stride);
    return v;
}

template <int KIND, int NCH, int QG>
__global__ __launch_bounds__(SCORE_WG) void attn_scores_kernel(ScoreArgs a) {
    constexpr bool FACTORABLE = KIND == CVC_ATTN_ADDITIVE && QG > 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nq_pad = (a.nq + QG - 1) / QG * QG;
    float* q_s = smem;                         // [nq_pad][A]
    float* w_s = smem + (size_t)nq_pad * a.A;  // [A]
    constexpr float EXP_C = 2.8853900817779268f;          // 2 log2(e)
    const int clip = blockIdx.y;
    const int s = (int)blockIdx.x < a.chunks0 ? 0 : 1;
    const int chunk = s == 0 ? blockIdx.x : blockIdx.x - a.chunks0;
    const cvc_attn_set& S = a.set[s];
    const int A = a.A, nq = a.nq, n = S.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // what a masked position is filled with: -1e8 (modules.py:42-46, 125-129), or -inf for with_sentinel=True (modules.py:40-41,
    // 123-124; cvc_attn_set.stream bit 2)
    const float fill = (S.stream & 4) ? -__builtin_inff() : CVC_MIN_VALUE;
    // alpha_net's bias, read ONCE: inside the row loop's store branch the compiler re-read it from memory for every query of every
    // row (it may alias the scores it stores) -- five dependent trips to L2 per row at beam 5
    const float bias_a = (KIND == CVC_ATTN_ADDITIVE && a.b_a != nullptr) ? a.b_a[0] : 0.f;

    int q_big = 0;                              // this thread staged a query element with |C q| > 30
    for (int i = tid * 4; i < nq_pad * A; i += SCORE_WG * 4) {
        f32x4 v = {0, 0, 0, 0};
#ifdef CVC_EXPERIMENTAL
        if (i < nq * A && a.q_from_slab) {
            // column block col / 32 of the h2attn group: tile = blk / 8, its segments summed in order; row = the query's batch row
            const int col = i % A, row = clip * a.nq_total + a.q0 + i / A, blk = col >> 5;
            const int nseg = gsk_nseg(a.q_slab, blk >> 3);
            const float* src = gsk_part(a.q_slab, blk >> 3, 0, blk & 7) + (size_t)row * 32 + (col & 31);
            v = ld4(src);
            for (int p = 1; p < nseg; ++p) v += ld4(src + (size_t)p * (8 * 2048));
            if (a.q_bias != nullptr) v += ld4(a.q_bias + col);
        } else
#endif
        if (i < nq * A) {
            const float* src = a.q + ((size_t)clip * a.nq_total + a.q0 + i / A) * a.q_ld + i % A;
            // the K-slice planes of the h2attn product, summed in plane order: all of a thread's loads are requested before the first
            // add (as a loop over a run-time count they were q_nparts dependent trips to L2 -- 8 in the greedy decode -- in front of
            // every workgroup's first feature row, and a launch is ONE round of workgroups: nothing covered them)
            v = sum_planes(src, a.q_nparts, a.q_part_stride);
            if (a.q_bias != nullptr) v += ld4(a.q_bias + (i % A));
        }
        if (KIND == CVC_ATTN_ADDITIVE) v *= EXP_C;
        if constexpr (FACTORABLE)
            q_big |= !(fabsf(v.x) <= 30.f && fabsf(v.y) <= 30.f && fabsf(v.z) <= 30.f && fabsf(v.w) <= 30.f);      // (NaN counts as big)
        st4(q_s + i, v);
    }
    if (KIND == CVC_ATTN_ADDITIVE)
        for (int i = tid * 4; i < A; i += SCORE_WG * 4) st4(w_s + i, ld4(a.w_a + i));
    bool factored = false;
    if constexpr (FACTORABLE) {
#ifndef CVC_SCORE_NO_FACTORED
        factored = __syncthreads_or(q_big) == 0;
#else
        __syncthreads();
#endif
        if (factored) {                        // every thread turns the elements IT staged into 2^(C q)
            for (int i = tid * 4; i < nq_pad * A; i += SCORE_WG * 4) {
                f32x4 v = ld4(q_s + i);
                v.x = __builtin_amdgcn_exp2f(v.x); v.y = __builtin_amdgcn_exp2f(v.y);
                v.z = __builtin_amdgcn_exp2f(v.z); v.w = __builtin_amdgcn_exp2f(v.w);
                st4(q_s + i, v);
            }
            for (int i = tid * 4; i < A; i += SCORE_WG * 4) st4(w_s + i, ld4(w_s + i) * -2.0f);       // (its own elements: -2 w)
            __syncthreads();
        }
    } else {
        __syncthreads();
    }

    const int row0 = chunk * a.rows_per_wg;
    const int row_end = min(n, row0 + a.rows_per_wg);
    const float* P = S.proj + (size_t)clip * n * A;

    // a set marked `stream` is read with the non-temporal policy so that it does not displace what the decode loop
    // re-reads every step from the 256 MB Infinity Cache (the caller budgets that: cvc/decode.py)
    // the QG scores of a row (uniform values: wave_sum broadcasts) leave in ONE store: lane u writes query q0 + u
    auto emit = [&](float sc, int q0, int r, bool masked) __attribute__((always_inline)) {      // sc: lane u holds query q0 + u's score
        if (lane < QG && q0 + lane < nq) {
            if (masked) sc = fill;
            const size_t o = ((size_t)clip * a.nq_total + a.q0 + q0 + lane) * n + r;
            S.scores[o] = sc;
            if (S.frame_masked != nullptr)
                S.frame_masked[o] = S.frame_mask[o] != 0 ? fill : sc;
        }
    };
    auto rows = [&](auto stream_tag, auto fact_tag, auto exact_tag) __attribute__((always_inline)) {
        constexpr bool STREAM = decltype(stream_tag)::value;
        constexpr bool FACT = decltype(fact_tag)::value;
        // EXACT: A == NCH * 256 (1024, 2048: the usual widths) as a compile-time constant -- no column guards, and the LDS reads of
        // the staged queries take immediate offsets instead of one address add each
        constexpr bool EXACT = decltype(exact_tag)::value;
        const int A = EXACT ? NCH * 256 : a.A;
#define LDF(ptr) (STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr)) : ld4(ptr))
    // Every load of the row loop is issued UNCONDITIONALLY (clamped addresses instead of guards): with a load under a
    // divergent `col < A` or a `next row exists` branch, hipcc's s_waitcnt insertion gives up counting and waits vmcnt(0)
    // right after the prefetch is issued -- the prefetch then overlaps nothing (the score pass sat at 31 us / 87 us for
    // 1 / 5 queries with most wave cycles in s_waitcnt).  Columns beyond A load a valid address and are zeroed by select.
    f32x4 cur[NCH], nxt[NCH];
    int colc[NCH];
    bool cok[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 4;
        cok[j] = EXACT || col < A;
        colc[j] = cok[j] ? col : A - 4;
    }
    int r = row0 + wave;
    if (r >= row_end) return;
    // region mask bits of this wave's rows (rows r, r + 4, ...), fetched before the pipelined loop: lane k reads the byte of the
    // wave's k-th row and a ballot collects them -- ONE load per wave (a loop of dependent byte loads, one per row, cost a wave
    // of the region set 10 - 25 us before its first feature row: each iteration waited out a trip to L2 / HBM)
    unsigned mbits = 0;
    if (S.mask != nullptr) {
        const int rr = r + 4 * lane;
        const bool m = rr < row_end && S.mask[(size_t)clip * n + (rr < row_end ? rr : r)] != 0;
        mbits = (unsigned)__builtin_amdgcn_ballot_w64(m);          // <= 32 rows per wave (ROWS_PER_WG_MAX)
    }
    // factored form: sum of this wave's alpha_net weights (score = sum(w) - 2 sum(w r))
    float w_total = 0.f;
    if constexpr (FACT) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const f32x4 w4 = ld4(w_s + colc[j]);                  // (-2 w in the factored form)
            if (cok[j]) w_total += (w4.x + w4.y) + (w4.z + w4.w);
        }
        w_total = -0.5f * wave_sum(w_total);
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) cur[j] = LDF(P + (size_t)r * A + colc[j]);
    // one row: request the next row into `nxt`, score `cur`.  The row loop calls it with the two buffers' roles swapped every other
    // row instead of copying nxt -> cur (16 - 32 register moves per row)
    auto step = [&](f32x4 (&cur)[NCH], f32x4 (&nxt)[NCH], int r, int k) __attribute__((always_inline)) {
#if defined(CVC_SC_ABL) && CVC_SC_ABL == 2
        const int rn = r;
#else
        const int rn = min(r + 4, row_end - 1);          // past the end: re-read the last row (unused)
#endif
#pragma unroll
        for (int j = 0; j < NCH; ++j) nxt[j] = LDF(P + (size_t)rn * A + colc[j]);
        const bool masked = (mbits >> k) & 1u;
        if constexpr (FACT) {
            using f32x2 = __attribute__((ext_vector_type(2))) float;
            // 2^(C p) of the row's elements, once for all queries; -2 w beside it
            f32x4 ep[NCH], w2[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const f32x4 x = cur[j] * EXP_C;
                ep[j].x = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x.x, -62.f, 62.f));
                ep[j].y = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x.y, -62.f, 62.f));
                ep[j].z = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x.z, -62.f, 62.f));
                ep[j].w = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x.w, -62.f, 62.f));
                w2[j] = cok[j] ? ld4(w_s + colc[j]) : f32x4{0, 0, 0, 0};
            }
            for (int q0 = 0; q0 < nq; q0 += QG) {
                f32x2 acc2[QG];
                float acc[QG];
#pragma unroll
                for (int u = 0; u < QG; ++u) acc2[u] = f32x2{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
#pragma unroll
                    for (int u = 0; u < QG; ++u) {
                        const f32x4 eq = ld4(q_s + (q0 + u) * A + colc[j]);
                        const f32x2 d01 = __builtin_elementwise_fma(f32x2{ep[j].x, ep[j].y}, f32x2{eq.x, eq.y}, f32x2{1.0f, 1.0f});
                        const f32x2 d23 = __builtin_elementwise_fma(f32x2{ep[j].z, ep[j].w}, f32x2{eq.z, eq.w}, f32x2{1.0f, 1.0f});
                        const f32x2 r01 = {fast_rcp(d01.x), fast_rcp(d01.y)}, r23 = {fast_rcp(d23.x), fast_rcp(d23.y)};
                        acc2[u] = __builtin_elementwise_fma(f32x2{w2[j].x, w2[j].y}, r01, acc2[u]);
                        acc2[u] = __builtin_elementwise_fma(f32x2{w2[j].z, w2[j].w}, r23, acc2[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < QG; ++u) acc[u] = (w_total + wave_sum(acc2[u].x + acc2[u].y)) + bias_a;
                float sc = acc[0];
#pragma unroll
                for (int u = 1; u < QG; ++u) sc = lane == u ? acc[u] : sc;
                emit(sc, q0, r, masked);
            }
        } else
        // queries in groups of QG: their accumulators are independent chains for the VALU, and the wave reductions of a group
        // are issued back to back (beams of a clip / the T localizer queries of a clip share this row's registers)
        for (int q0 = 0; q0 < nq; q0 += QG) {
            float acc[QG];
            // additive form with several queries per row: the pass is bound by its two quarter-rate transcendentals per
            // element-query (exp2, rcp); the four full-rate operations around them are taken two elements at a time on the
            // packed-fp32 VALU (v_pk_add_f32 / v_pk_fma_f32), with even / odd partial sums
            using f32x2 = __attribute__((ext_vector_type(2))) float;
            f32x2 acc2[QG];
#pragma unroll
            for (int u = 0; u < QG; ++u) { acc[u] = 0.f; acc2[u] = f32x2{0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                f32x4 p = cok[j] ? cur[j] : f32x4{0, 0, 0, 0};
                f32x4 w4 = {0, 0, 0, 0};
                if (KIND == CVC_ATTN_ADDITIVE) {
                    w4 = ld4(w_s + colc[j]);
                    if (!cok[j]) w4 = f32x4{0, 0, 0, 0};
                    p *= EXP_C;
                }
#pragma unroll
                for (int u = 0; u < QG; ++u) {
                    const f32x4 q4 = ld4(q_s + (q0 + u) * A + colc[j]);
                    if (KIND == CVC_ATTN_ADDITIVE) {
                        const f32x4 x = p + q4;                    // C (p + q)
                        if constexpr (QG > 1) {
                            const f32x2 e01 = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                            const f32x2 e23 = {__builtin_amdgcn_exp2f(x.z), __builtin_amdgcn_exp2f(x.w)};
                            const f32x2 d01 = e01 + 1.0f, d23 = e23 + 1.0f;
                            const f32x2 r01 = {fast_rcp(d01.x), fast_rcp(d01.y)}, r23 = {fast_rcp(d23.x), fast_rcp(d23.y)};
                            const f32x2 t01 = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r01, f32x2{1.0f, 1.0f});
                            const f32x2 t23 = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r23, f32x2{1.0f, 1.0f});
                            acc2[u] = __builtin_elementwise_fma(f32x2{w4.x, w4.y}, t01, acc2[u]);
                            acc2[u] = __builtin_elementwise_fma(f32x2{w4.z, w4.w}, t23, acc2[u]);
                        } else {
                            acc[u] = fmaf(w4.x, fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(x.x)), 1.0f), acc[u]);
                            acc[u] = fmaf(w4.y, fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(x.y)), 1.0f), acc[u]);
                            acc[u] = fmaf(w4.z, fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(x.z)), 1.0f), acc[u]);
                            acc[u] = fmaf(w4.w, fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(x.w)), 1.0f), acc[u]);
                        }
                    } else {
                        acc[u] += p.x * q4.x + p.y * q4.y + p.z * q4.z + p.w * q4.w;
                    }
                }
            }
            if constexpr (KIND == CVC_ATTN_ADDITIVE && QG > 1) {
#pragma unroll
                for (int u = 0; u < QG; ++u) acc[u] = acc2[u].x + acc2[u].y;
            }
#pragma unroll
            for (int u = 0; u < QG; ++u) acc[u] = KIND == CVC_ATTN_ADDITIVE ? wave_sum(acc[u]) + bias_a : wave_sum(acc[u]) * a.inv_temp;
            float sc = acc[0];
#pragma unroll
            for (int u = 1; u < QG; ++u) sc = lane == u ? acc[u] : sc;
            emit(sc, q0, r, masked);
        }
    };
    for (int k = 0;;) {
        step(cur, nxt, r, k);
        r += 4; ++k;
        if (r >= row_end) break;
        step(nxt, cur, r, k);
        r += 4; ++k;
        if (r >= row_end) break;
    }
#undef LDF
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if constexpr (FACTORABLE) {
        if (factored) {
            // (the exact-width specialisation only here: the one-query and direct forms are not instruction-bound)
            if (a.A == NCH * 256) { if (S.stream & 1) rows(T_{}, T_{}, T_{}); else rows(F_{}, T_{}, T_{}); }
            else { if (S.stream & 1) rows(T_{}, T_{}, F_{}); else rows(F_{}, T_{}, F_{}); }
            return;
        }
    }
    if (S.stream & 1) rows(T_{}, F_{}, F_{}); else rows(F_{}, F_{}, F_{});
}

template <int KIND, int QG>
int launch_scores_q(const ScoreArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    const int nch = (a.A + 255) / 256;
#define CVC_SC(N) hipLaunchKernelGGL((attn_scores_kernel<KIND, N, QG>), grid, dim3(SCORE_WG), lds, st, a)
    if (nch <= 1) CVC_SC(1);
    else if (nch <= 2) CVC_SC(2);
    else if (nch <= 4) CVC_SC(4);
    else if (nch <= 8) CVC_SC(8);
    else if (nch <= 16) CVC_SC(16);
    else return CVC_E_TOOBIG;
#undef CVC_SC
    return cvc_launch_status();
}

// queries per group for nq queries: 1 / 4 / 5 are compiled; the list is padded to a multiple (5 beams -> 5, 20 localizer
// queries -> 5, 2 or 3 -> 4 with a little padding)
inline int score_group(int nq) {
    if (nq <= 1) return 1;
    const int p4 = (nq + 3) / 4 * 4, p5 = (nq + 4) / 5 * 5;
    return p5 <= p4 ? 5 : 4;
}

template <int KIND>
int launch_scores(const ScoreArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    switch (score_group(a.nq)) {
        case 1: return launch_scores_q<KIND, 1>(a, grid, lds, st);
        case 4: return launch_scores_q<KIND, 4>(a, grid, lds, st);
        default: return launch_scores_q<KIND, 5>(a, grid, lds, st);
    }
}


// Rows per workgroup for a launch with several queries per clip.  Such a launch is compute-bound and its workgroups run in ROUNDS of
// `slots` (workgroups resident on the chip at once: 4 per CU by registers, fewer when the staged queries fill the LDS), so what
// decides its duration is the number of rounds x the rows a workgroup walks -- a fixed 32 rows gave 1 216 workgroups for 1 024 slots
// at config 3 (two rounds, the second 19 % full) and 1 600 for 768 at config 5 (three rounds).  Picked: the r that minimises
// rounds(r) x (ceil(r / 4) + fixed), fixed = the per-workgroup staging of the queries expressed in rows per wave.
// CVC_SCORE_ROWS_RT=<r> overrides it (measurement runs).
inline int score_rows_per_wg(int nclip, int n0, int n1, size_t lds_bytes) {
    static const int forced = [] { const char* e = getenv("CVC_SCORE_ROWS_RT"); return e ? atoi(e) : 0; }();
    if (forced >= 4 && forced <= ROWS_PER_WG_MAX) return forced;
    int per_cu = (int)((160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1));
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    const long long slots = 256LL * per_cu;
    int best = ROWS_PER_WG;
    double best_cost = 1e30;
    for (int r = 8; r <= ROWS_PER_WG_MAX; ++r) {
        const long long wgs = (long long)nclip * ((n0 + r - 1) / r + (n1 > 0 ? (n1 + r - 1) / r : 0));
        const long long rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * ((r + 3) / 4 + 1.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = r; }
    }
    return best;
}

// Launch the score pass for `sets` (1 or 2), splitting long query lists so that the queries of
// a clip (plus alpha_net's weight) fit 64 KB of LDS.
inline int run_scores(int kind, const float* q, const float* w_a, const float* b_a, float inv_temp,
                      const cvc_attn_set* sets, int nsets, int nclip, int nq, int A, hipStream_t st,
                      int q_nparts = 1, const float* q_bias = nullptr, const void* q_slab = nullptr, long long q_ld = 0,
                      long long q_part_stride = 0) {
    int q_per_launch = (int)((64 * 1024) / ((size_t)A * 4)) - 1;
    if (q_per_launch < 1) return CVC_E_TOOBIG;
    if (q_per_launch > nq) q_per_launch = nq;
    // the padded query list of a launch must fit too: keep whole groups of 5 (or 4) per launch
    if (q_per_launch < nq && q_per_launch >= 5) q_per_launch = q_per_launch / 5 * 5;
    while (q_per_launch > 1 && (size_t)((q_per_launch + score_group(q_per_launch) - 1) / score_group(q_per_launch) *
                                        score_group(q_per_launch) + 1) * A * 4 > 64 * 1024) --q_per_launch;
    ScoreArgs sa;
    sa.set[0] = sets[0];
    sa.set[1] = nsets > 1 ? sets[1] : sets[0];
    sa.q = q; sa.w_a = w_a; sa.b_a = b_a; sa.inv_temp = inv_temp; sa.A = A; sa.nq_total = nq;
    sa.q_nparts = q_nparts < 1 ? 1 : q_nparts; sa.q_bias = q_bias;
    sa.q_ld = q_ld > 0 ? q_ld : A;
    sa.q_part_stride = q_part_stride > 0 ? q_part_stride : (long long)nclip * nq * A;
#ifdef CVC_EXPERIMENTAL
    sa.q_from_slab = q_slab != nullptr;
    sa.q_slab = q_slab != nullptr ? *static_cast<const GskSegs*>(q_slab) : GskSegs{};
#else
    if (q_slab != nullptr) return CVC_E_BADARG;
#endif
    for (int q0 = 0; q0 < nq; q0 += q_per_launch) {
        sa.q0 = q0;
        sa.nq = nq - q0 < q_per_launch ? nq - q0 : q_per_launch;
        const int qg = score_group(sa.nq);
        const size_t lds1 = (size_t)((sa.nq + qg - 1) / qg * qg + 1) * A * sizeof(float);
        // one query per clip: HBM-bound, the fixed 32 rows (measured insensitive); several: balanced rounds
        sa.rows_per_wg = sa.nq > 1 ? score_rows_per_wg(nclip, sets[0].n, nsets > 1 ? sets[1].n : 0, lds1) : ROWS_PER_WG;
        sa.chunks0 = (sets[0].n + sa.rows_per_wg - 1) / sa.rows_per_wg;
        const int chunks1 = nsets > 1 ? (sets[1].n + sa.rows_per_wg - 1) / sa.rows_per_wg : 0;
        dim3 g1(sa.chunks0 + chunks1, nclip);
        int rc = kind == CVC_ATTN_ADDITIVE ? launch_scores<CVC_ATTN_ADDITIVE>(sa, g1, lds1, st)
                                           : launch_scores<CVC_ATTN_DOT>(sa, g1, lds1, st);
        if (rc != 0) return rc;
    }
    return 0;
}

}  // namespace
