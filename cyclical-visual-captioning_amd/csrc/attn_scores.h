// Score pass shared by the attention forward, the attention backward (d_attn = d_ctx . ctx_n is
// the same streaming dot product) and the grounder (captioner.py:154-158).  See attn_fwd.hip.
#pragma once
#include "cvc_common.h"
#include <type_traits>

namespace {

#ifndef CVC_SCORE_ROWS
#define CVC_SCORE_ROWS 32
#endif
constexpr int ROWS_PER_WG = CVC_SCORE_ROWS;   // score kernel: rows per workgroup (4 waves, interleaved)
constexpr int SCORE_WG = 256;

struct ScoreArgs {
    cvc_attn_set set[2];
    int chunks0;                   // blockIdx.x < chunks0 -> set 0 else set 1
    const float* q;                // [q_nparts][nclip*nq, A]: partial sums of the query GEMM (split-K), summed on load
    const float* q_bias;           // [A] added once (h2attn.bias) or null
    int q_nparts;
    long long q_part_stride;
    const float* w_a;              // [A] (additive)
    const float* b_a;              // device scalar (alpha_net.bias) or null
    float inv_temp;
    int nq, A;                     // nq = queries handled by this launch
    int nq_total, q0;              // row = clip * nq_total + q0 + qi
};

template <int KIND, int NCH>
__global__ __launch_bounds__(SCORE_WG) void attn_scores_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q_s = smem;                         // [nq][A]
    float* w_s = smem + (size_t)a.nq * a.A;    // [A]
    const int clip = blockIdx.y;
    const int s = (int)blockIdx.x < a.chunks0 ? 0 : 1;
    const int chunk = s == 0 ? blockIdx.x : blockIdx.x - a.chunks0;
    const cvc_attn_set& S = a.set[s];
    const int A = a.A, nq = a.nq, n = S.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid * 4; i < nq * A; i += SCORE_WG * 4) {
        const float* src = a.q + ((size_t)clip * a.nq_total + a.q0) * A + i;
        f32x4 v = ld4(src);
        for (int p = 1; p < a.q_nparts; ++p) v += ld4(src + (size_t)p * a.q_part_stride);
        if (a.q_bias != nullptr) v += ld4(a.q_bias + (i % A));
        st4(q_s + i, v);
    }
    if (KIND == CVC_ATTN_ADDITIVE)
        for (int i = tid * 4; i < A; i += SCORE_WG * 4) st4(w_s + i, ld4(a.w_a + i));
    __syncthreads();

    const int row0 = chunk * ROWS_PER_WG;
    const int row_end = min(n, row0 + ROWS_PER_WG);
    const float* P = S.proj + (size_t)clip * n * A;

    // a set marked `stream` is read with the non-temporal policy so that it does not displace what the decode loop
    // re-reads every step from the 256 MB Infinity Cache (the caller budgets that: cvc/decode.py)
    auto rows = [&](auto stream_tag) __attribute__((always_inline)) {
        constexpr bool STREAM = decltype(stream_tag)::value;
#define LDF(ptr) (STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr)) : ld4(ptr))
    f32x4 cur[NCH], nxt[NCH];
    int r = row0 + wave;
    if (r < row_end) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int col = (j * 64 + lane) * 4;
            cur[j] = col < A ? LDF(P + (size_t)r * A + col) : f32x4{0, 0, 0, 0};
        }
    }
    for (; r < row_end; r += 4) {
        const int rn = r + 4;
        if (rn < row_end) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                int col = (j * 64 + lane) * 4;
                nxt[j] = col < A ? LDF(P + (size_t)rn * A + col) : f32x4{0, 0, 0, 0};
            }
        }
        const bool masked = S.mask != nullptr && S.mask[(size_t)clip * n + r] != 0;
        for (int qi = 0; qi < nq; ++qi) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                int col = (j * 64 + lane) * 4;
                if (col < A) {
                    f32x4 q4 = ld4(q_s + qi * A + col);
                    f32x4 p = cur[j];
                    if (KIND == CVC_ATTN_ADDITIVE) {
                        f32x4 w4 = ld4(w_s + col);
                        acc += w4.x * fast_tanh(p.x + q4.x);
                        acc += w4.y * fast_tanh(p.y + q4.y);
                        acc += w4.z * fast_tanh(p.z + q4.z);
                        acc += w4.w * fast_tanh(p.w + q4.w);
                    } else {
                        acc += p.x * q4.x + p.y * q4.y + p.z * q4.z + p.w * q4.w;
                    }
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) {
                float sc = KIND == CVC_ATTN_ADDITIVE ? acc + (a.b_a != nullptr ? a.b_a[0] : 0.f) : acc * a.inv_temp;
                if (masked) sc = CVC_MIN_VALUE;
                const size_t o = ((size_t)clip * a.nq_total + a.q0 + qi) * n + r;
                S.scores[o] = sc;
                if (S.frame_masked != nullptr)
                    S.frame_masked[o] = S.frame_mask[o] != 0 ? CVC_MIN_VALUE : sc;
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) cur[j] = nxt[j];
    }
#undef LDF
    };
    if (S.stream & 1) rows(std::true_type{}); else rows(std::false_type{});
}

template <int KIND>
int launch_scores(const ScoreArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    const int nch = (a.A + 255) / 256;
#define CVC_SC(N) hipLaunchKernelGGL((attn_scores_kernel<KIND, N>), grid, dim3(SCORE_WG), lds, st, a)
    if (nch <= 1) CVC_SC(1);
    else if (nch <= 2) CVC_SC(2);
    else if (nch <= 4) CVC_SC(4);
    else if (nch <= 8) CVC_SC(8);
    else if (nch <= 16) CVC_SC(16);
    else return CVC_E_TOOBIG;
#undef CVC_SC
    return cvc_launch_status();
}


// Launch the score pass for `sets` (1 or 2), splitting long query lists so that the queries of
// a clip (plus alpha_net's weight) fit 64 KB of LDS.
inline int run_scores(int kind, const float* q, const float* w_a, const float* b_a, float inv_temp,
                      const cvc_attn_set* sets, int nsets, int nclip, int nq, int A, hipStream_t st,
                      int q_nparts = 1, const float* q_bias = nullptr) {
    int q_per_launch = (int)((64 * 1024) / ((size_t)A * 4)) - 1;
    if (q_per_launch < 1) return CVC_E_TOOBIG;
    if (q_per_launch > nq) q_per_launch = nq;
    ScoreArgs sa;
    sa.set[0] = sets[0];
    sa.set[1] = nsets > 1 ? sets[1] : sets[0];
    sa.chunks0 = (sets[0].n + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const int chunks1 = nsets > 1 ? (sets[1].n + ROWS_PER_WG - 1) / ROWS_PER_WG : 0;
    sa.q = q; sa.w_a = w_a; sa.b_a = b_a; sa.inv_temp = inv_temp; sa.A = A; sa.nq_total = nq;
    sa.q_nparts = q_nparts < 1 ? 1 : q_nparts; sa.q_bias = q_bias; sa.q_part_stride = (long long)nclip * nq * A;
    dim3 g1(sa.chunks0 + chunks1, nclip);
    for (int q0 = 0; q0 < nq; q0 += q_per_launch) {
        sa.q0 = q0;
        sa.nq = nq - q0 < q_per_launch ? nq - q0 : q_per_launch;
        const size_t lds1 = (size_t)(sa.nq + 1) * A * sizeof(float);
        int rc = kind == CVC_ATTN_ADDITIVE ? launch_scores<CVC_ATTN_ADDITIVE>(sa, g1, lds1, st)
                                           : launch_scores<CVC_ATTN_DOT>(sa, g1, lds1, st);
        if (rc != 0) return rc;
    }
    return 0;
}

}  // namespace
