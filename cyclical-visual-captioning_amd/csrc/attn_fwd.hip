// Region / frame attention forward for the caption decoder (model/modules.py:24-159 in the
// reference) as two streaming kernels, both HBM-bound:
//
//   1. attn_scores_kernel : one pass over proj_context [nclip, n, A].  A wave owns a feature
//      row at a time (1 KiB-per-instruction coalesced dwordx4 loads, next row prefetched),
//      the clip's query vectors q (and alpha_net's weight) sit in LDS and are shared by all
//      rows / all beams; add -> tanh -> dot(w) happens in registers, one wave reduction per
//      (row, query).  Writes masked pre-softmax scores (+ the frame-masked copy).
//   2. attn_wsum_kernel   : softmax over n (recomputed per workgroup from the scores, a few
//      KB) and one pass over context [nclip, n, R]: a workgroup owns (query row, 256-column
//      block), its 4 waves split n, each lane carries 4 columns; partial sums meet in LDS.
//      Both feature sets of a decoder step (regions + frames) are handled in one launch and
//      their contexts are summed in registers (decoder_core.py:59).
//
// Every feature byte is read exactly once per clip-step: algorithmic bytes per clip-step
// = 4 (N+F)(A+R) + N  (SURVEY.md section 8(d)).
#include "attn_scores.h"
#include <math.h>

namespace {

constexpr int WG = 256;

struct WsumArgs {
    cvc_attn_set set[2];
    int nsets, nq, R;
    float* ctx_sum;               // [rows, R] or null
    int ctx_quad;                 // 1: ctx_sum is written in the GEMM's quad layout [R/4][64][4] (rows <= 64)
                                  // 2: as bf16 split-term fragments of the tile GEMM (gemm_tile.hip), row-block stride below
    long long frag_stride;
    float* ctx_sum_rm;            // one-query kernel: the summed context once more, row-major [rows, R] (training loops), or null
    float raw_scale;              // several-queries kernel, != 0: the weights are `scores` * raw_scale as they are (no softmax, `attn`
                                  // not written): out[row, :] = scale * sum_n w[row, n] X[clip, n, :] -- the dot-product score backward
};

// 4 consecutive columns of row m as the three bf16 terms (hi, mid, lo: exact truncation split, gemm_split.h) inside the
// 1 KiB fragments [k half][row & 31][8 k] of the tile GEMM's activation layout
__device__ __forceinline__ void store_ctx_frag(uint16_t* xb, long long mblk_stride, int m, int k, const f32x4 v) {
    using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;
    u16x4 p[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u0 = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
        p[0][e] = (uint16_t)(u0 >> 16); p[1][e] = (uint16_t)(u1 >> 16); p[2][e] = (uint16_t)(__float_as_uint(r2) >> 16);
    }
    uint16_t* base = xb + (size_t)(m >> 5) * mblk_stride + (size_t)(k >> 4) * 1536 + (((k >> 3) & 1) * 32 + (m & 31)) * 8 + (k & 7);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u16x4*>(base + pl * 512) = p[pl];
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();                         // red[] may still be read from a previous call
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < WG / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

// HOIST (n <= 512 in every set): the softmax rows of both feature sets are made together before the first context byte is
// requested -- their score loads, the two block-wide maxima and the two block-wide sums each share ONE exchange (2 barriers instead
// of 8 ahead of the streaming loops); per set the same partial sums in the same order as the per-set form: bit-identical weights.
template <bool HOIST>
__global__ __launch_bounds__(WG) void attn_wsum_kernel(WsumArgs a, int n_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* part = reinterpret_cast<f32x4*>(smem);      // [4 waves][64 lanes]
    float* red = smem + 4 * 64 * 4;                    // [16]
    float* a_s = red + 16;                             // [n_max]  (HOIST: [sets][n_max])
    const int row = blockIdx.y, cb = blockIdx.x;
    const int clip = row / a.nq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.R;
    const int col = cb * 256 + lane * 4;
    const bool col_ok = col < R;
    f32x4 total = {0, 0, 0, 0};

    // HOIST: the first context rows of set 0 (what the streaming loop's first round takes) are requested BEFORE the softmax rows
    // are made -- they do not depend on them, and their HBM latency passes under the score loads and the two exchanges
    f32x4 pre[4] = {};
    bool pre_ok = false;
    if constexpr (HOIST) {
        const cvc_attn_set& S0 = a.set[0];
        pre_ok = col_ok && S0.n >= 16 && (S0.ctx_out != nullptr || a.ctx_sum != nullptr);
        if (pre_ok) {
            const float* C0 = S0.ctx + (size_t)clip * S0.n * R + col + (size_t)wave * R;
            if (S0.stream & 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) pre[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(C0 + (size_t)(4 * k) * R));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) pre[k] = ld4(C0 + (size_t)(4 * k) * R);
            }
        }
    }

    if constexpr (HOIST) {
        float v[2][2], m[2], sum[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = s < a.nsets ? a.set[s].n : 0;
            const float* sc = a.set[s].scores + (size_t)row * n;
            v[s][0] = tid < n ? sc[tid] : -INFINITY;
            v[s][1] = tid + WG < n ? sc[tid + WG] : -INFINITY;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            m[s] = wave_max(fmaxf(v[s][0], v[s][1]));
            if (lane == 0) red[s * 4 + wave] = m[s];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = s < a.nsets ? a.set[s].n : 0;
            m[s] = fmaxf(fmaxf(fmaxf(red[s * 4], red[s * 4 + 1]), red[s * 4 + 2]), red[s * 4 + 3]);
            float part_sum = 0.f;
            if (tid < n) { v[s][0] = expf(v[s][0] - m[s]); part_sum += v[s][0]; }
            if (tid + WG < n) { v[s][1] = expf(v[s][1] - m[s]); part_sum += v[s][1]; }
            part_sum = wave_sum(part_sum);
            if (lane == 0) red[8 + s * 4 + wave] = part_sum;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = s < a.nsets ? a.set[s].n : 0;
            sum[s] = ((red[8 + s * 4] + red[8 + s * 4 + 1]) + red[8 + s * 4 + 2]) + red[8 + s * 4 + 3];
            float* as = a_s + (size_t)s * n_max;
            if (tid < n) { const float p = v[s][0] / sum[s]; as[tid] = p; if (cb == 0) a.set[s].attn[(size_t)row * n + tid] = p; }
            if (tid + WG < n) { const float p = v[s][1] / sum[s]; as[tid + WG] = p; if (cb == 0) a.set[s].attn[(size_t)row * n + tid + WG] = p; }
        }
        __syncthreads();
    }

    for (int s = 0; s < a.nsets; ++s) {
        const cvc_attn_set& S = a.set[s];
        const int n = S.n;
        if constexpr (HOIST) a_s = red + 16 + (size_t)s * n_max;
        else {
        const float* sc = S.scores + (size_t)row * n;
        // softmax over n (torch.softmax: exp(x - max) / sum)
        float m = -INFINITY;
        for (int i = tid; i < n; i += WG) m = fmaxf(m, sc[i]);
        m = block_reduce(m, red, true);
        float sum = 0.f;
        for (int i = tid; i < n; i += WG) {
            float e = expf(sc[i] - m);
            a_s[i] = e;
            sum += e;
        }
        sum = block_reduce(sum, red, false);
        for (int i = tid; i < n; i += WG) {
            float p = a_s[i] / sum;
            a_s[i] = p;
            if (cb == 0) S.attn[(size_t)row * n + i] = p;
        }
        __syncthreads();
        }
        if (S.ctx_out == nullptr && a.ctx_sum == nullptr) continue;

        const float* C = S.ctx + (size_t)clip * n * R + col;
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        // a set marked `stream` (bit 1) is read non-temporally, see attn_scores.h
        auto accumulate = [&](auto stream_tag) __attribute__((always_inline)) {
            constexpr bool STREAM = decltype(stream_tag)::value;
#define LDF(ptr) (STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr)) : ld4(ptr))
            int i = wave;
            if (HOIST && s == 0 && pre_ok) {               // (n >= 16: the round the loop below would have started with)
                acc0 += a_s[i] * pre[0];
                acc1 += a_s[i + 4] * pre[1];
                acc2 += a_s[i + 8] * pre[2];
                acc3 += a_s[i + 12] * pre[3];
                i += 16;
            }
            for (; i + 12 < n; i += 16) {
                f32x4 c0 = LDF(C + (size_t)i * R), c1 = LDF(C + (size_t)(i + 4) * R);
                f32x4 c2 = LDF(C + (size_t)(i + 8) * R), c3 = LDF(C + (size_t)(i + 12) * R);
                acc0 += a_s[i] * c0;
                acc1 += a_s[i + 4] * c1;
                acc2 += a_s[i + 8] * c2;
                acc3 += a_s[i + 12] * c3;
            }
            for (; i < n; i += 4) acc0 += a_s[i] * LDF(C + (size_t)i * R);
#undef LDF
        };
        if (col_ok) {
            if (S.stream & 2) accumulate(std::true_type{}); else accumulate(std::false_type{});
        }
        part[wave * 64 + lane] = (acc0 + acc1) + (acc2 + acc3);
        __syncthreads();
        if (wave == 0 && col_ok) {
            f32x4 v = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
            if (S.ctx_out != nullptr) st4(S.ctx_out + (size_t)row * R + col, v);
            total += v;
        }
        __syncthreads();
    }
    if (a.ctx_sum != nullptr && wave == 0 && col_ok) {
        if (a.ctx_quad == 2) store_ctx_frag(reinterpret_cast<uint16_t*>(a.ctx_sum), a.frag_stride, row, col, total);
        else if (a.ctx_quad) st4(a.ctx_sum + ((size_t)(col >> 2) * 64 + row) * 4, total);
        else st4(a.ctx_sum + (size_t)row * R + col, total);
    }
    if (a.ctx_sum_rm != nullptr && wave == 0 && col_ok) st4(a.ctx_sum_rm + (size_t)row * R + col, total);
}

// Several queries per clip (beams of a clip, the T localizer queries of a clip): one workgroup = (clip, 256-column block,
// group of up to QB queries).  The clip's context rows are streamed ONCE for the whole group and accumulated into QB register
// accumulators; the one-query kernel above would re-read them per query (5 x the bytes at beam 5).
// NW waves per workgroup split the clip's rows: 4, or 8 (beam groups of up to 5 queries over long feature sets: a launch is only
// (R / 256) x clips workgroups -- two per compute unit -- so four waves each leave a CU with 8 waves and 64 KB of loads in flight)
// HOIST: the softmax rows of BOTH feature sets are made before the first context byte is requested, every wave working on its
// (set, query) pairs together (their score loads, reductions and exponentials interleave: one latency chain instead of one per
// round and set), values in registers (n <= 512); a_s holds [sets][QB][n_max].  Same arithmetic in the same order as the per-set
// form (lane-strided max / sum, wave reductions): bit-identical weights.
template <int QB, int NW = 4, bool HOIST = false>
__global__ __launch_bounds__(NW * 64) void attn_wsum_mq_kernel(WsumArgs a, int n_max) {
    constexpr int WG = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* part = reinterpret_cast<f32x4*>(smem);      // [NW waves][QB][64 lanes]
    float* red = smem + NW * QB * 64 * 4;              // [16]
    float* a_s = red + 16;                             // [QB][n_max]  (HOIST: [sets][QB][n_max])
    const int clip = blockIdx.y, cb = blockIdx.x;
    const int qbase = blockIdx.z * QB;
    const int nqb = min(QB, a.nq - qbase);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.R;
    const int col = cb * 256 + lane * 4;
    const bool col_ok = col < R;
    f32x4 total[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) total[u] = f32x4{0, 0, 0, 0};

    // (requesting set 0's first round of context rows before the softmax rows, as attn_wsum_kernel does, was measured SLOWER here:
    // 56 -> 62 us at config 3, 136 -> 147 us at config 5 -- 30 to 40 more live registers per lane)
    if constexpr (HOIST) {
        constexpr int PV = 8;                              // scores per lane and pair (n <= 512)
        constexpr int PP = (2 * QB + NW - 1) / NW;         // pairs per wave
        for (int s = 0; s < a.nsets; ++s)
            for (int u = nqb; u < QB; ++u)
                for (int i = tid; i < a.set[s].n; i += WG) a_s[((size_t)s * QB + u) * n_max + i] = 0.f;
        const int npairs = a.nsets * nqb;
        float v[PP][PV];
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int p = wave + j * NW;
            if (p < npairs) {
                const int s = p / nqb, u = p - s * nqb, n = a.set[s].n;
                const float* sc = a.set[s].scores + (size_t)(clip * a.nq + qbase + u) * n;
#pragma unroll
                for (int k = 0; k < PV; ++k) v[j][k] = lane + 64 * k < n ? sc[lane + 64 * k] : -INFINITY;
            }
        }
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int p = wave + j * NW;
            if (p < npairs) {
                const int s = p / nqb, u = p - s * nqb, n = a.set[s].n;
                const int row = clip * a.nq + qbase + u;
                float* as = a_s + ((size_t)s * QB + u) * n_max;
                if (a.raw_scale != 0.f) {
#pragma unroll
                    for (int k = 0; k < PV; ++k) if (lane + 64 * k < n) as[lane + 64 * k] = v[j][k] * a.raw_scale;
                    continue;
                }
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < PV; ++k) if (lane + 64 * k < n) m = fmaxf(m, v[j][k]);
                m = wave_max(m);
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < PV; ++k) if (lane + 64 * k < n) { v[j][k] = expf(v[j][k] - m); sum += v[j][k]; }
                sum = wave_sum(sum);
#pragma unroll
                for (int k = 0; k < PV; ++k) if (lane + 64 * k < n) {
                    const float pr = v[j][k] / sum;
                    as[lane + 64 * k] = pr;
                    if (cb == 0) a.set[s].attn[(size_t)row * n + lane + 64 * k] = pr;
                }
            }
        }
        __syncthreads();
    }

    for (int s = 0; s < a.nsets; ++s) {
        const cvc_attn_set& S = a.set[s];
        const int n = S.n;
        if constexpr (HOIST) a_s = red + 16 + (size_t)s * QB * n_max;
        else {
        for (int u = nqb; u < QB; ++u)                  // unused slots of the group weigh nothing: the FMA loop has no branches
            for (int i = tid; i < n; i += WG) a_s[(size_t)u * n_max + i] = 0.f;
        // softmax over n per query (torch.softmax: exp(x - max) / sum): one WAVE per query, reductions inside the wave -- the
        // block-wide form cost 2 reductions x 2 barriers per query and feature set before the first context byte was requested
        for (int u = wave; u < nqb; u += WG / 64) {
            const int row = clip * a.nq + qbase + u;
            const float* sc = S.scores + (size_t)row * n;
            float* as = a_s + (size_t)u * n_max;
            if (a.raw_scale != 0.f) {
                for (int i = lane; i < n; i += 64) as[i] = sc[i] * a.raw_scale;
                continue;
            }
            float m = -INFINITY;
            for (int i = lane; i < n; i += 64) m = fmaxf(m, sc[i]);
            m = wave_max(m);
            float sum = 0.f;
            for (int i = lane; i < n; i += 64) {
                float e = expf(sc[i] - m);
                as[i] = e;
                sum += e;
            }
            sum = wave_sum(sum);
            for (int i = lane; i < n; i += 64) {
                float p = as[i] / sum;
                as[i] = p;
                if (cb == 0) S.attn[(size_t)row * n + i] = p;
            }
        }
        __syncthreads();
        }
        if (S.ctx_out == nullptr && a.ctx_sum == nullptr) continue;

        const float* C = S.ctx + (size_t)clip * n * R + col;
        f32x4 acc[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) acc[u] = f32x4{0, 0, 0, 0};
        auto accumulate = [&](auto stream_tag) __attribute__((always_inline)) {
            constexpr bool STREAM = decltype(stream_tag)::value;
#define LDF(ptr) (STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr)) : ld4(ptr))
            int i = wave;
            for (; i + 7 * NW < n; i += 8 * NW) {          // 8 rows (8 KB per wave) in flight
                f32x4 c[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = LDF(C + (size_t)(i + NW * k) * R);
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const float* as = a_s + (size_t)u * n_max;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[u] += as[i + NW * k] * c[k];
                }
            }
            for (; i + 3 * NW < n; i += 4 * NW) {
                const f32x4 c0 = LDF(C + (size_t)i * R), c1 = LDF(C + (size_t)(i + NW) * R);
                const f32x4 c2 = LDF(C + (size_t)(i + 2 * NW) * R), c3 = LDF(C + (size_t)(i + 3 * NW) * R);
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const float* as = a_s + (size_t)u * n_max;
                    acc[u] += as[i] * c0;
                    acc[u] += as[i + NW] * c1;
                    acc[u] += as[i + 2 * NW] * c2;
                    acc[u] += as[i + 3 * NW] * c3;
                }
            }
            for (; i < n; i += NW) {
                const f32x4 c0 = LDF(C + (size_t)i * R);
#pragma unroll
                for (int u = 0; u < QB; ++u) acc[u] += a_s[(size_t)u * n_max + i] * c0;
            }
#undef LDF
        };
        if (col_ok) {
            if (S.stream & 2) accumulate(std::true_type{}); else accumulate(std::false_type{});
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) part[(wave * QB + u) * 64 + lane] = acc[u];
        __syncthreads();
        if (wave == 0 && col_ok) {
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                if (u < nqb) {
                    f32x4 v = (part[(0 * QB + u) * 64 + lane] + part[(1 * QB + u) * 64 + lane]) +
                              (part[(2 * QB + u) * 64 + lane] + part[(3 * QB + u) * 64 + lane]);
                    if (NW == 8) v += (part[(4 * QB + u) * 64 + lane] + part[(5 * QB + u) * 64 + lane]) +
                                      (part[(6 * QB + u) * 64 + lane] + part[(7 * QB + u) * 64 + lane]);
                    if (S.ctx_out != nullptr) st4(S.ctx_out + (size_t)(clip * a.nq + qbase + u) * R + col, v);
                    total[u] += v;
                }
            }
        }
        __syncthreads();
    }
    if (a.ctx_sum != nullptr && wave == 0 && col_ok) {
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            if (u < nqb) {
                const int row = clip * a.nq + qbase + u;
                if (a.ctx_quad == 2) store_ctx_frag(reinterpret_cast<uint16_t*>(a.ctx_sum), a.frag_stride, row, col, total[u]);
                else if (a.ctx_quad) st4(a.ctx_sum + ((size_t)(col >> 2) * 64 + row) * 4, total[u]);
                else st4(a.ctx_sum + (size_t)row * R + col, total[u]);
            }
        }
    }
}

}  // namespace

static int check_sets(const cvc_attn_set* sets, int nsets, int nclip, int nq, int A, int R, int* n_max) {
    if (sets == nullptr || nsets < 1 || nsets > 2 || nclip < 1 || nq < 1 || (A & 3) || (R & 3)) return CVC_E_BADARG;
    *n_max = 0;
    for (int s = 0; s < nsets; ++s) {
        if (sets[s].n < 1 || !sets[s].proj || !sets[s].ctx || !sets[s].scores || !sets[s].attn) return CVC_E_BADARG;
        if ((sets[s].frame_masked != nullptr) != (sets[s].frame_mask != nullptr)) return CVC_E_BADARG;
        *n_max = sets[s].n > *n_max ? sets[s].n : *n_max;
    }
    return 0;
}

extern "C" int cvc_attn_scores(int kind, const float* q, const float* w_a, const float* b_a, float inv_temp,
                               const cvc_attn_set* sets, int nsets, int nclip, int nq, int A, cvc_stream_t stream) {
    int n_max;
    int rc = check_sets(sets, nsets, nclip, nq, A, 0, &n_max);
    if (rc) return rc;
    if (q == nullptr || (kind == CVC_ATTN_ADDITIVE && w_a == nullptr)) return CVC_E_BADARG;
    if (kind != CVC_ATTN_ADDITIVE && kind != CVC_ATTN_DOT) return CVC_E_BADARG;
    return run_scores(kind, q, w_a, b_a, inv_temp, sets, nsets, nclip, nq, A, (hipStream_t)stream);
}

extern "C" int cvc_attn_scores_qparts(int kind, const float* q_parts, int q_nparts, const float* q_bias, const float* w_a,
                                      const float* b_a, float inv_temp, const cvc_attn_set* sets, int nsets, int nclip,
                                      int nq, int A, cvc_stream_t stream) {
    int n_max;
    int rc = check_sets(sets, nsets, nclip, nq, A, 0, &n_max);
    if (rc) return rc;
    if (q_parts == nullptr || q_nparts < 1 || (kind == CVC_ATTN_ADDITIVE && w_a == nullptr)) return CVC_E_BADARG;
    if (kind != CVC_ATTN_ADDITIVE && kind != CVC_ATTN_DOT) return CVC_E_BADARG;
    return run_scores(kind, q_parts, w_a, b_a, inv_temp, sets, nsets, nclip, nq, A, (hipStream_t)stream, q_nparts, q_bias);
}

#ifdef CVC_EXPERIMENTAL
extern "C" int cvc_attn_scores_qslab(int kind, const cvc_gsk_segs* q, const float* q_bias, const float* w_a, const float* b_a,
                                     float inv_temp, const cvc_attn_set* sets, int nsets, int nclip, int nq, int A,
                                     cvc_stream_t stream) {
    int n_max;
    int rc = check_sets(sets, nsets, nclip, nq, A, 0, &n_max);
    if (rc) return rc;
    if (q == nullptr || q->slab == nullptr || q->nchunk < 1 || q->U < 1 || q->maxseg < 1 || q->unit0 < 0 || (A & 31) || nclip * nq > 64 ||
        (kind == CVC_ATTN_ADDITIVE && w_a == nullptr))
        return CVC_E_BADARG;
    if (kind != CVC_ATTN_ADDITIVE && kind != CVC_ATTN_DOT) return CVC_E_BADARG;
    return run_scores(kind, q->slab, w_a, b_a, inv_temp, sets, nsets, nclip, nq, A, (hipStream_t)stream, 1, q_bias, q);
}
#endif

static int wsum_impl(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, float* ctx_sum, int ctx_quad,
                     cvc_stream_t stream, long long frag_stride = 0, float* ctx_sum_rm = nullptr, float raw_scale = 0.f);

extern "C" int cvc_attn_wsum(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, float* ctx_sum,
                             cvc_stream_t stream) {
    return wsum_impl(sets, nsets, nclip, nq, R, ctx_sum, 0, stream);
}

extern "C" int cvc_attn_wsum_quad(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, float* ctx_sum_q,
                                  cvc_stream_t stream) {
    if (ctx_sum_q == nullptr || nclip * nq > 64) return CVC_E_BADARG;
    return wsum_impl(sets, nsets, nclip, nq, R, ctx_sum_q, 1, stream);
}

extern "C" int cvc_attn_wsum_frag(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, void* ctx_frag,
                                  long long frag_mblk_stride, cvc_stream_t stream) {
    if (ctx_frag == nullptr || (R & 15) || (frag_mblk_stride & 3)) return CVC_E_BADARG;
    return wsum_impl(sets, nsets, nclip, nq, R, (float*)ctx_frag, 2, stream, frag_mblk_stride);
}

// the summed context in the quad layout AND row-major (the training loops: next gate GEMM's operand + the dW product's rows)
extern "C" int cvc_attn_wsum_quad_rm(const cvc_attn_set* sets, int nsets, int nclip, int R, float* ctx_sum_q, float* ctx_sum_rm,
                                     cvc_stream_t stream) {
    if (ctx_sum_q == nullptr || ctx_sum_rm == nullptr || nclip > 64) return CVC_E_BADARG;
    return wsum_impl(sets, nsets, nclip, 1, R, ctx_sum_q, 1, stream, 0, ctx_sum_rm);
}

static int wsum_impl(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, float* ctx_sum, int ctx_quad,
                     cvc_stream_t stream, long long frag_stride, float* ctx_sum_rm, float raw_scale) {
    int n_max;
    int rc = check_sets(sets, nsets, nclip, nq, 0, R, &n_max);
    if (rc) return rc;
    WsumArgs wa;
    wa.set[0] = sets[0];
    wa.set[1] = nsets > 1 ? sets[1] : sets[0];
    wa.nsets = nsets; wa.nq = nq; wa.R = R; wa.ctx_sum = ctx_sum; wa.ctx_quad = ctx_quad; wa.frag_stride = frag_stride;
    wa.ctx_sum_rm = ctx_sum_rm;
    wa.raw_scale = raw_scale;
    if (ctx_sum_rm != nullptr && nq != 1) return CVC_E_BADARG;
    if (raw_scale != 0.f && nq < 2) return CVC_E_BADARG;
    if (nq > 1) {
        // queries of a clip in groups of QB (the largest group whose softmax rows + partials fit 64 KB of LDS): the clip's
        // context rows are read once per group; if not even 2 fit, the one-query kernel below takes every row on its own
        // (10 when it halves the passes over the clip's rows: the T = 20 localizer queries in 2 groups instead of 3)
        int QB = nq <= 2 ? 2 : (nq <= 4 ? 4 : (nq <= 5 ? 5 : ((nq + 9) / 10 < (nq + 7) / 8 ? 10 : 8)));
        auto lds_of = [&](int qb, int nw = 4) { return ((size_t)nw * qb * 64 * 4 + 16 + (size_t)qb * n_max) * sizeof(float); };
        while (QB > 1 && lds_of(QB) > 64 * 1024) QB = QB == 10 ? 8 : (QB == 8 ? 5 : (QB == 5 ? 4 : (QB == 4 ? 2 : 1)));
        if (QB > 1) {
            dim3 g((R + 255) / 256, nclip, (nq + QB - 1) / QB);
            // beam decode's group of 5 (CVC_WSUM_MQ_WAVES=4|8 and CVC_WSUM_MQ_HOIST=0|1: A/B): both sets' softmax rows up front where
            // they fit (config 3: 61.6 -> 57.7 us, config 5: 165.6 -> 136.3 us = 0.76 of HBM); with it, 8 waves per workgroup are
            // slightly ahead on a grid of two workgroups per CU (config 3: 56.2 us) and slightly behind on four (config 5: 138.2)
            static const int waves_env = [] { const char* e = getenv("CVC_WSUM_MQ_WAVES"); return e ? atoi(e) : 0; }();
            static const int hoist_env = [] { const char* e = getenv("CVC_WSUM_MQ_HOIST"); return e ? atoi(e) : -1; }();
            const long long wgs = (long long)g.x * g.y * g.z;
            if (QB == 5 && n_max <= 512) {
                const bool eight = waves_env == 8 || (waves_env != 4 && wgs <= 2 * 256);
                auto lds_h = [&](int nw) { return ((size_t)nw * 5 * 64 * 4 + 16 + (size_t)nsets * 5 * n_max) * sizeof(float); };
                const bool hoist = hoist_env != 0 && lds_h(eight ? 8 : 4) <= 64 * 1024;
                const hipStream_t st = (hipStream_t)stream;
                if (hoist && eight) { hipLaunchKernelGGL((attn_wsum_mq_kernel<5, 8, true>), g, dim3(512), lds_h(8), st, wa, n_max); return cvc_launch_status(); }
                if (hoist) { hipLaunchKernelGGL((attn_wsum_mq_kernel<5, 4, true>), g, dim3(256), lds_h(4), st, wa, n_max); return cvc_launch_status(); }
                if (eight && lds_of(5, 8) <= 64 * 1024) {
                    hipLaunchKernelGGL((attn_wsum_mq_kernel<5, 8>), g, dim3(512), lds_of(5, 8), st, wa, n_max);
                    return cvc_launch_status();
                }
            }
            const size_t lds = lds_of(QB);
            switch (QB) {
                case 2: hipLaunchKernelGGL(attn_wsum_mq_kernel<2>, g, dim3(WG), lds, (hipStream_t)stream, wa, n_max); break;
                case 4: hipLaunchKernelGGL(attn_wsum_mq_kernel<4>, g, dim3(WG), lds, (hipStream_t)stream, wa, n_max); break;
                case 5: hipLaunchKernelGGL(attn_wsum_mq_kernel<5>, g, dim3(WG), lds, (hipStream_t)stream, wa, n_max); break;
                case 10: hipLaunchKernelGGL(attn_wsum_mq_kernel<10>, g, dim3(WG), lds, (hipStream_t)stream, wa, n_max); break;
                default: hipLaunchKernelGGL(attn_wsum_mq_kernel<8>, g, dim3(WG), lds, (hipStream_t)stream, wa, n_max); break;
            }
            return cvc_launch_status();
        }
    }
    if (raw_scale != 0.f) return CVC_E_TOOBIG;          // (the one-query kernel has no raw-weights form)
    const size_t lds2 = (4 * 64 * 4 + 16 + n_max) * sizeof(float);
    if (lds2 > 64 * 1024) return CVC_E_TOOBIG;
    dim3 g2((R + 255) / 256, nclip * nq);
    static const int hoist1_env = [] { const char* e = getenv("CVC_WSUM_HOIST"); return e ? atoi(e) : -1; }();      // A/B: 0
    if (n_max <= 2 * WG && hoist1_env != 0)
        hipLaunchKernelGGL(attn_wsum_kernel<true>, g2, dim3(WG), lds2 + (size_t)n_max * sizeof(float), (hipStream_t)stream, wa, n_max);
    else
        hipLaunchKernelGGL(attn_wsum_kernel<false>, g2, dim3(WG), lds2, (hipStream_t)stream, wa, n_max);
    return cvc_launch_status();
}

// out[row, :] = scale * sum_n w[row, n] X[clip(row), n, :] for nq >= 2 rows per clip: X streamed once per group of queries (the
// several-queries weighted sum without its softmax).  d_q of dot-product attention: w = d_scores, X = proj_context, scale = 1 / temp.
extern "C" int cvc_attn_weighted_rows(const float* w, const float* X, int nclip, int nq, int n, int R, float scale, float* out,
                                      cvc_stream_t stream) {
    if (!w || !X || !out || scale == 0.f || n < 1) return CVC_E_BADARG;
    cvc_attn_set s{};
    s.proj = X; s.ctx = X; s.scores = const_cast<float*>(w); s.attn = const_cast<float*>(w); s.n = n;
    return wsum_impl(&s, 1, nclip, nq, R, out, 0, stream, 0, nullptr, scale);
}

extern "C" int cvc_attn_fwd(int kind, const float* q, const float* w_a, const float* b_a, float inv_temp,
                            const cvc_attn_set* sets, int nsets, int nclip, int nq, int A, int R,
                            float* ctx_sum, cvc_stream_t stream) {
    int rc = cvc_attn_scores(kind, q, w_a, b_a, inv_temp, sets, nsets, nclip, nq, A, stream);
    if (rc) return rc;
    return cvc_attn_wsum(sets, nsets, nclip, nq, R, ctx_sum, stream);
}
