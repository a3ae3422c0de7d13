// Attention backward with score recompute, and the grounder.
//
// Nothing of size [rows, n, A] is ever saved by the forward (the reference's autograd keeps the
// tanh output, O(B T (N+F) A)): the backward re-streams proj_context once, recomputes
// tanh(proj_n + q) in registers and reduces d_q over n inside the workgroup.  Three passes:
//   1. d_attn[row, n] = d_ctx[row, :] . ctx[clip, n, :]      (the forward's DOT score pass)
//   2. softmax backward per row: d_s = attn * (d_attn - <attn, d_attn>) + d_fm
//      (the reference fills masks on .data, modules.py:124-144, so autograd routes the full
//       gradient through masked positions; attn == 0 there unless the whole row is masked)
//   3. score backward: d_q[row, :], per-row partials of d_w_alpha, optional d_proj.
// plus the optional outer product d_ctx_feats[clip, n, :] += attn[row, n] d_ctx[row, :].
// Reductions are ordered (LDS, fixed wave order; no float atomics): run-to-run deterministic.
#include "attn_scores.h"
#include <math.h>

namespace {

constexpr int WG = 256;

__device__ __forceinline__ float block_sum4(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(WG) void softmax_bwd_kernel(const float* attn, const float* d_fm, int n, int have_da,
                                                         float* d_scores) {
    __shared__ float red[4];
    const size_t o = (size_t)blockIdx.x * n;
    float dot = 0.f;
    if (have_da)
        for (int i = threadIdx.x; i < n; i += WG) dot += attn[o + i] * d_scores[o + i];
    dot = block_sum4(dot, red);
    for (int i = threadIdx.x; i < n; i += WG) {
        float ds = have_da ? attn[o + i] * (d_scores[o + i] - dot) : 0.f;
        if (d_fm != nullptr) ds += d_fm[o + i];
        d_scores[o + i] = ds;
    }
}

struct ScoreBwdArgs {
    const float* q;        // [rows, A]
    const float* w_a;      // [A]
    const float* proj;     // [nclip, n, A]
    const float* d_scores; // [rows, n]
    float* d_q;            // [rows, A]
    float* d_w_part;       // [rows, A] or null
    float* d_proj;         // [nclip, n, A] accumulate, or null
    float inv_temp;
    int nq, n, A;
};

// WANT_DP: also accumulate into d_proj (the features carry gradient: encoder in the loop).  Rows are taken 8 at a time per wave
// with all 8 loads requested before the first use (the rolled form kept 4 KB per wave in flight: 1.6 TB/s, latency-bound).
template <int KIND, bool WANT_DP>
__global__ __launch_bounds__(WG) void attn_score_bwd_kernel(ScoreBwdArgs a) {
    __shared__ f32x4 part[2][4][64];
    const int clip = blockIdx.y, cb = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A = a.A, n = a.n;
    const int col = cb * 256 + lane * 4;
    const bool ok = col < A;
    const float* P = a.proj + (size_t)clip * n * A + col;
    float* dP = WANT_DP ? a.d_proj + (size_t)clip * n * A + col : nullptr;
    f32x4 w4 = {0, 0, 0, 0};
    if (KIND == CVC_ATTN_ADDITIVE && ok) w4 = ld4(a.w_a + col);
    for (int qi = 0; qi < a.nq; ++qi) {
        const size_t row = (size_t)clip * a.nq + qi;
        f32x4 q4 = ok ? ld4(a.q + row * A + col) : f32x4{0, 0, 0, 0};
        f32x4 dq = {0, 0, 0, 0}, dw = {0, 0, 0, 0};
        const float* ds_row = a.d_scores + row * n;
        auto one = [&](const float ds, const f32x4 p, const int i) __attribute__((always_inline)) {
            f32x4 dpre;
            if (KIND == CVC_ATTN_ADDITIVE) {
                f32x4 t;
                t.x = fast_tanh(p.x + q4.x); t.y = fast_tanh(p.y + q4.y);
                t.z = fast_tanh(p.z + q4.z); t.w = fast_tanh(p.w + q4.w);
                dpre = ds * w4 * (1.f - t * t);
                dw += ds * t;
                dq += dpre;
            } else {
                const float g = ds * a.inv_temp;
                dpre = g * q4;
                dq += g * p;
            }
            if constexpr (WANT_DP) st4(dP + (size_t)i * A, ld4(dP + (size_t)i * A) + dpre);
        };
        if (ok) {
            int i = wave;
            for (; i + 28 < n; i += 32) {
                f32x4 p[8];
                float ds[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { p[k] = ld4(P + (size_t)(i + 4 * k) * A); ds[k] = ds_row[i + 4 * k]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) one(ds[k], p[k], i + 4 * k);
            }
            for (; i < n; i += 4) one(ds_row[i], ld4(P + (size_t)i * A), i);
        }
        __syncthreads();
        part[0][wave][lane] = dq;
        part[1][wave][lane] = dw;
        __syncthreads();
        if (wave == 0 && ok) {
            st4(a.d_q + row * A + col, (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]));
            if (KIND == CVC_ATTN_ADDITIVE && a.d_w_part != nullptr)
                st4(a.d_w_part + row * A + col,
                    (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]));
        }
    }
}

__global__ __launch_bounds__(WG) void ctxfeat_bwd_kernel(const float* attn, const float* d_ctx, int nq, int n, int R,
                                                         float* d_feat) {
    const int clip = blockIdx.y, cb = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = cb * 256 + lane * 4;
    if (col >= R) return;
    float* D = d_feat + (size_t)clip * n * R + col;
    for (int qi = 0; qi < nq; ++qi) {
        const size_t row = (size_t)clip * nq + qi;
        const f32x4 g = ld4(d_ctx + row * R + col);
        for (int i = wave; i < n; i += 4) st4(D + (size_t)i * R, ld4(D + (size_t)i * R) + attn[row * n + i] * g);
    }
}

// d_feat[clip, i, :] += sum_q attn[q * sa_q + clip * sa_c + i] * g[(q * sg_q + clip * sg_c) * R + :]   -- the context-feature gradient
// of ALL nq queries of a clip in one pass: a workgroup = (256 columns, a chunk of feature rows, clip) stages the nq gradient rows'
// columns in LDS once, then every wave walks its feature rows with the nq weights of a row as scalars -- each row of d_feat is read
// and written ONCE (ctxfeat_bwd_kernel re-reads and re-writes the whole tensor per query: 20 x 2 x 304 MB at config 3 with T = 20).
// Strides cover both callers: the training loop's arenas ([T][B][n] weights, [T][128][R] gradients: sa_q = B n, sa_c = n,
// sg_q = 128, sg_c = 1) and cvc_attn_bwd's [clip][q] rows (sa_q = n, sa_c = nq n, sg_q = 1, sg_c = nq).
constexpr int CFB_ROWS = 64, CFB_MAXQ = 32;
__global__ __launch_bounds__(WG) void ctxfeat_bwd_batched_kernel(const float* attn, long long sa_q, long long sa_c, const float* g,
                                                                 long long sg_q, long long sg_c, int nq, int n, int R, float* d_feat,
                                                                 float scale) {
    __shared__ f32x4 gs[CFB_MAXQ][64];
    const int cb = blockIdx.x, chunk = blockIdx.y, clip = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = cb * 256 + lane * 4;
    const bool ok = col < R;
    for (int q = wave; q < nq; q += 4)
        gs[q][lane] = ok ? ld4(g + ((size_t)q * sg_q + (size_t)clip * sg_c) * R + col) : f32x4{0, 0, 0, 0};
    __syncthreads();
    const int i_end = min(n, (chunk + 1) * CFB_ROWS);
    const float* aw = attn + (size_t)clip * sa_c;
    for (int i = chunk * CFB_ROWS + wave; i < i_end; i += 4) {
        f32x4 acc = {0, 0, 0, 0};
        for (int q = 0; q < nq; ++q) acc += aw[(size_t)q * sa_q + i] * gs[q][lane];
        if (ok) {
            float* D = d_feat + ((size_t)clip * n + i) * R + col;
            st4(D, ld4(D) + scale * acc);
        }
    }
}

// d_proj[clip, i, :] += sum_t d_s[t][clip][i] * w * (1 - tanh^2(P[clip, i, :] + q_t[clip, :]))   (additive attention, all T steps of
// the training loop in one pass): the projected-feature gradient of captioner.py:242-270's T decoder steps.  Per step it is a
// read-modify-write of [B, n, A] inside the score backward; here a workgroup = (256 columns, a chunk of feature rows, clip) stages the
// T queries' columns in LDS (K-slice planes of the h2attn product summed on load, + bias), then every wave walks its rows: the row's
// P once, T tanh evaluations per element in registers, ONE read-modify-write of d_proj.
__global__ __launch_bounds__(WG) void dproj_steps_kernel(const float* q, long long q_step, long long q_plane, int q_nplanes, const float* q_bias,
                                                         const float* w_a, const float* proj, const float* ds, int T, int B, int n, int A,
                                                         float* d_proj) {
    __shared__ f32x4 qs[CFB_MAXQ][64];
    const int cb = blockIdx.x, chunk = blockIdx.y, clip = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = cb * 256 + lane * 4;
    const bool ok = col < A;
    for (int t = wave; t < T; t += 4) {
        f32x4 v = {0, 0, 0, 0};
        if (ok) {
            const float* qp = q + (size_t)t * q_step + (size_t)clip * A + col;
            v = ld4(qp);
            for (int k = 1; k < q_nplanes; ++k) v += ld4(qp + (size_t)k * q_plane);
            if (q_bias != nullptr) v += ld4(q_bias + col);
        }
        qs[t][lane] = v;
    }
    __syncthreads();
    if (!ok) return;
    const f32x4 w4 = ld4(w_a + col);
    const int i_end = min(n, (chunk + 1) * CFB_ROWS);
    for (int i = chunk * CFB_ROWS + wave; i < i_end; i += 4) {
        const f32x4 p = ld4(proj + ((size_t)clip * n + i) * A + col);
        f32x4 acc = {0, 0, 0, 0};
        for (int t = 0; t < T; ++t) {
            const float d = ds[((size_t)t * B + clip) * n + i];
            const f32x4 x = p + qs[t][lane];
            f32x4 th;
            th.x = fast_tanh(x.x); th.y = fast_tanh(x.y); th.z = fast_tanh(x.z); th.w = fast_tanh(x.w);
            acc += d * (1.f - th * th);
        }
        float* D = d_proj + ((size_t)clip * n + i) * A + col;
        st4(D, ld4(D) + w4 * acc);
    }
}

// out[clip, i, :] = sum_q w[clip][q * ws_q + i * ws_i] * rows[clip, q, :]   (i < ni, q < nq; rows [nclip, nq, R], out [nclip, ni, R])
// -- the two small per-clip products of the grounder's backward: a workgroup = (256 columns, output row i, clip); waves split q
__global__ __launch_bounds__(WG) void weighted_rows_kernel(const float* w, long long w_clip, int ws_q, int ws_i, const float* rows,
                                                           int nq, int ni, int R, float* out) {
    __shared__ f32x4 part[4][64];
    const int cb = blockIdx.x, i = blockIdx.y, clip = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = cb * 256 + lane * 4;
    const bool ok = col < R;
    const float* W = w + (size_t)clip * w_clip + (size_t)i * ws_i;
    const float* X = rows + (size_t)clip * nq * R + col;
    f32x4 acc = {0, 0, 0, 0};
    if (ok)
        for (int q = wave; q < nq; q += 4) acc += W[(size_t)q * ws_q] * ld4(X + (size_t)q * R);
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && ok)
        st4(out + ((size_t)clip * ni + i) * R + col, (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}

__global__ __launch_bounds__(WG) void grounder_epilogue_kernel(const float* bias, const uint8_t* mask, size_t total,
                                                               float* out) {
    const size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i >= total) return;
    float v = out[i];
    if (bias != nullptr) v += bias[i];
    if (mask != nullptr && mask[i] != 0) v = CVC_MIN_VALUE;
    out[i] = v;
}

// context-feature gradient of the [clip][q] row layout (cvc_attn_bwd): one pass for several queries per clip, the per-query kernel for one
inline void ctxfeat_rows(const float* attn, const float* d_ctx, int nclip, int nq, int n, int R, float* d_ctxfeat, hipStream_t st) {
    if (nq > 1 && nq <= CFB_MAXQ)
        hipLaunchKernelGGL(ctxfeat_bwd_batched_kernel, dim3((R + 255) / 256, (n + CFB_ROWS - 1) / CFB_ROWS, nclip), dim3(WG), 0, st, attn,
                           (long long)n, (long long)nq * n, d_ctx, 1LL, (long long)nq, nq, n, R, d_ctxfeat, 1.0f);
    else
        hipLaunchKernelGGL(ctxfeat_bwd_kernel, dim3((R + 255) / 256, nclip), dim3(WG), 0, st, attn, d_ctx, nq, n, R, d_ctxfeat);
}

}  // namespace

extern "C" int cvc_attn_bwd(int kind, const float* q, const float* w_a, float inv_temp, const float* proj,
                            const float* ctx, const float* attn, const float* d_ctx, const float* d_fm, int nclip, int nq,
                            int n, int A, int R, float* d_scores, float* d_q, float* d_w_part, float* d_proj,
                            float* d_ctxfeat, cvc_stream_t stream) {
    if (!q || !proj || !ctx || !attn || !d_scores || !d_q || nclip < 1 || nq < 1 || n < 1 || (A & 3) || (R & 3))
        return CVC_E_BADARG;
    if (kind != CVC_ATTN_ADDITIVE && kind != CVC_ATTN_DOT) return CVC_E_BADARG;
    if (kind == CVC_ATTN_ADDITIVE && !w_a) return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int rows = nclip * nq;
    if (d_ctx != nullptr) {
        cvc_attn_set s{};
        s.proj = ctx; s.ctx = ctx; s.scores = d_scores; s.attn = d_scores; s.n = n;
        int rc = run_scores(CVC_ATTN_DOT, d_ctx, nullptr, nullptr, 1.f, &s, 1, nclip, nq, R, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(WG), 0, st, attn, d_fm, n, d_ctx != nullptr ? 1 : 0, d_scores);
    if (kind == CVC_ATTN_DOT && nq > 1 && (d_proj == nullptr || nq <= CFB_MAXQ)) {
        // several dot-product queries per clip (the T localizer queries): d_q[row, :] = (1 / temp) sum_n d_s[row, n] P[clip, n, :]
        // is the several-queries weighted sum with d_s as the weights -- the clip's rows streamed once per group of queries instead
        // of once per query (20 passes at T = 20: 363 us for the frame features of config 3)
        int rc = cvc_attn_weighted_rows(d_scores, proj, nclip, nq, n, A, inv_temp, d_q, stream);
        if (rc != CVC_E_TOOBIG) {
            if (rc) return rc;
            // d_proj[clip, i, :] += (1 / temp) sum_q d_s[clip, q, i] q[clip, q, :]: the same one-pass weighted rows with d_s as the
            // weights and the queries as the rows (per query it was a read-modify-write of [nclip, n, A]: 20 passes at T = 20)
            if (d_proj != nullptr)
                hipLaunchKernelGGL(ctxfeat_bwd_batched_kernel, dim3((A + 255) / 256, (n + CFB_ROWS - 1) / CFB_ROWS, nclip), dim3(WG), 0, st,
                                   d_scores, (long long)n, (long long)nq * n, q, 1LL, (long long)nq, nq, n, A, d_proj, inv_temp);
            if (d_ctxfeat != nullptr && d_ctx != nullptr) ctxfeat_rows(attn, d_ctx, nclip, nq, n, R, d_ctxfeat, st);
            return cvc_launch_status();
        }
    }
    ScoreBwdArgs a{q, w_a, proj, d_scores, d_q, d_w_part, d_proj, inv_temp, nq, n, A};
    dim3 grid((A + 255) / 256, nclip);
    if (kind == CVC_ATTN_ADDITIVE) {
        if (d_proj != nullptr) hipLaunchKernelGGL((attn_score_bwd_kernel<CVC_ATTN_ADDITIVE, true>), grid, dim3(WG), 0, st, a);
        else hipLaunchKernelGGL((attn_score_bwd_kernel<CVC_ATTN_ADDITIVE, false>), grid, dim3(WG), 0, st, a);
    } else {
        if (d_proj != nullptr) hipLaunchKernelGGL((attn_score_bwd_kernel<CVC_ATTN_DOT, true>), grid, dim3(WG), 0, st, a);
        else hipLaunchKernelGGL((attn_score_bwd_kernel<CVC_ATTN_DOT, false>), grid, dim3(WG), 0, st, a);
    }
    if (d_ctxfeat != nullptr && d_ctx != nullptr) ctxfeat_rows(attn, d_ctx, nclip, nq, n, R, d_ctxfeat, st);
    return cvc_launch_status();
}

// the training loop's form: every step's weights [T][B][n] and context gradients [T][128][R] (cvc_train_loop.d_ctx_all)
extern "C" int cvc_ctxfeat_bwd_steps(const float* attn, const float* d_ctx_all, int T, int B, int n, int R, float* d_feat, cvc_stream_t stream) {
    if (!attn || !d_ctx_all || !d_feat || T < 1 || T > CFB_MAXQ || B < 1 || n < 1 || (R & 3)) return CVC_E_BADARG;
    hipLaunchKernelGGL(ctxfeat_bwd_batched_kernel, dim3((R + 255) / 256, (n + CFB_ROWS - 1) / CFB_ROWS, B), dim3(WG), 0, (hipStream_t)stream,
                       attn, (long long)B * n, (long long)n, d_ctx_all, 128LL, 1LL, T, n, R, d_feat, 1.0f);
    return cvc_launch_status();
}

extern "C" int cvc_dproj_bwd_steps(const float* q, long long q_step, long long q_plane, int q_nplanes, const float* q_bias, const float* w_a,
                                   const float* proj, const float* ds, int T, int B, int n, int A, float* d_proj, cvc_stream_t stream) {
    if (!q || !w_a || !proj || !ds || !d_proj || T < 1 || T > CFB_MAXQ || B < 1 || n < 1 || (A & 3) || q_nplanes < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(dproj_steps_kernel, dim3((A + 255) / 256, (n + CFB_ROWS - 1) / CFB_ROWS, B), dim3(WG), 0, (hipStream_t)stream, q, q_step,
                       q_plane, q_nplanes, q_bias, w_a, proj, ds, T, B, n, A, d_proj);
    return cvc_launch_status();
}

// ---- both feature sets of a decoder step in one backward (the C-driven training loop, csrc/train_driver.hip) ----------------
namespace {

struct SoftmaxBwd2Args {
    const float* attn[2];
    const float* d_fm[2];
    float* d_scores[2];
    int n[2];
    int have_da;
};

// grid (rows, sets): softmax_bwd_kernel for set blockIdx.y
__global__ __launch_bounds__(WG) void softmax_bwd2_kernel(SoftmaxBwd2Args a) {
    __shared__ float red[4];
    const bool s1 = blockIdx.y != 0;
    const float* attn = s1 ? a.attn[1] : a.attn[0];
    const float* d_fm = s1 ? a.d_fm[1] : a.d_fm[0];
    float* d_scores = s1 ? a.d_scores[1] : a.d_scores[0];
    const int n = s1 ? a.n[1] : a.n[0];
    const size_t o = (size_t)blockIdx.x * n;
    float dot = 0.f;
    if (a.have_da)
        for (int i = threadIdx.x; i < n; i += WG) dot += attn[o + i] * d_scores[o + i];
    dot = block_sum4(dot, red);
    for (int i = threadIdx.x; i < n; i += WG) {
        float ds = a.have_da ? attn[o + i] * (d_scores[o + i] - dot) : 0.f;
        if (d_fm != nullptr) ds += d_fm[o + i];
        d_scores[o + i] = ds;
    }
}

struct ScoreBwd2Args {
    cvc_grad_src q;        // [rows, A], possibly as the split-K planes of the h2attn product (+ q_bias)
    const float* q_bias;   // [A] or null
    const float* w_a;      // [A]
    const float* proj[2];  // [nclip, n_s, A]
    const float* d_scores[2];
    float* d_proj[2];      // accumulate, or null
    int n[2];
    int nsets;
    float* d_q;            // [rows, A]
    float* d_q_q;          // [A/4][64][4] or null
    float* d_w_part;       // [rows, A] or null
    float inv_temp;
    int nq, A;
};

// attn_score_bwd_kernel over both sets: the query row's d_q (and the d_w_alpha partial) is the sum over the sets, taken in set
// order inside the workgroup -- no second launch, no add.  Same row batching as the one-set kernel.
// (The projected features are read with ordinary loads whatever the sets' cache policy says: non-temporal loads here measured
// 92 against 88 us for the backward pair at config 3.)
template <int KIND, bool WANT_DP>
__global__ __launch_bounds__(WG) void attn_score_bwd2_kernel(ScoreBwd2Args a) {
    __shared__ f32x4 part[2][4][64];
    const int clip = blockIdx.y, cb = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A = a.A;
    const int col = cb * 256 + lane * 4;
    const bool ok = col < A;
    f32x4 w4 = {0, 0, 0, 0};
    if (KIND == CVC_ATTN_ADDITIVE && ok) w4 = ld4(a.w_a + col);
    for (int qi = 0; qi < a.nq; ++qi) {
        const size_t row = (size_t)clip * a.nq + qi;
        f32x4 q4 = {0, 0, 0, 0};
        if (ok) {
            const float* qp = a.q.p + row * a.q.ld + col;
            q4 = ld4(qp);
            for (int k = 1; k < a.q.nplanes; ++k) q4 += ld4(qp + (size_t)k * a.q.plane_stride);
            if (a.q_bias != nullptr) q4 += ld4(a.q_bias + col);
        }
        f32x4 dq = {0, 0, 0, 0}, dw = {0, 0, 0, 0};
        for (int s = 0; s < a.nsets; ++s) {
            const int n = s ? a.n[1] : a.n[0];
            const float* P = (s ? a.proj[1] : a.proj[0]) + (size_t)clip * n * A + col;
            float* dP = nullptr;
            if constexpr (WANT_DP) {
                dP = s ? a.d_proj[1] : a.d_proj[0];
                if (dP != nullptr) dP += (size_t)clip * n * A + col;
            }
            const float* ds_row = (s ? a.d_scores[1] : a.d_scores[0]) + row * n;
            auto one = [&](const float ds, const f32x4 p, const int i) __attribute__((always_inline)) {
                f32x4 dpre;
                if (KIND == CVC_ATTN_ADDITIVE) {
                    f32x4 t;
                    t.x = fast_tanh(p.x + q4.x); t.y = fast_tanh(p.y + q4.y);
                    t.z = fast_tanh(p.z + q4.z); t.w = fast_tanh(p.w + q4.w);
                    dpre = ds * w4 * (1.f - t * t);
                    dw += ds * t;
                    dq += dpre;
                } else {
                    const float g = ds * a.inv_temp;
                    dpre = g * q4;
                    dq += g * p;
                }
                if constexpr (WANT_DP) {
                    if (dP != nullptr) st4(dP + (size_t)i * A, ld4(dP + (size_t)i * A) + dpre);
                }
            };
            if (ok) {
                int i = wave;
                for (; i + 28 < n; i += 32) {
                    f32x4 p[8];
                    float ds[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { p[k] = ld4(P + (size_t)(i + 4 * k) * A); ds[k] = ds_row[i + 4 * k]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) one(ds[k], p[k], i + 4 * k);
                }
                for (; i < n; i += 4) one(ds_row[i], ld4(P + (size_t)i * A), i);
            }
        }
        __syncthreads();
        part[0][wave][lane] = dq;
        part[1][wave][lane] = dw;
        __syncthreads();
        if (wave == 0 && ok) {
            const f32x4 v = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
            st4(a.d_q + row * A + col, v);
            if (a.d_q_q != nullptr) st4(a.d_q_q + ((size_t)(col >> 2) * 64 + row) * 4, v);
            if (KIND == CVC_ATTN_ADDITIVE && a.d_w_part != nullptr)
                st4(a.d_w_part + row * A + col,
                    (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]));
        }
    }
}

}  // namespace

extern "C" int cvc_attn_bwd_pair(int kind, const cvc_grad_src* q, const float* q_bias, const float* w_a, float inv_temp,
                                 const cvc_attn_set* sets, int nsets, const cvc_grad_src* d_ctx_src, int nclip, int nq, int A, int R,
                                 float* d_q, float* d_q_q, float* d_w_part, float* const* d_proj, float* const* d_ctxfeat,
                                 cvc_stream_t stream) {
    if (!q || !q->p || q->nplanes < 1 || q->ld < A || (q->ld & 3) || (q->plane_stride & 3)) return CVC_E_BADARG;
    const float* d_ctx = d_ctx_src != nullptr ? d_ctx_src->p : nullptr;
    if (d_ctx != nullptr && (d_ctx_src->nplanes < 1 || d_ctx_src->ld < R)) return CVC_E_BADARG;
    if (d_ctx != nullptr && d_ctxfeat != nullptr && (d_ctx_src->nplanes != 1 || d_ctx_src->ld != R)) return CVC_E_BADARG;
    if (!sets || nsets < 1 || nsets > 2 || !d_q || nclip < 1 || nq < 1 || (A & 3) || (R & 3)) return CVC_E_BADARG;
    if (kind != CVC_ATTN_ADDITIVE && kind != CVC_ATTN_DOT) return CVC_E_BADARG;
    if (kind == CVC_ATTN_ADDITIVE && !w_a) return CVC_E_BADARG;
    const int rows = nclip * nq;
    if (d_q_q != nullptr && rows > 64) return CVC_E_BADARG;
    for (int s = 0; s < nsets; ++s)
        if (!sets[s].proj || !sets[s].ctx || !sets[s].attn || !sets[s].scores || sets[s].n < 1) return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (d_ctx != nullptr) {
        // d_attn[row, i] = d_ctx[row, :] . ctx[clip, i, :] for both sets: the forward's dot-product score pass over the contexts
        cvc_attn_set ss[2]{};
        for (int s = 0; s < nsets; ++s) {
            ss[s].proj = sets[s].ctx; ss[s].ctx = sets[s].ctx; ss[s].scores = sets[s].scores; ss[s].attn = sets[s].scores;
            ss[s].n = sets[s].n;
            ss[s].stream = (sets[s].stream & 2) ? 1 : 0;       // the contexts' cache policy (they are this pass's streamed operand)
        }
        int rc = run_scores(CVC_ATTN_DOT, d_ctx, nullptr, nullptr, 1.f, ss, nsets, nclip, nq, R, st, d_ctx_src->nplanes, nullptr,
                            nullptr, d_ctx_src->ld, d_ctx_src->plane_stride);
        if (rc) return rc;
    }
    SoftmaxBwd2Args sm{};
    ScoreBwd2Args a{};
    for (int s = 0; s < 2; ++s) {
        const cvc_attn_set& S = sets[s < nsets ? s : 0];
        sm.attn[s] = S.attn; sm.d_fm[s] = S.frame_masked; sm.d_scores[s] = S.scores; sm.n[s] = S.n;
        a.proj[s] = S.proj; a.d_scores[s] = S.scores; a.n[s] = S.n;
        a.d_proj[s] = (d_proj != nullptr && s < nsets) ? d_proj[s] : nullptr;
    }
    sm.have_da = d_ctx != nullptr ? 1 : 0;
    hipLaunchKernelGGL(softmax_bwd2_kernel, dim3(rows, nsets), dim3(WG), 0, st, sm);
    a.q = *q; a.q_bias = q_bias; a.w_a = w_a; a.nsets = nsets; a.d_q = d_q; a.d_q_q = d_q_q; a.d_w_part = d_w_part; a.inv_temp = inv_temp; a.nq = nq; a.A = A;
    dim3 grid((A + 255) / 256, nclip);
    const bool want_dp = a.d_proj[0] != nullptr || a.d_proj[1] != nullptr;
    if (kind == CVC_ATTN_ADDITIVE) {
        if (want_dp) hipLaunchKernelGGL((attn_score_bwd2_kernel<CVC_ATTN_ADDITIVE, true>), grid, dim3(WG), 0, st, a);
        else hipLaunchKernelGGL((attn_score_bwd2_kernel<CVC_ATTN_ADDITIVE, false>), grid, dim3(WG), 0, st, a);
    } else {
        if (want_dp) hipLaunchKernelGGL((attn_score_bwd2_kernel<CVC_ATTN_DOT, true>), grid, dim3(WG), 0, st, a);
        else hipLaunchKernelGGL((attn_score_bwd2_kernel<CVC_ATTN_DOT, false>), grid, dim3(WG), 0, st, a);
    }
    if (d_ctxfeat != nullptr && d_ctx != nullptr)
        for (int s = 0; s < nsets; ++s)
            if (d_ctxfeat[s] != nullptr)
                hipLaunchKernelGGL(ctxfeat_bwd_kernel, dim3((R + 255) / 256, nclip), dim3(WG), 0, st, sets[s].attn, d_ctx, nq,
                                   sets[s].n, R, d_ctxfeat[s]);
    return cvc_launch_status();
}

// d_xt[b,t,:] = sum_n d[b,t,n] feats[b,n,:],  d_feats[b,n,:] = sum_t d[b,t,n] xt[b,t,:]   (either output may be null)
extern "C" int cvc_grounder_bwd(const float* d, const float* xt, const float* feats, int B, int T, int N, int G, float* d_xt,
                                float* d_feats, cvc_stream_t stream) {
    if (!d || !xt || !feats || B < 1 || T < 1 || N < 1 || (G & 3) || (!d_xt && !d_feats)) return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const long long dclip = (long long)T * N;
    if (d_xt != nullptr)         // q = region n (stride 1 in d), i = step t (stride N)
        hipLaunchKernelGGL(weighted_rows_kernel, dim3((G + 255) / 256, T, B), dim3(WG), 0, st, d, dclip, 1, N, feats, N, T, G, d_xt);
    if (d_feats != nullptr)      // q = step t (stride N), i = region n (stride 1)
        hipLaunchKernelGGL(weighted_rows_kernel, dim3((G + 255) / 256, N, B), dim3(WG), 0, st, d, dclip, N, 1, xt, T, N, G, d_feats);
    return cvc_launch_status();
}

extern "C" int cvc_grounder_fwd(const float* xt, const float* feats, const float* bias, const uint8_t* mask, int B, int T,
                                int N, int G, float* out, cvc_stream_t stream) {
    if (!xt || !feats || !out || B < 1 || T < 1 || N < 1 || (G & 3)) return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    cvc_attn_set s{};
    s.proj = feats; s.ctx = feats; s.scores = out; s.attn = out; s.n = N;
    int rc = run_scores(CVC_ATTN_DOT, xt, nullptr, nullptr, 1.f, &s, 1, B, T, G, st);
    if (rc) return rc;
    if (bias != nullptr || mask != nullptr) {
        const size_t total = (size_t)B * T * N;
        hipLaunchKernelGGL(grounder_epilogue_kernel, dim3((unsigned)((total + WG - 1) / WG)), dim3(WG), 0, st, bias, mask,
                           total, out);
    }
    return cvc_launch_status();
}
