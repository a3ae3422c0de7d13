// Word-embedding lookup, vocabulary head (log-softmax, greedy top-2 with UNK suppression,
// masked NLL and its fused backward), beam bookkeeping and LSTM pointwise backward.
// Reference: model/captioner.py:53-68 (embed), :266/:361/:437 (log_softmax(logit)),
// :415-422 (top-2 / UNK rule), misc/utils.py:132-146,181-192 (masked NLL).
// All of it is small elementwise / row-reduction work next to the streaming kernels; the
// point is to keep it on the device and inside the captured decode graph.
#include "cvc_common.h"
#include "gsk.h"
#include "dropout_rng.h"
#include <math.h>

namespace {

constexpr int WG = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------ embedding
__global__ __launch_bounds__(WG) void embed_relu_fwd_kernel(const float* table, const int64_t* idx, const float* drop, DropSpec rng,
                                                            int M, int E, float* out) {
    const int m = blockIdx.y;
    const int e = (blockIdx.x * WG + threadIdx.x) * 4;
    if (e >= E) return;
    f32x4 v = ld4(table + (size_t)idx[m] * E + e);
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    if (drop != nullptr) v *= ld4(drop + (size_t)m * E + e);
    if (rng.state != nullptr) {                               // mask of element (m, e) generated here (dropout_rng.h)
        const uint32_t s0 = rng.state[0], s1 = rng.state[1], s2 = rng.state[2], i0 = (uint32_t)m * (uint32_t)E + (uint32_t)e;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= cvc_drop_mult(rng, s0, s1, s2, i0 + k);
    }
    st4(out + (size_t)m * E + e, v);
}

// y = x * mask(site, element index): nn.Dropout with the counter-based masks, for the sites no producing kernel fuses
__global__ __launch_bounds__(WG) void dropout_rng_kernel(const float* x, DropSpec rng, size_t n, float* y) {
    const size_t i = ((size_t)blockIdx.x * WG + threadIdx.x) * 4;
    if (i >= n) return;
    const uint32_t s0 = rng.state[0], s1 = rng.state[1], s2 = rng.state[2];
    if (i + 4 <= n) {
        f32x4 v = ld4(x + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= cvc_drop_mult(rng, s0, s1, s2, (uint32_t)(i + k));
        st4(y + i, v);
    } else {
        for (size_t k = i; k < n; ++k) y[k] = x[k] * cvc_drop_mult(rng, s0, s1, s2, (uint32_t)k);
    }
}

// Deterministic scatter-add without atomics and without a host round trip: the host passes a stable argsort
// of the word indices, so the rows of one word are contiguous in `order` (a "run").  Two passes, both in a fixed
// order: (1) the sorted positions are cut into chunks of EMB_CHUNK; a workgroup walks its chunk and writes one
// partial sum per (word, chunk) piece to part[first sorted position of the piece]; (2) the workgroup of a run's
// first position adds the run's pieces (its own, then the ones starting at the following chunk boundaries) and
// writes that word's gradient row.  A long run -- the BOS / padding word owns about a third of the B*T rows --
// is thereby summed by run/EMB_CHUNK workgroups in parallel instead of one workgroup walking it row by row.
constexpr int EMB_CHUNK = 16;

__global__ __launch_bounds__(WG) void embed_bwd_pieces_kernel(const int64_t* idx, const int64_t* order, const float* drop, DropSpec rng,
                                                              const float* d_out, int M, int E, float* part) {
    const int r0 = blockIdx.y * EMB_CHUNK;
    const int e = (blockIdx.x * WG + threadIdx.x) * 4;
    if (e >= E) return;
    const int r1 = min(M, r0 + EMB_CHUNK);
    int start = r0;
    int64_t w = idx[order[r0]];
    f32x4 acc = {0, 0, 0, 0};
    for (int r = r0; r < r1; ++r) {
        const int64_t m = order[r];
        const int64_t wr = idx[m];
        if (wr != w) {                                         // the piece ends: flush it, open the next one
            st4(part + (size_t)start * E + e, acc);
            acc = f32x4{0, 0, 0, 0};
            start = r;
            w = wr;
        }
        f32x4 g = ld4(d_out + (size_t)m * E + e);
        if (drop != nullptr) g *= ld4(drop + (size_t)m * E + e);
        if (rng.state != nullptr) {                           // the forward's mask, regenerated
            const uint32_t s0 = rng.state[0], s1 = rng.state[1], s2 = rng.state[2], i0 = (uint32_t)m * (uint32_t)E + (uint32_t)e;
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] *= cvc_drop_mult(rng, s0, s1, s2, i0 + k);
        }
        acc += g;
    }
    st4(part + (size_t)start * E + e, acc);
}

__global__ __launch_bounds__(WG) void embed_bwd_runs_kernel(const float* table, const int64_t* idx, const int64_t* order,
                                                            const float* part, int M, int E, float* d_table) {
    const int r0 = blockIdx.y;
    const int64_t w = idx[order[r0]];
    if (r0 > 0 && idx[order[r0 - 1]] == w) return;            // not the first row of its word
    const int e = (blockIdx.x * WG + threadIdx.x) * 4;
    if (e >= E) return;
    f32x4 acc = ld4(part + (size_t)r0 * E + e);               // the piece that starts the run
    for (int r = (r0 / EMB_CHUNK + 1) * EMB_CHUNK; r < M && idx[order[r]] == w; r += EMB_CHUNK)
        acc += ld4(part + (size_t)r * E + e);                 // pieces that start at the following chunk boundaries
    const f32x4 t = ld4(table + (size_t)w * E + e);
    acc.x = t.x > 0.f ? acc.x : 0.f; acc.y = t.y > 0.f ? acc.y : 0.f;
    acc.z = t.z > 0.f ? acc.z : 0.f; acc.w = t.w > 0.f ? acc.w : 0.f;
    // ACCUMULATED: a fresh gradient starts from the caller's zero fill (0 + x = x exactly); the three embeddings of a cyclical
    // pass (loops A, B, C share the table, captioner.py:53-68) add up in ONE buffer, call after call, instead of in three
    st4(d_table + (size_t)w * E + e, ld4(d_table + (size_t)w * E + e) + acc);
}

// ------------------------------------------------------------------ log-softmax / top-2 / NLL
__global__ __launch_bounds__(WG) void log_softmax_kernel(const float* logits, int V, float* logp) {
    __shared__ float red[4];
    const float* x = logits + (size_t)blockIdx.x * V;
    float* y = logp + (size_t)blockIdx.x * V;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += WG) m = fmaxf(m, x[v]);
    m = block_max(m, red);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += WG) s += expf(x[v] - m);
    s = block_sum(s, red);
    const float lse = m + logf(s);
    for (int v = threadIdx.x; v < V; v += WG) y[v] = x[v] - lse;
}

// d_x = d_y - exp(y) * sum(d_y)   (y = log_softmax(x))
__global__ __launch_bounds__(WG) void log_softmax_bwd_kernel(const float* logp, const float* d_logp, int V, float* d_logits) {
    __shared__ float red[4];
    const size_t o = (size_t)blockIdx.x * V;
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += WG) s += d_logp[o + v];
    s = block_sum(s, red);
    for (int v = threadIdx.x; v < V; v += WG) d_logits[o + v] = d_logp[o + v] - expf(logp[o + v]) * s;
}

// d_logp[m, :] = 0 except d_logp[m, target[m]] = -w[m] * g[0]
__global__ __launch_bounds__(WG) void nll_bwd_kernel(const int64_t* target, const float* w, const float* g, int V,
                                                     float* d_logp) {
    const int m = blockIdx.y;
    const int v = blockIdx.x * WG + threadIdx.x;
    if (v >= V) return;
    d_logp[(size_t)m * V + v] = v == (int)target[m] ? -w[m] * g[0] : 0.f;
}

struct Top2 { float v1; int i1; float v2; int i2; };
// (bitwise, not short-circuit: three compares and two mask operations instead of branches)
__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) { return (va > vb) | ((va == vb) & (ia < ib)); }
__device__ __forceinline__ Top2 merge(Top2 a, Top2 b) {
    Top2 r;
    if (better(a.v1, a.i1, b.v1, b.i1)) {
        r.v1 = a.v1; r.i1 = a.i1;
        if (better(a.v2, a.i2, b.v1, b.i1)) { r.v2 = a.v2; r.i2 = a.i2; } else { r.v2 = b.v1; r.i2 = b.i1; }
    } else {
        r.v1 = b.v1; r.i1 = b.i1;
        if (better(b.v2, b.i2, a.v1, a.i1)) { r.v2 = b.v2; r.i2 = b.i2; } else { r.v2 = a.v1; r.i2 = a.i1; }
    }
    return r;
}

__global__ __launch_bounds__(WG) void top2_unk_kernel(const float* logits, int V, int unk, int64_t* word, int wstride,
                                                      float* logprob) {
    __shared__ float red[4];
    __shared__ Top2 tred[4];
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * V;
    Top2 t{-INFINITY, 0x7fffffff, -INFINITY, 0x7fffffff};
    for (int v = threadIdx.x; v < V; v += WG) {
        const float xv = x[v];
        if (better(xv, v, t.v1, t.i1)) { t.v2 = t.v1; t.i2 = t.i1; t.v1 = xv; t.i1 = v; }
        else if (better(xv, v, t.v2, t.i2)) { t.v2 = xv; t.i2 = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Top2 u;
        u.v1 = __shfl_xor(t.v1, o, 64); u.i1 = __shfl_xor(t.i1, o, 64);
        u.v2 = __shfl_xor(t.v2, o, 64); u.i2 = __shfl_xor(t.i2, o, 64);
        t = merge(t, u);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) tred[wave] = t;
    __syncthreads();
    t = merge(merge(tred[0], tred[1]), merge(tred[2], tred[3]));
    const float m = t.v1;
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += WG) s += expf(x[v] - m);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const bool use2 = (t.i1 == unk) && V > 1;        // captioner.py:417-421
        int w = use2 ? t.i2 : t.i1;
        if ((unsigned)w >= (unsigned)V) w = 0;             // all-NaN row (diverged checkpoint): nothing compares greater; stay in range
        word[(size_t)row * wstride] = w;
        if (logprob != nullptr) logprob[row] = (use2 ? t.v2 : t.v1) - (m + logf(s));
    }
}

// Final stage of the fused vocabulary head: merge the per-block partial records written by the logits
// GEMM epilogue ({top1 v,i, top2 v,i, max, sumexp} per 32-column block and row), apply the UNK rule,
// emit the word, its log-prob, and (optionally) next step's embedded word relu(Emb[word]).
__global__ __launch_bounds__(WG) void top2_final_kernel(const float* part, int nblocks, int unk, int64_t* word, int wstride,
                                                        float* logprob, const float* table, int E, float* emb_out, int emb_ld) {
    __shared__ float red[4];
    __shared__ Top2 tred[4];
    __shared__ int chosen;
    const int row = blockIdx.x;
    Top2 t{-INFINITY, 0x7fffffff, -INFINITY, 0x7fffffff};
    float mx = -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += WG) {
        const float* rec = part + ((size_t)b * 64 + row) * 6;
        Top2 u{rec[0], __float_as_int(rec[1]), rec[2], __float_as_int(rec[3])};
        t = merge(t, u);
        mx = fmaxf(mx, rec[4]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Top2 u;
        u.v1 = __shfl_xor(t.v1, o, 64); u.i1 = __shfl_xor(t.i1, o, 64);
        u.v2 = __shfl_xor(t.v2, o, 64); u.i2 = __shfl_xor(t.i2, o, 64);
        t = merge(t, u);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) tred[wave] = t;
    mx = block_max(mx, red);                       // (contains the barriers that publish tred)
    t = merge(merge(tred[0], tred[1]), merge(tred[2], tred[3]));
    float se = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += WG) {
        const float* rec = part + ((size_t)b * 64 + row) * 6;
        se += rec[5] * expf(rec[4] - mx);
    }
    se = block_sum(se, red);
    if (threadIdx.x == 0) {
        const bool use2 = (t.i1 == unk) && t.i2 != 0x7fffffff;          // captioner.py:417-421
        int w = use2 ? t.i2 : t.i1;
        if (w == 0x7fffffff || w < 0) w = 0;               // all-NaN logits: no record ever compared greater; the word is
                                                           // also a gather address (table + w * E) here and next step
        word[(size_t)row * wstride] = w;
        if (logprob != nullptr) logprob[row] = (use2 ? t.v2 : t.v1) - (mx + logf(se));
        chosen = w;
    }
    if (emb_out != nullptr) {
        __syncthreads();
        const float* src = table + (size_t)chosen * E;
        for (int e = threadIdx.x * 4; e < E; e += WG * 4) {
            f32x4 v = ld4(src + e);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (emb_ld > 0) st4(emb_out + (size_t)row * emb_ld + e, v);
            else st4(emb_out + ((size_t)(e >> 2) * 64 + row) * 4, v);       // emb_ld == 0: quad layout [E/4][64][4]
        }
    }
}

// Word selection straight from the partial tiles of a stream-K vocabulary GEMM (gemm_gsk.hip): one workgroup per batch row sums
// the segments of its 32-column blocks (+ bias), takes top-2 / max / sum-exp over the row and finishes like top2_final_kernel.
// ITEMS float4 items per thread (item = 4 columns of one block), every segment load of every item requested before the first
// sum: the kernel is one round trip to the slabs, not ITEMS x segments of them.
template <int ITEMS>
__global__ __launch_bounds__(WG) void top2_slab_kernel(GskSegs g, const float* bias, int V, int unk, int64_t* word, int wstride,
                                                       float* logprob, const float* table, int E, float* emb_out, int emb_ld) {
    __shared__ float red[4];
    __shared__ Top2 tred[4];
    __shared__ int chosen;
    constexpr int MAXS = 6;
    const int row = blockIdx.x;
    const int nitem = ((V + 31) >> 5) * 8;
    f32x4 sv[ITEMS][MAXS], bv[ITEMS];
    int nseg[ITEMS], col0[ITEMS];
    const float* p0[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int it = min((int)threadIdx.x + k * WG, nitem - 1);          // past the end: re-read the last item (masked below)
        const int blk = it >> 3, rq = it & 7;
        nseg[k] = gsk_nseg(g, blk >> 3);
        col0[k] = (int)threadIdx.x + k * WG < nitem ? blk * 32 + rq * 4 : V;
        p0[k] = gsk_part(g, blk >> 3, 0, blk & 7) + (size_t)row * 32 + rq * 4;
#pragma unroll
        for (int s = 0; s < MAXS; ++s) sv[k][s] = ld4(p0[k] + (size_t)(s < nseg[k] ? s : nseg[k] - 1) * (8 * 2048));
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[k][e] = (bias != nullptr && col0[k] + e < V) ? bias[col0[k] + e] : 0.f;
    }
    Top2 t{-INFINITY, 0x7fffffff, -INFINITY, 0x7fffffff};
    float mx = -INFINITY;
    f32x4 val[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        f32x4 x = sv[k][0];
#pragma unroll
        for (int s = 1; s < MAXS; ++s) x += s < nseg[k] ? sv[k][s] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = MAXS; s < nseg[k]; ++s) x += ld4(p0[k] + (size_t)s * (8 * 2048));
        x += bv[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = col0[k] + e;
            const float xv = v < V ? x[e] : -INFINITY;
            val[k][e] = xv;
            if (v < V) {
                if (better(xv, v, t.v1, t.i1)) { t.v2 = t.v1; t.i2 = t.i1; t.v1 = xv; t.i1 = v; }
                else if (better(xv, v, t.v2, t.i2)) { t.v2 = xv; t.i2 = v; }
                mx = fmaxf(mx, xv);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Top2 u;
        u.v1 = __shfl_xor(t.v1, o, 64); u.i1 = __shfl_xor(t.i1, o, 64);
        u.v2 = __shfl_xor(t.v2, o, 64); u.i2 = __shfl_xor(t.i2, o, 64);
        t = merge(t, u);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) tred[wave] = t;
    mx = block_max(mx, red);                       // (contains the barriers that publish tred)
    t = merge(merge(tred[0], tred[1]), merge(tred[2], tred[3]));
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) se += val[k][e] == -INFINITY ? 0.f : expf(val[k][e] - mx);
    se = block_sum(se, red);
    if (threadIdx.x == 0) {
        const bool use2 = (t.i1 == unk) && t.i2 != 0x7fffffff;          // captioner.py:417-421
        int w = use2 ? t.i2 : t.i1;
        if (w == 0x7fffffff || w < 0) w = 0;               // all-NaN logits: see top2_final_kernel
        word[(size_t)row * wstride] = w;
        if (logprob != nullptr) logprob[row] = (use2 ? t.v2 : t.v1) - (mx + logf(se));
        chosen = w;
    }
    if (emb_out != nullptr) {
        __syncthreads();
        const float* src = table + (size_t)chosen * E;
        for (int e = threadIdx.x * 4; e < E; e += WG * 4) {
            f32x4 v = ld4(src + e);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (emb_ld > 0) st4(emb_out + (size_t)row * emb_ld + e, v);
            else st4(emb_out + ((size_t)(e >> 2) * 64 + row) * 4, v);       // emb_ld == 0: quad layout [E/4][64][4]
        }
    }
}

__global__ __launch_bounds__(WG) void nll_fwd_kernel(const float* logp, const int64_t* target, const float* w, int M, int V,
                                                     float* loss_sum) {
    __shared__ float red[4];
    float s = 0.f;
    for (int m = threadIdx.x; m < M; m += WG) {
        const float wm = w[m];
        if (wm != 0.f) s -= wm * logp[(size_t)m * V + target[m]];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss_sum[0] += s;
}

// Fused vocabulary criterion: one pass over a row of raw logits gives its log-sum-exp, its argmax (ties -> lowest
// index, what torch.max over the log-probs returns) and the row's weighted NLL; no [M, V] log-prob matrix is written.
__global__ __launch_bounds__(WG) void vocab_nll_fwd_kernel(const float* logits, const int64_t* target, const float* w, int V,
                                                           float* lse_out, int64_t* argmax, float* row_loss) {
    __shared__ float red[4];
    __shared__ int ired[4];
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * V;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += WG) {
        const float xv = x[v];
        if (xv > m) { m = xv; mi = v; }                       // ascending v per thread: first occurrence wins
    }
    const float bm = block_max(m, red);
    int cand = m == bm ? mi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ired[threadIdx.x >> 6] = cand;
    __syncthreads();
    const int best = min(min(ired[0], ired[1]), min(ired[2], ired[3]));
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += WG) s += expf(x[v] - bm);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float lse = bm + logf(s);
        lse_out[row] = lse;
        if (argmax != nullptr) argmax[row] = best == 0x7fffffff ? 0 : best;      // all-NaN row: keep the index in range
        const float wm = w[row];
        row_loss[row] = wm != 0.f ? wm * (lse - x[target[row]]) : 0.f;
    }
}

__global__ __launch_bounds__(WG) void row_sum_kernel(const float* row_loss, int M, float* loss_sum) {
    __shared__ float red[4];
    float s = 0.f;
    for (int m = threadIdx.x; m < M; m += WG) s += row_loss[m];
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss_sum[0] = s;
}

// d_logits[m, v] = g[0] * w[m] * (softmax(logits[m])[v] - [v == target[m]])
__global__ __launch_bounds__(WG) void vocab_nll_bwd_kernel(const float* logits, const float* lse, const int64_t* target,
                                                           const float* w, const float* g, int V, float* d_logits) {
    const int m = blockIdx.y;
    const int v = blockIdx.x * WG + threadIdx.x;
    if (v >= V) return;
    const float gw = g[0] * w[m];
    const size_t o = (size_t)m * V + v;
    d_logits[o] = gw == 0.f ? 0.f : gw * (expf(logits[o] - lse[m]) - (v == (int)target[m] ? 1.f : 0.f));
}

// Vocabulary head criterion folded into the GEMM's finishing pass (captioner.py:266, :313, :361 +
// misc/utils.py:132-146, 181-192): the K-slice slabs of the head's tile GEMM are summed (+ bias) into registers, the row's
// log-sum-exp, argmax and weighted NLL are taken there, and what is WRITTEN is not the logits but
//     pre[m, v] = w[m] * (softmax(logits[m])[v] - [v == target[m]])
// -- the gradient of the row's loss term with respect to the logits, which the backward scales by the upstream scalar.  The
// [B*T, V] logits never exist as a tensor in training.  `pre` may alias slab 0 (every thread reads its columns of all slabs
// before it writes).  V <= 256 * NLL_CACHE.
constexpr int NLL_CACHE = 32;
__global__ __launch_bounds__(WG) void vocab_head_nll_kernel(const float* parts, int nparts, long long part_stride, int ld, const float* bias,
                                                            const int64_t* target, const float* w, int V, float* pre, int ld_pre,
                                                            int64_t* argmax, float* row_loss) {
    __shared__ float red[4];
    __shared__ int ired[4];
    __shared__ float xt_s;
    const int row = blockIdx.x;
    const float* x = parts + (size_t)row * ld;
    const int tgt = (int)target[row];
    float vals[NLL_CACHE];
    float m = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < NLL_CACHE; ++u) {
        const int v = threadIdx.x + u * WG;
        float xv = -INFINITY;
        if (v < V) {
            xv = x[v];
            for (int p = 1; p < nparts; ++p) xv += x[(size_t)p * part_stride + v];
            if (bias != nullptr) xv += bias[v];
            if (xv > m) { m = xv; mi = v; }                   // ascending v per thread: first occurrence wins
            if (v == tgt) xt_s = xv;
        }
        vals[u] = xv;
    }
    const float bm = block_max(m, red);
    int cand = m == bm ? mi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ired[threadIdx.x >> 6] = cand;
    __syncthreads();
    const int best = min(min(ired[0], ired[1]), min(ired[2], ired[3]));
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NLL_CACHE; ++u)
        if (threadIdx.x + u * WG < V) s += expf(vals[u] - bm);
    s = block_sum(s, red);
    const float lse = bm + logf(s);
    const float wm = w[row];
    if (threadIdx.x == 0) {
        if (argmax != nullptr) argmax[row] = best == 0x7fffffff ? 0 : best;
        row_loss[row] = wm != 0.f ? wm * (lse - xt_s) : 0.f;
    }
    float* out = pre + (size_t)row * ld_pre;
#pragma unroll
    for (int u = 0; u < NLL_CACHE; ++u) {
        const int v = threadIdx.x + u * WG;
        if (v < V) out[v] = wm == 0.f ? 0.f : wm * (expf(vals[u] - lse) - (v == tgt ? 1.f : 0.f));
    }
}

// y[i] = g[0] * x[i] (the backward of the fused head criterion: d_logits = upstream scalar x pre), float4 granules
__global__ __launch_bounds__(WG) void scale_by_scalar_kernel(const float* x, const float* g, long long n4, float* y) {
    const long long i = (long long)blockIdx.x * WG + threadIdx.x;
    if (i >= n4) return;
    const float s = g[0];
    const f32x4 v = ld4(x + i * 4);
    st4(y + i * 4, f32x4{v.x * s, v.y * s, v.z * s, v.w * s});
}

// ------------------------------------------------------------------ LSTM pointwise backward
__global__ __launch_bounds__(WG) void lstm_pointwise_bwd_kernel(const float* d_h, const float* d_h2, const float* d_h3, const float* d_c,
                                                                const float* gates, const float* c_prev, const float* c_new, int M,
                                                                int R, float* d_gates, float* d_c_prev, float* d_gates_q,
                                                                DropSpec rng3) {
    const int j = blockIdx.x * WG + threadIdx.x;
    const int m = blockIdx.y;
    if (j >= R) return;
    const size_t o = (size_t)m * R + j, g0 = (size_t)m * 4 * R + j;
    const float ig = gates[g0], fg = gates[g0 + R], gg = gates[g0 + 2 * R], og = gates[g0 + 3 * R];
    const float tc = tanhf(c_new[o]);
    // h' may have gone out as up to three tensors (one per consumer): their gradients are summed here, not by autograd
    // (rng3: the third tensor went out through nn.Dropout fused into the cell's launch -- its gradient takes the same mask)
    float dh3 = d_h3 != nullptr ? d_h3[o] : 0.f;
    if (rng3.state != nullptr) dh3 *= cvc_drop_mult(rng3, rng3.state[0], rng3.state[1], rng3.state[2], (uint32_t)o);
    const float dh = ((d_h != nullptr ? d_h[o] : 0.f) + (d_h2 != nullptr ? d_h2[o] : 0.f)) + dh3;
    const float dcn = (d_c != nullptr ? d_c[o] : 0.f) + dh * og * (1.f - tc * tc);
    const float d0 = dcn * gg * ig * (1.f - ig), d1 = dcn * c_prev[o] * fg * (1.f - fg);
    const float d2 = dcn * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
    d_gates[g0] = d0;
    d_gates[g0 + R] = d1;
    d_gates[g0 + 2 * R] = d2;
    d_gates[g0 + 3 * R] = d3;
    d_c_prev[o] = dcn * fg;
    if (d_gates_q != nullptr) {                      // column g*R + j -> quad (g*R + j)/4, element j&3 (R % 4 == 0)
        const size_t q0 = ((size_t)(j >> 2) * 64 + m) * 4 + (j & 3), qs = (size_t)(R >> 2) * 256;
        d_gates_q[q0] = d0;
        d_gates_q[q0 + qs] = d1;
        d_gates_q[q0 + 2 * qs] = d2;
        d_gates_q[q0 + 3 * qs] = d3;
    }
}

// four gradients of h' (the C-driven training loops: this step's other consumers, the next step's two cells, the dropped output),
// summed in the order d_h[0], d_h[1], d_h[2], d_hd * mask.  Each d_h[i] may arrive as the K-slice planes of the backward-data
// product that produced it: summed here, in plane order, instead of by a launch of their own.
struct HSrc3 { cvc_grad_src s[3]; };
__device__ __forceinline__ float hsrc_load(const cvc_grad_src& g, int m, int j) {
    if (g.p == nullptr) return 0.f;
    const float* p = g.p + (size_t)m * g.ld + j;
    float v = p[0];
    for (int k = 1; k < g.nplanes; ++k) v += p[(size_t)k * g.plane_stride];
    return v;
}
__global__ __launch_bounds__(WG) void lstm_pointwise_bwd4_kernel(HSrc3 src, const float* d_hd, DropSpec rng, const float* d_c,
                                                                 const float* gates, const float* c_prev, const float* c_new, int M,
                                                                 int R, float* d_gates, float* d_c_prev, float* d_gates_q, float* dg_sum,
                                                                 int q_row0) {
    const int j = blockIdx.x * WG + threadIdx.x;
    const int m = blockIdx.y;
    if (j >= R) return;
    const size_t o = (size_t)m * R + j, g0 = (size_t)m * 4 * R + j;
    const float ig = gates[g0], fg = gates[g0 + R], gg = gates[g0 + 2 * R], og = gates[g0 + 3 * R];
    const float tc = tanhf(c_new[o]);
    // running sum over the steps of the loop (dg_sum [M, 4R], zero before the last step's launch): the bias gradients and the dY of
    // the step-invariant fc_feats columns, accumulated by the same thread launch after launch (stream-ordered: deterministic)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (dg_sum != nullptr) { s0 = dg_sum[g0]; s1 = dg_sum[g0 + R]; s2 = dg_sum[g0 + 2 * R]; s3 = dg_sum[g0 + 3 * R]; }
    float dhd = d_hd != nullptr ? d_hd[o] : 0.f;
    if (rng.state != nullptr) dhd *= cvc_drop_mult(rng, rng.state[0], rng.state[1], rng.state[2], (uint32_t)o);
    const float dh = ((hsrc_load(src.s[0], m, j) + hsrc_load(src.s[1], m, j)) + hsrc_load(src.s[2], m, j)) + dhd;
    const float dcn = (d_c != nullptr ? d_c[o] : 0.f) + dh * og * (1.f - tc * tc);
    const float d0 = dcn * gg * ig * (1.f - ig), d1 = dcn * c_prev[o] * fg * (1.f - fg);
    const float d2 = dcn * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
    d_gates[g0] = d0;
    d_gates[g0 + R] = d1;
    d_gates[g0 + 2 * R] = d2;
    d_gates[g0 + 3 * R] = d3;
    d_c_prev[o] = dcn * fg;
    if (dg_sum != nullptr) { dg_sum[g0] = s0 + d0; dg_sum[g0 + R] = s1 + d1; dg_sum[g0 + 2 * R] = s2 + d2; dg_sum[g0 + 3 * R] = s3 + d3; }
    if (d_gates_q != nullptr) {        // (q_row0: this launch's rows sit behind another loop's in a joint 64-row operand)
        const size_t q0 = ((size_t)(j >> 2) * 64 + q_row0 + m) * 4 + (j & 3), qs = (size_t)(R >> 2) * 256;
        d_gates_q[q0] = d0;
        d_gates_q[q0 + qs] = d1;
        d_gates_q[q0 + 2 * qs] = d2;
        d_gates_q[q0 + 3 * qs] = d3;
    }
}

// The same arithmetic, four hidden units per thread: every operand as one 16-byte load, and the K-slice planes of a gradient source
// requested four at a time before the first is added (the scalar form above walks a source's planes one dependent load after the
// other -- three sources x 5-8 planes of memory latency in an 11 us launch).  Sums in the same order: same bits.
__device__ __forceinline__ f32x4 hsrc_load4(const cvc_grad_src& g, int m, int j) {
    if (g.p == nullptr) return f32x4{0.f, 0.f, 0.f, 0.f};
    const float* p = g.p + (size_t)m * g.ld + j;
    f32x4 v = ld4(p);
    int k = 1;
    for (; k + 3 < g.nplanes; k += 4) {
        const f32x4 a = ld4(p + (size_t)k * g.plane_stride), b = ld4(p + (size_t)(k + 1) * g.plane_stride);
        const f32x4 c = ld4(p + (size_t)(k + 2) * g.plane_stride), d = ld4(p + (size_t)(k + 3) * g.plane_stride);
        v += a; v += b; v += c; v += d;
    }
    for (; k < g.nplanes; ++k) v += ld4(p + (size_t)k * g.plane_stride);
    return v;
}
__device__ __forceinline__ void lstm_pointwise_bwd4v_body(const HSrc3& src, const float* d_hd, const DropSpec& rng, const float* d_c,
                                                          const float* gates, const float* c_prev, const float* c_new, int M,
                                                          int R, float* d_gates, float* d_c_prev, float* d_gates_q, float* dg_sum,
                                                          int q_row0) {
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;      // (64-thread workgroups: 8 x M of them at R = 2048)
    const int m = blockIdx.y;
    if (j >= R || m >= M) return;
    const size_t o = (size_t)m * R + j, g0 = (size_t)m * 4 * R + j;
    const f32x4 ig = ld4(gates + g0), fg = ld4(gates + g0 + R), gg = ld4(gates + g0 + 2 * R), og = ld4(gates + g0 + 3 * R);
    const f32x4 cn = ld4(c_new + o), cp = ld4(c_prev + o);
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    if (dg_sum != nullptr) { s0 = ld4(dg_sum + g0); s1 = ld4(dg_sum + g0 + R); s2 = ld4(dg_sum + g0 + 2 * R); s3 = ld4(dg_sum + g0 + 3 * R); }
    f32x4 dhd = d_hd != nullptr ? ld4(d_hd + o) : f32x4{0, 0, 0, 0};
    const f32x4 dcin = d_c != nullptr ? ld4(d_c + o) : f32x4{0, 0, 0, 0};
    const f32x4 h0 = hsrc_load4(src.s[0], m, j), h1 = hsrc_load4(src.s[1], m, j), h2 = hsrc_load4(src.s[2], m, j);
    if (rng.state != nullptr) {
        const uint32_t r0 = rng.state[0], r1 = rng.state[1], r2 = rng.state[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) dhd[e] *= cvc_drop_mult(rng, r0, r1, r2, (uint32_t)(o + e));
    }
    f32x4 d0, d1, d2, d3, dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float tc = tanhf(cn[e]);
        const float dh = ((h0[e] + h1[e]) + h2[e]) + dhd[e];
        const float dcn = dcin[e] + dh * og[e] * (1.f - tc * tc);
        d0[e] = dcn * gg[e] * ig[e] * (1.f - ig[e]);
        d1[e] = dcn * cp[e] * fg[e] * (1.f - fg[e]);
        d2[e] = dcn * ig[e] * (1.f - gg[e] * gg[e]);
        d3[e] = dh * tc * og[e] * (1.f - og[e]);
        dcp[e] = dcn * fg[e];
    }
    st4(d_gates + g0, d0); st4(d_gates + g0 + R, d1); st4(d_gates + g0 + 2 * R, d2); st4(d_gates + g0 + 3 * R, d3);
    st4(d_c_prev + o, dcp);
    if (dg_sum != nullptr) { st4(dg_sum + g0, s0 + d0); st4(dg_sum + g0 + R, s1 + d1); st4(dg_sum + g0 + 2 * R, s2 + d2); st4(dg_sum + g0 + 3 * R, s3 + d3); }
    if (d_gates_q != nullptr) {
        float* q = d_gates_q + ((size_t)(j >> 2) * 64 + q_row0 + m) * 4;
        const size_t qs = (size_t)(R >> 2) * 256;
        st4(q, d0); st4(q + qs, d1); st4(q + 2 * qs, d2); st4(q + 3 * qs, d3);
    }
}
__global__ __launch_bounds__(WG) void lstm_pointwise_bwd4v_kernel(HSrc3 src, const float* d_hd, DropSpec rng, const float* d_c,
                                                                  const float* gates, const float* c_prev, const float* c_new, int M,
                                                                  int R, float* d_gates, float* d_c_prev, float* d_gates_q, float* dg_sum,
                                                                  int q_row0) {
    lstm_pointwise_bwd4v_body(src, d_hd, rng, d_c, gates, c_prev, c_new, M, R, d_gates, d_c_prev, d_gates_q, dg_sum, q_row0);
}
// Two independent gate-gradient launches as ONE (blockIdx.z picks the argument set): the two loops of the cyclical pass share the
// LSTM cells, and in the joint back-propagation their gate-gradient kernels of a step are two ~10 us, latency-bound launches of the
// same shape -- 80 launches per training step become 40.
struct PwOne {
    HSrc3 src; const float* d_hd; DropSpec rng; const float *d_c, *gates, *c_prev, *c_new; int M; float *d_gates, *d_c_prev, *d_gates_q, *dg_sum;
    int q_row0;
};
__global__ __launch_bounds__(WG) void lstm_pointwise_bwd4v_pair_kernel(PwOne a, PwOne b, int R) {
    // (uniform branch, not an indexed argument array: indexing kernel arguments makes hipcc copy them to scratch)
    if (blockIdx.z == 0) lstm_pointwise_bwd4v_body(a.src, a.d_hd, a.rng, a.d_c, a.gates, a.c_prev, a.c_new, a.M, R, a.d_gates, a.d_c_prev, a.d_gates_q, a.dg_sum, a.q_row0);
    else lstm_pointwise_bwd4v_body(b.src, b.d_hd, b.rng, b.d_c, b.gates, b.c_prev, b.c_new, b.M, R, b.d_gates, b.d_c_prev, b.d_gates_q, b.dg_sum, b.q_row0);
}


// ------------------------------------------------------------------ beam bookkeeping
// Candidate (k, v) scores: score[k] + logit[k,v] - lse[k]; -inf for v == unk; a finished hypothesis
// only offers (k, 0) at its carried score.  The `beam` best of a clip's beam*V candidates are among the
// per-row top-`beam`, so stage 1 (one workgroup per hypothesis row, values cached in registers) finds
// those and stage 2 (one workgroup per clip) merges beam*beam candidates.  Ties -> lowest flat (k, v).
constexpr int BEAM_MAX = 8;
constexpr int ROW_CACHE = 32;        // values per thread kept in registers: V <= 256 * 32

// logits: [rows, V], or -- nparts > 1 / bias given -- the K-slice slabs of the vocabulary GEMM [nparts][rows, V] (+ bias [V]),
// summed in slab order while the row is loaded (the tile path's logits never exist as one matrix)
__global__ __launch_bounds__(WG) void beam_rowtop_kernel(const float* logits, int nparts, long long part_stride, const float* bias,
                                                         int beam, int V, int unk, float* cand_v, int* cand_i, float* lse_out) {
    __shared__ float red[4];
    __shared__ float bestv[4];
    __shared__ int besti[4];
    __shared__ int winner;
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * V;
    float vals[ROW_CACHE];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < ROW_CACHE; ++u) {
        const int v = threadIdx.x + u * WG;
        float xv = -INFINITY;
        if (v < V) {
            xv = x[v];
            for (int p = 1; p < nparts; ++p) xv += x[(size_t)p * part_stride + v];
            if (bias != nullptr) xv += bias[v];
        }
        vals[u] = xv;
        m = fmaxf(m, vals[u]);
    }
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < ROW_CACHE; ++u) s += expf(vals[u] - m);       // exp(-inf) = 0 for padding
    s = block_sum(s, red);
    if (threadIdx.x == 0) lse_out[row] = m + logf(s);
#pragma unroll
    for (int u = 0; u < ROW_CACHE; ++u)
        if (threadIdx.x + u * WG == unk) vals[u] = -INFINITY;
    for (int sel = 0; sel < beam; ++sel) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < ROW_CACHE; ++u) {
            const int v = threadIdx.x + u * WG;
            if (v < V && better(vals[u], v, bv, bi)) { bv = vals[u]; bi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (better(bestv[w], besti[w], bv, bi)) { bv = bestv[w]; bi = besti[w]; }
            cand_v[row * BEAM_MAX + sel] = bv;
            cand_i[row * BEAM_MAX + sel] = bi;
            winner = bi;
        }
        __syncthreads();
        const int wv = winner;
#pragma unroll
        for (int u = 0; u < ROW_CACHE; ++u)
            if (threadIdx.x + u * WG == wv) vals[u] = -INFINITY;   // taken
    }
}

// Fast form for one finished logit matrix with V % 4 == 0: float4 loads all requested up front (the general kernel's inner slab
// loop and bias branch serialised its 20 loads per thread), log-sum-exp from per-wave (max, sum) pairs, and the `beam` best
// found per WAVE without a barrier (shuffles only), then merged by wave 0 -- the row's top `beam` are among the 4 x beam wave
// winners, so the result is the general kernel's (ties broken towards the lower index in both).  3 barriers instead of 19.
// NP > 0: the logits are still the NP K-slice slabs of the vocabulary GEMM (+ bias): summed on load in the finishing pass's order
// (slab 0 + slab 1 + ... + bias, cvc_tile_linear_finish), so the finished matrix is never written or read back.
template <int NG, int NP = 0>      // float4 groups per thread: V <= NG * 1024
__global__ __launch_bounds__(WG) void beam_rowtop4_kernel(const float* logits, long long part_stride, const float* bias, int beam, int V,
                                                          int unk, float* cand_v, int* cand_i, float* lse_out) {
    __shared__ float wm[4], ws[4];
    __shared__ float wv[4 * BEAM_MAX];
    __shared__ int wi[4 * BEAM_MAX];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)row * V;
    f32x4 v4[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int e = (tid + g * WG) * 4;
        if (e < V) {
            v4[g] = ld4(x + e);
            if constexpr (NP > 0) {
                f32x4 pv[NP];
#pragma unroll
                for (int p = 1; p < NP; ++p) pv[p] = ld4(x + (size_t)p * part_stride + e);
#pragma unroll
                for (int p = 1; p < NP; ++p) v4[g] += pv[p];
                if (bias != nullptr) v4[g] += ld4(bias + e);
            }
        } else v4[g] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    float m = -INFINITY;
#pragma unroll
    for (int g = 0; g < NG; ++g) m = fmaxf(fmaxf(m, fmaxf(v4[g].x, v4[g].y)), fmaxf(v4[g].z, v4[g].w));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += expf(v4[g][e] - m);          // exp(-inf) = 0 for padding
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (m == -INFINITY) s = 0.f;                                      // a wave that holds only padding (-inf - -inf = nan)
    if (lane == 0) { wm[wave] = m; ws[wave] = s; }
    // the beam best of this wave (unk never selected)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[g][e] = ((tid + g * WG) * 4 + e == unk) ? -INFINITY : v4[g][e];
    for (int sel = 0; sel < beam; ++sel) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = (tid + g * WG) * 4 + e;
                const bool take = (v < V) & better(v4[g][e], v, bv, bi);
                bv = take ? v4[g][e] : bv;
                bi = take ? v : bi;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            const bool take = better(ov, oi, bv, bi);
            bv = take ? ov : bv;
            bi = take ? oi : bi;
        }
        if (lane == 0) { wv[wave * BEAM_MAX + sel] = bv; wi[wave * BEAM_MAX + sel] = bi; }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[g][e] = ((tid + g * WG) * 4 + e == bi) ? -INFINITY : v4[g][e];   // taken
    }
    __syncthreads();
    if (wave == 0) {
        if (lane == 0) {
            const float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
            float S = 0.f;
            for (int w = 0; w < 4; ++w)
                if (wm[w] != -INFINITY) S += ws[w] * expf(wm[w] - M);
            lse_out[row] = M + logf(S);
        }
        const int cw = lane / BEAM_MAX, cs = lane % BEAM_MAX;      // lane -> (wave, rank) candidate
        float cv = -INFINITY;
        int ci = 0x7fffffff;
        if (cw < 4 && cs < beam) { cv = wv[cw * BEAM_MAX + cs]; ci = wi[cw * BEAM_MAX + cs]; }
        for (int sel = 0; sel < beam; ++sel) {
            float bv = cv;
            int bi = ci;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                const bool take = better(ov, oi, bv, bi);
                bv = take ? ov : bv;
                bi = take ? oi : bi;
            }
            if (lane == 0) { cand_v[row * BEAM_MAX + sel] = bv; cand_i[row * BEAM_MAX + sel] = bi; }
            if (ci == bi) { cv = -INFINITY; ci = 0x7fffffff; }
        }
    }
}

__global__ __launch_bounds__(64) void beam_merge_kernel(const float* cand_v, const int* cand_i, const float* lse,
                                                        const float* score_in, const uint8_t* done_in, int beam, int V,
                                                        int first_step, int64_t* parent, int64_t* word, float* score_out,
                                                        uint8_t* done_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    // lane -> candidate (k = lane / beam, r = lane % beam), beam*beam <= 64
    const int k = lane / beam, r = lane - k * beam;
    float cv = -INFINITY;
    int flat = 0x7fffffff;
    if (k < beam) {
        const int row = b * beam + k;
        float sc = score_in[row];
        if (first_step && k > 0) sc = -INFINITY;
        if (done_in[row]) {
            // frozen hypothesis: candidates (k, 0) at the carried score, every other (k, v) at -inf; keep the
            // -inf fillers at distinct low flat indices so that tie-breaking matches a full scan
            cv = r == 0 ? sc : -INFINITY;
            flat = k * V + r;
        } else {
            const float v = cand_v[row * BEAM_MAX + r];
            const int vi = cand_i[row * BEAM_MAX + r];
            cv = (v == -INFINITY || sc == -INFINITY) ? -INFINITY : sc + (v - lse[row]);
            flat = vi == 0x7fffffff ? 0x7fffffff : k * V + vi;
        }
    }
    for (int sel = 0; sel < beam; ++sel) {
        float bv = cv;
        int bi = flat;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            int kk = bi / V, vv = bi - kk * V;
            if (bi == 0x7fffffff) { kk = 0; vv = 0; }      // every candidate NaN: parent / word feed gathers, keep them in range
            parent[b * beam + sel] = kk;
            word[b * beam + sel] = vv;
            score_out[b * beam + sel] = bv;
            done_out[b * beam + sel] = (done_in[b * beam + kk] != 0 || vv == 0) ? 1 : 0;
        }
        if (flat == bi) { cv = -INFINITY; flat = 0x7fffffff; }      // taken (flat indices are unique)
    }
}

__global__ __launch_bounds__(WG) void gather_rows_kernel(const float* src, const int64_t* parent, int beam, int width,
                                                         float* dst) {
    const int r = blockIdx.y;
    const int c = (blockIdx.x * WG + threadIdx.x) * 4;
    if (c >= width) return;
    const size_t s = (size_t)((r / beam) * beam + (int)parent[r]) * width + c;
    st4(dst + (size_t)r * width + c, ld4(src + s));
}

}  // namespace

#ifdef CVC_EXPERIMENTAL
extern "C" const char* cvc_version(void) { return "cvc_hip 0.2 gfx950 +experimental"; }
#else
extern "C" const char* cvc_version(void) { return "cvc_hip 0.2 gfx950"; }
#endif

static int embed_fwd_impl(const float* table, const int64_t* idx, const float* drop, DropSpec rng, int M, int E, float* out,
                          cvc_stream_t stream) {
    if (!table || !idx || !out || M < 1 || E < 4 || (E & 3) || (long long)M * E > 0xffffffffll) return CVC_E_BADARG;
    hipLaunchKernelGGL(embed_relu_fwd_kernel, dim3((E / 4 + WG - 1) / WG, M), dim3(WG), 0, (hipStream_t)stream, table, idx,
                       drop, rng, M, E, out);
    return cvc_launch_status();
}

static int embed_bwd_impl(const float* table, const int64_t* idx, const int64_t* order, const float* drop, DropSpec rng,
                          const float* d_out, int M, int E, float* d_table, float* workspace, cvc_stream_t stream) {
    if (!table || !idx || !order || !d_out || !d_table || !workspace || M < 1 || E < 4 || (E & 3)) return CVC_E_BADARG;
    const int gx = (E / 4 + WG - 1) / WG;
    hipLaunchKernelGGL(embed_bwd_pieces_kernel, dim3(gx, (M + EMB_CHUNK - 1) / EMB_CHUNK), dim3(WG), 0, (hipStream_t)stream, idx,
                       order, drop, rng, d_out, M, E, workspace);
    hipLaunchKernelGGL(embed_bwd_runs_kernel, dim3(gx, M), dim3(WG), 0, (hipStream_t)stream, table, idx, order, workspace, M, E,
                       d_table);
    return cvc_launch_status();
}

extern "C" int cvc_embed_relu_fwd(const float* table, const int64_t* idx, const float* drop, int M, int E, float* out,
                                  cvc_stream_t stream) {
    return embed_fwd_impl(table, idx, drop, cvc_drop_spec(nullptr, 0, 0.f), M, E, out, stream);
}

extern "C" int cvc_embed_relu_bwd(const float* table, const int64_t* idx, const int64_t* order, const float* drop,
                                  const float* d_out, int M, int E, float* d_table, float* workspace, cvc_stream_t stream) {
    return embed_bwd_impl(table, idx, order, drop, cvc_drop_spec(nullptr, 0, 0.f), d_out, M, E, d_table, workspace, stream);
}

extern "C" int cvc_embed_relu_rng_fwd(const float* table, const int64_t* idx, const uint32_t* rng_state, unsigned site, float p, int M,
                                      int E, float* out, cvc_stream_t stream) {
    if (!rng_state || p < 0.f || p >= 1.f) return CVC_E_BADARG;
    return embed_fwd_impl(table, idx, nullptr, cvc_drop_spec(rng_state, site, p), M, E, out, stream);
}

extern "C" int cvc_embed_relu_rng_bwd(const float* table, const int64_t* idx, const int64_t* order, const uint32_t* rng_state,
                                      unsigned site, float p, const float* d_out, int M, int E, float* d_table, float* workspace,
                                      cvc_stream_t stream) {
    if (!rng_state || p < 0.f || p >= 1.f) return CVC_E_BADARG;
    return embed_bwd_impl(table, idx, order, nullptr, cvc_drop_spec(rng_state, site, p), d_out, M, E, d_table, workspace, stream);
}

extern "C" int cvc_dropout_rng(const float* x, long long n, const uint32_t* rng_state, unsigned site, float p, float* y,
                               cvc_stream_t stream) {
    if (!x || !y || !rng_state || n < 1 || n > 0xffffffffll || p < 0.f || p >= 1.f || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        return CVC_E_BADARG;
    hipLaunchKernelGGL(dropout_rng_kernel, dim3((unsigned)((n / 4 + WG) / WG)), dim3(WG), 0, (hipStream_t)stream, x,
                       cvc_drop_spec(rng_state, site, p), (size_t)n, y);
    return cvc_launch_status();
}

extern "C" int cvc_log_softmax_fwd(const float* logits, int M, int V, float* logp, cvc_stream_t stream) {
    if (!logits || !logp || M < 1 || V < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(log_softmax_kernel, dim3(M), dim3(WG), 0, (hipStream_t)stream, logits, V, logp);
    return cvc_launch_status();
}

extern "C" int cvc_log_softmax_bwd(const float* logp, const float* d_logp, int M, int V, float* d_logits,
                                   cvc_stream_t stream) {
    if (!logp || !d_logp || !d_logits || M < 1 || V < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(M), dim3(WG), 0, (hipStream_t)stream, logp, d_logp, V, d_logits);
    return cvc_launch_status();
}

extern "C" int cvc_nll_bwd(const int64_t* target, const float* w, const float* g, int M, int V, float* d_logp,
                           cvc_stream_t stream) {
    if (!target || !w || !g || !d_logp || M < 1 || V < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((V + WG - 1) / WG, M), dim3(WG), 0, (hipStream_t)stream, target, w, g, V, d_logp);
    return cvc_launch_status();
}

extern "C" int cvc_top2_unk(const float* logits, int M, int V, int unk_idx, int64_t* word, int word_stride,
                            float* logprob, cvc_stream_t stream) {
    if (!logits || !word || M < 1 || V < 1 || word_stride < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(top2_unk_kernel, dim3(M), dim3(WG), 0, (hipStream_t)stream, logits, V, unk_idx, word, word_stride,
                       logprob);
    return cvc_launch_status();
}

extern "C" int cvc_top2_final(const float* part, int nblocks, int M, int unk_idx, int64_t* word, int word_stride,
                              float* logprob, const float* table, int E, float* emb_out, int emb_ld,
                              cvc_stream_t stream) {
    if (!part || !word || nblocks < 1 || M < 1 || M > 64 || word_stride < 1) return CVC_E_BADARG;
    if (emb_out != nullptr && (!table || E < 4 || (E & 3) || (emb_ld & 3) || (emb_ld != 0 && emb_ld < E))) return CVC_E_BADARG;
    hipLaunchKernelGGL(top2_final_kernel, dim3(M), dim3(WG), 0, (hipStream_t)stream, part, nblocks, unk_idx, word, word_stride,
                       logprob, table, E, emb_out, emb_ld);
    return cvc_launch_status();
}

extern "C" int cvc_top2_slab(const cvc_gsk_segs* logits, const float* bias, int V, int M, int unk_idx, int64_t* word,
                             int word_stride, float* logprob, const float* table, int E, float* emb_out, int emb_ld,
                             cvc_stream_t stream) {
    if (!logits || !logits->slab || logits->nchunk < 1 || logits->U < 1 || logits->maxseg < 1 || logits->unit0 < 0 || !word || V < 2 ||
        M < 1 || M > 64 || word_stride < 1)
        return CVC_E_BADARG;
    if (emb_out != nullptr && (!table || E < 4 || (E & 3) || (emb_ld & 3) || (emb_ld != 0 && emb_ld < E))) return CVC_E_BADARG;
    const int nitem = ((V + 31) / 32) * 8;
    hipStream_t st = (hipStream_t)stream;
    if (nitem <= 5 * WG)
        hipLaunchKernelGGL(top2_slab_kernel<5>, dim3(M), dim3(WG), 0, st, *logits, bias, V, unk_idx, word, word_stride, logprob, table, E,
                           emb_out, emb_ld);
    else if (nitem <= 8 * WG)
        hipLaunchKernelGGL(top2_slab_kernel<8>, dim3(M), dim3(WG), 0, st, *logits, bias, V, unk_idx, word, word_stride, logprob, table, E,
                           emb_out, emb_ld);
    else
        return CVC_E_TOOBIG;
    return cvc_launch_status();
}

extern "C" int cvc_nll_fwd(const float* logp, const int64_t* target, const float* w, int M, int V, float* loss_sum,
                           cvc_stream_t stream) {
    if (!logp || !target || !w || !loss_sum || M < 1 || V < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(1), dim3(WG), 0, (hipStream_t)stream, logp, target, w, M, V, loss_sum);
    return cvc_launch_status();
}

extern "C" int cvc_vocab_nll_fwd(const float* logits, const int64_t* target, const float* w, int M, int V, float* lse,
                                 int64_t* argmax, float* row_loss, float* loss_sum, cvc_stream_t stream) {
    if (!logits || !target || !w || !lse || !row_loss || !loss_sum || M < 1 || V < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(vocab_nll_fwd_kernel, dim3(M), dim3(WG), 0, (hipStream_t)stream, logits, target, w, V, lse, argmax,
                       row_loss);
    hipLaunchKernelGGL(row_sum_kernel, dim3(1), dim3(WG), 0, (hipStream_t)stream, row_loss, M, loss_sum);
    return cvc_launch_status();
}

extern "C" int cvc_vocab_head_nll_fwd(const float* parts, int nparts, long long part_stride, int ld, const float* bias,
                                      const int64_t* target, const float* w, int M, int V, float* pre, int ld_pre, int64_t* argmax,
                                      float* row_loss, float* loss_sum, cvc_stream_t stream) {
    if (!parts || nparts < 1 || !target || !w || !pre || !row_loss || !loss_sum || M < 1 || V < 1 || V > WG * NLL_CACHE || ld < V ||
        ld_pre < V)
        return CVC_E_BADARG;
    hipLaunchKernelGGL(vocab_head_nll_kernel, dim3(M), dim3(WG), 0, (hipStream_t)stream, parts, nparts, part_stride, ld, bias, target, w, V,
                       pre, ld_pre, argmax, row_loss);
    hipLaunchKernelGGL(row_sum_kernel, dim3(1), dim3(WG), 0, (hipStream_t)stream, row_loss, M, loss_sum);
    return cvc_launch_status();
}

// ---- small dense helpers of the training step that used to be library launches (a radix sort of 1 280 keys is three launches; a
// bias gradient is a generic reduction): both deterministic, fixed summation order.
//
// Stable order of n int64 keys by counting: rank(i) = #{j : key[j] < key[i]} + #{j < i : key[j] == key[i]}; order[rank(i)] = i.
// A workgroup ranks 32 keys: it holds all n keys in LDS, thread (key il, segment sg) counts over an eighth of the j range (all
// lanes of a segment read the same LDS word: broadcast), the eight counts are added through LDS.
__global__ __launch_bounds__(WG) void stable_order_kernel(const int64_t* key, int n, int64_t* order) {
    extern __shared__ int64_t keys[];
    __shared__ int cnt[8][32];
    for (int j = threadIdx.x; j < n; j += WG) keys[j] = key[j];
    __syncthreads();
    const int il = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + il;
    const int per = (n + 7) / 8;
    const int j0 = sg * per, j1 = min(n, j0 + per);
    int r = 0;
    if (i < n) {
        const int64_t k = keys[i];
        // j < i: ties count; j > i: strictly smaller only.  Split at i so that the loops carry no per-element branch.
        const int m = min(max(i, j0), j1);
#pragma unroll 8
        for (int j = j0; j < m; ++j) r += keys[j] <= k ? 1 : 0;
#pragma unroll 8
        for (int j = max(m, i + 1); j < j1; ++j) r += keys[j] < k ? 1 : 0;
    }
    cnt[sg][il] = r;
    __syncthreads();
    if (sg == 0 && i < n) {
        int t = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += cnt[g][il];
        order[t] = i;
    }
}

// out[c] (and out2[c]) = sum over the S rows of x[s, c].  One column per lane (a wave reads 256 contiguous bytes of a row); a
// workgroup owns 64 columns x one chunk of rows, its 4 waves take rows w, w + 4, ... with 8 loads in flight each and are combined
// in wave order.  More than one chunk: the chunks' sums go to a [chunks, n] workspace and a second launch of the same kernel adds
// them, again in a fixed order.
__global__ __launch_bounds__(WG) void col_sum_kernel(const float* x, long long ld, int S, int rows_per_chunk, int n, float* out, long long ld_out,
                                                     float* out2) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(S, r0 + rows_per_chunk);
    float acc[8], tail = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    if (c < n) {
        const float* p = x + c;
        int r = r0 + wave;
        for (; r + 28 < r1; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += p[(size_t)(r + 4 * u) * ld];
        }
        for (; r < r1; r += 4) tail += p[(size_t)r * ld];
    }
    part[wave][lane] = (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))) + tail;
    __syncthreads();
    if (wave == 0 && c < n) {
        const float v = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        out[(size_t)blockIdx.y * ld_out + c] = v;
        if (out2 != nullptr) out2[c] = v;
    }
}

extern "C" int cvc_stable_order(const int64_t* key, int n, int64_t* order, cvc_stream_t stream) {
    if (!key || !order || n < 1 || n > 7168) return CVC_E_BADARG;      // keys + counters within the 64 KB a workgroup may take
    hipLaunchKernelGGL(stable_order_kernel, dim3((n + 31) / 32), dim3(WG), (size_t)n * 8, (hipStream_t)stream, key, n, order);
    return cvc_launch_status();
}

static int col_sum_chunks(int S) { return S <= 128 ? 1 : (S + 63) / 64 < 64 ? (S + 63) / 64 : 64; }

extern "C" long long cvc_col_sum_ws(int S, int n) {
    const int ch = col_sum_chunks(S);
    return ch > 1 ? (long long)ch * n : 0;
}

extern "C" int cvc_col_sum(const float* x, long long ld, int S, int n, float* out, float* out2, float* ws, cvc_stream_t stream) {
    if (!x || !out || S < 1 || n < 1 || ld < n) return CVC_E_BADARG;
    const int ch = col_sum_chunks(S);
    if (ch == 1) {
        hipLaunchKernelGGL(col_sum_kernel, dim3((n + 63) / 64, 1), dim3(WG), 0, (hipStream_t)stream, x, ld, S, S, n, out, 0, out2);
        return cvc_launch_status();
    }
    if (!ws) return CVC_E_BADARG;
    const int rpc = (S + ch - 1) / ch;
    hipLaunchKernelGGL(col_sum_kernel, dim3((n + 63) / 64, ch), dim3(WG), 0, (hipStream_t)stream, x, ld, S, rpc, n, ws, (long long)n, nullptr);
    hipLaunchKernelGGL(col_sum_kernel, dim3((n + 63) / 64, 1), dim3(WG), 0, (hipStream_t)stream, ws, (long long)n, ch, ch, n, out, 0, out2);
    return cvc_launch_status();
}

extern "C" int cvc_scale_by_scalar(const float* x, const float* g, long long n, float* y, cvc_stream_t stream) {
    if (!x || !g || !y || n < 4 || (n & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return CVC_E_BADARG;
    const long long n4 = n / 4;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)((n4 + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, x, g, n4, y);
    return cvc_launch_status();
}

extern "C" int cvc_vocab_nll_bwd(const float* logits, const float* lse, const int64_t* target, const float* w,
                                 const float* g, int M, int V, float* d_logits, cvc_stream_t stream) {
    if (!logits || !lse || !target || !w || !g || !d_logits || M < 1 || V < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(vocab_nll_bwd_kernel, dim3((V + WG - 1) / WG, M), dim3(WG), 0, (hipStream_t)stream, logits, lse,
                       target, w, g, V, d_logits);
    return cvc_launch_status();
}

extern "C" int cvc_lstm_pointwise_bwd(const float* d_h, const float* d_c, const float* gates, const float* c_prev,
                                      const float* c_new, int M, int R, float* d_gates, float* d_c_prev,
                                      float* d_gates_q, cvc_stream_t stream) {
    if (!gates || !c_prev || !c_new || !d_gates || !d_c_prev || M < 1 || R < 1) return CVC_E_BADARG;
    if (d_gates_q != nullptr && (M > 64 || (R & 3))) return CVC_E_BADARG;
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3((R + WG - 1) / WG, M), dim3(WG), 0, (hipStream_t)stream, d_h, nullptr, nullptr,
                       d_c, gates, c_prev, c_new, M, R, d_gates, d_c_prev, d_gates_q, cvc_drop_spec(nullptr, 0, 0.f));
    return cvc_launch_status();
}

// the same with the gradients of up to three copies of h' (cvc_packed_lstm_train_fwd's h_out, h_out2, h_out3), summed in order
extern "C" int cvc_lstm_pointwise_bwd3(const float* d_h, const float* d_h2, const float* d_h3, const float* d_c, const float* gates,
                                       const float* c_prev, const float* c_new, int M, int R, float* d_gates, float* d_c_prev,
                                       float* d_gates_q, cvc_stream_t stream) {
    if (!gates || !c_prev || !c_new || !d_gates || !d_c_prev || M < 1 || R < 1) return CVC_E_BADARG;
    if (d_gates_q != nullptr && (M > 64 || (R & 3))) return CVC_E_BADARG;
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3((R + WG - 1) / WG, M), dim3(WG), 0, (hipStream_t)stream, d_h, d_h2, d_h3, d_c,
                       gates, c_prev, c_new, M, R, d_gates, d_c_prev, d_gates_q, cvc_drop_spec(nullptr, 0, 0.f));
    return cvc_launch_status();
}

// ... where the third copy of h' went out through the dropout fused into cvc_packed_lstm_train_drop_fwd: d_h3 takes that mask
extern "C" int cvc_lstm_pointwise_bwd3_drop(const float* d_h, const float* d_h2, const float* d_h3, const uint32_t* rng_state,
                                            unsigned site, float p, const float* d_c, const float* gates, const float* c_prev,
                                            const float* c_new, int M, int R, float* d_gates, float* d_c_prev, float* d_gates_q,
                                            cvc_stream_t stream) {
    if (!gates || !c_prev || !c_new || !d_gates || !d_c_prev || M < 1 || R < 1 || !rng_state || p < 0.f || p >= 1.f) return CVC_E_BADARG;
    if (d_gates_q != nullptr && (M > 64 || (R & 3))) return CVC_E_BADARG;
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3((R + WG - 1) / WG, M), dim3(WG), 0, (hipStream_t)stream, d_h, d_h2, d_h3, d_c,
                       gates, c_prev, c_new, M, R, d_gates, d_c_prev, d_gates_q, cvc_drop_spec(rng_state, site, p));
    return cvc_launch_status();
}

extern "C" int cvc_lstm_pointwise_bwd4(const cvc_grad_src* d_h, const float* d_hd, const uint32_t* rng_state, unsigned site, float p,
                                       const float* d_c, const float* gates, const float* c_prev, const float* c_new, int M, int R,
                                       float* d_gates, float* d_c_prev, float* d_gates_q, float* dg_sum, int q_row0, cvc_stream_t stream) {
    if (!d_h || !gates || !c_prev || !c_new || !d_gates || !d_c_prev || M < 1 || R < 1 || p < 0.f || p >= 1.f) return CVC_E_BADARG;
    if (d_gates_q != nullptr && (q_row0 < 0 || q_row0 + M > 64 || (R & 3))) return CVC_E_BADARG;
    HSrc3 src;
    for (int i = 0; i < 3; ++i) {
        src.s[i] = d_h[i];
        if (src.s[i].p != nullptr && (src.s[i].nplanes < 1 || src.s[i].ld < R)) return CVC_E_BADARG;
    }
    // four hidden units per thread when every operand allows 16-byte accesses (the training loops' buffers all do)
    bool vec = (R & 3) == 0;
    auto al = [&](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    vec = vec && al(d_hd) && al(d_c) && al(gates) && al(c_prev) && al(c_new) && al(d_gates) && al(d_c_prev) && al(d_gates_q) && al(dg_sum);
    for (int i = 0; i < 3; ++i)
        if (src.s[i].p != nullptr) vec = vec && al(src.s[i].p) && (src.s[i].ld & 3) == 0 && (src.s[i].plane_stride & 3) == 0;
    if (vec)
        hipLaunchKernelGGL(lstm_pointwise_bwd4v_kernel, dim3((R / 4 + 63) / 64, M), dim3(64), 0, (hipStream_t)stream, src, d_hd,
                           cvc_drop_spec(rng_state, site, p), d_c, gates, c_prev, c_new, M, R, d_gates, d_c_prev, d_gates_q, dg_sum, q_row0);
    else
        hipLaunchKernelGGL(lstm_pointwise_bwd4_kernel, dim3((R + WG - 1) / WG, M), dim3(WG), 0, (hipStream_t)stream, src, d_hd,
                           cvc_drop_spec(rng_state, site, p), d_c, gates, c_prev, c_new, M, R, d_gates, d_c_prev, d_gates_q, dg_sum, q_row0);
    return cvc_launch_status();
}

// cvc_lstm_pointwise_bwd4 for two argument sets in one launch (vector form only: every operand 16-byte aligned, R % 4 == 0)
extern "C" int cvc_lstm_pointwise_bwd4_pair(const cvc_pw_bwd_args* x, const cvc_pw_bwd_args* y, int R, cvc_stream_t stream) {
    if (!x || !y || R < 4 || (R & 3)) return CVC_E_BADARG;
    PwOne o[2];
    const cvc_pw_bwd_args* in[2] = {x, y};
    auto al = [&](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    for (int k = 0; k < 2; ++k) {
        const cvc_pw_bwd_args& a = *in[k];
        if (!a.gates || !a.c_prev || !a.c_new || !a.d_gates || !a.d_c_prev || a.M < 1 || a.p < 0.f || a.p >= 1.f) return CVC_E_BADARG;
        if (a.d_gates_q != nullptr && (a.q_row0 < 0 || a.q_row0 + a.M > 64)) return CVC_E_BADARG;
        bool vec = al(a.d_hd) && al(a.d_c) && al(a.gates) && al(a.c_prev) && al(a.c_new) && al(a.d_gates) && al(a.d_c_prev) && al(a.d_gates_q) && al(a.dg_sum);
        for (int i = 0; i < 3; ++i) {
            o[k].src.s[i] = a.d_h[i];
            if (a.d_h[i].p != nullptr) {
                if (a.d_h[i].nplanes < 1 || a.d_h[i].ld < R) return CVC_E_BADARG;
                vec = vec && al(a.d_h[i].p) && (a.d_h[i].ld & 3) == 0 && (a.d_h[i].plane_stride & 3) == 0;
            }
        }
        if (!vec) return CVC_E_BADARG;
        o[k].d_hd = a.d_hd; o[k].rng = cvc_drop_spec(a.rng_state, a.site, a.p); o[k].d_c = a.d_c; o[k].gates = a.gates; o[k].c_prev = a.c_prev;
        o[k].c_new = a.c_new; o[k].M = a.M; o[k].d_gates = a.d_gates; o[k].d_c_prev = a.d_c_prev; o[k].d_gates_q = a.d_gates_q;
        o[k].dg_sum = a.dg_sum; o[k].q_row0 = a.q_row0;
    }
    const int Mmax = x->M > y->M ? x->M : y->M;
    hipLaunchKernelGGL(lstm_pointwise_bwd4v_pair_kernel, dim3((R / 4 + 63) / 64, Mmax, 2), dim3(64), 0, (hipStream_t)stream, o[0], o[1], R);
    return cvc_launch_status();
}

extern "C" int cvc_beam_select_parts(const float* parts, int nparts, long long part_stride, const float* bias,
                                     const float* score_in, const uint8_t* done_in, int B, int beam, int V, int unk_idx,
                                     int first_step, int64_t* parent, int64_t* word, float* score_out, uint8_t* done_out,
                                     float* workspace, cvc_stream_t stream);

extern "C" int cvc_beam_select(const float* logits, const float* score_in, const uint8_t* done_in, int B, int beam, int V,
                               int unk_idx, int first_step, int64_t* parent, int64_t* word, float* score_out,
                               uint8_t* done_out, float* workspace, cvc_stream_t stream) {
    return cvc_beam_select_parts(logits, 1, 0, nullptr, score_in, done_in, B, beam, V, unk_idx, first_step, parent, word, score_out,
                                 done_out, workspace, stream);
}

extern "C" int cvc_beam_select_parts(const float* logits, int nparts, long long part_stride, const float* bias,
                                     const float* score_in, const uint8_t* done_in, int B, int beam, int V, int unk_idx,
                                     int first_step, int64_t* parent, int64_t* word, float* score_out, uint8_t* done_out,
                                     float* workspace, cvc_stream_t stream) {
    if (nparts < 1) return CVC_E_BADARG;
    if (!logits || !score_in || !done_in || !parent || !word || !score_out || !done_out || !workspace) return CVC_E_BADARG;
    if (B < 1 || beam < 1 || beam > BEAM_MAX || V < beam + 1 || V > WG * ROW_CACHE) return CVC_E_BADARG;
    // workspace: [rows*8] candidate values, [rows*8] candidate indices, [rows] lse   (rows = B*beam)
    const int rows = B * beam;
    float* cand_v = workspace;
    int* cand_i = reinterpret_cast<int*>(workspace + (size_t)rows * BEAM_MAX);
    float* lse = workspace + (size_t)rows * BEAM_MAX * 2;
    const bool aligned = (V & 3) == 0 && ((uintptr_t)logits & 15) == 0 && 4 * BEAM_MAX <= 64;
    const bool slabs = aligned && (nparts == 2 || nparts == 4 || nparts == 6 || nparts == 8) && (part_stride & 3) == 0 &&
                       ((uintptr_t)bias & 15) == 0;
    // (a ONE-workgroup-per-clip form -- 2 * beam waves scanning the clip's rows, both merges in LDS, no second launch -- was built
    // and measured in round 6: bit-identical, but 36 us against 25 us for the two launches below at 64 clips x beam 5, V = 5000,
    // 6 slabs: 64 workgroups pulling 600 KB each ingest at ~20 GB/s per compute unit; 320 row workgroups spread the same bytes
    // over the chip)
    if ((aligned && nparts == 1 && bias == nullptr) || slabs) {
#define CVC_RT4P(NG_, NP_) hipLaunchKernelGGL((beam_rowtop4_kernel<NG_, NP_>), dim3(rows), dim3(WG), 0, (hipStream_t)stream, logits, \
                                              part_stride, bias, beam, V, unk_idx, cand_v, cand_i, lse)
#define CVC_RT4(NG_) do { switch (slabs ? nparts : 0) { case 2: CVC_RT4P(NG_, 2); break; case 4: CVC_RT4P(NG_, 4); break; \
                                                         case 6: CVC_RT4P(NG_, 6); break; case 8: CVC_RT4P(NG_, 8); break; \
                                                         default: CVC_RT4P(NG_, 0); break; } } while (0)
        switch ((V + 4 * WG - 1) / (4 * WG)) {
            case 1: CVC_RT4(1); break;
            case 2: CVC_RT4(2); break;
            case 3: CVC_RT4(3); break;
            case 4: CVC_RT4(4); break;
            case 5: CVC_RT4(5); break;
            case 6: CVC_RT4(6); break;
            case 7: CVC_RT4(7); break;
            default: CVC_RT4(8); break;
        }
#undef CVC_RT4
#undef CVC_RT4P
    } else
        hipLaunchKernelGGL(beam_rowtop_kernel, dim3(rows), dim3(WG), 0, (hipStream_t)stream, logits, nparts, part_stride, bias, beam,
                           V, unk_idx, cand_v, cand_i, lse);
    hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, cand_v, cand_i, lse, score_in, done_in,
                       beam, V, first_step, parent, word, score_out, done_out);
    return cvc_launch_status();
}

namespace {
// rank-0 hypothesis of every clip: walk the parent pointers back from the last step (one thread), then copy the words and
// the attention rows of the path (the attention of step t was computed for the PARENT row of the word chosen at step t)
__global__ __launch_bounds__(256) void beam_backtrack_kernel(const int64_t* words, const int64_t* parent, const float* att, int beam,
                                                            int T, int N, int rows, int64_t* seq, float* att_out) {
    __shared__ int path[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int k = 0;
        for (int t = T - 1; t >= 0; --t) {
            const size_t at = (size_t)t * rows + (size_t)b * beam + k;
            seq[(size_t)b * T + t] = words[at];
            int kp = (int)parent[at];
            kp = kp < 0 ? 0 : (kp >= beam ? beam - 1 : kp);
            path[t] = kp;
            k = kp;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < T * N; idx += 256) {
        const int t = idx / N, n = idx - t * N;
        att_out[((size_t)b * T + t) * N + n] = att[((size_t)t * rows + (size_t)b * beam + path[t]) * N + n];
    }
}
}  // namespace

extern "C" int cvc_beam_backtrack(const int64_t* words, const int64_t* parent, const float* att, int B, int beam, int T, int N,
                                  int64_t* seq, float* att_out, cvc_stream_t stream) {
    if (!words || !parent || !att || !seq || !att_out || B < 1 || beam < 1 || T < 1 || T > 256 || N < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, words, parent, att, beam, T, N, B * beam,
                       seq, att_out);
    return cvc_launch_status();
}

extern "C" int cvc_gather_rows(const float* src, const int64_t* parent, int rows, int beam, int width, float* dst,
                               cvc_stream_t stream) {
    if (!src || !parent || !dst || rows < 1 || beam < 1 || width < 4 || (width & 3)) return CVC_E_BADARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((width / 4 + WG - 1) / WG, rows), dim3(WG), 0, (hipStream_t)stream, src,
                       parent, beam, width, dst);
    return cvc_launch_status();
}
