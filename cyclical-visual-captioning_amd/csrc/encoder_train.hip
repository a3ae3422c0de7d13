// Training forms of the once-per-clip encoder's small pieces (model/backbone.py:55-81, 215-235, 274-277, 325-333): what
// csrc/encoder_ops.hip fuses for inference, with the backward passes and the train-mode semantics the reference's modules have --
// nn.Dropout after the ReLU of every Linear -> ReLU -> Dropout block (mask generated in the kernel, csrc/dropout_rng.h),
// BatchNorm1d on BATCH statistics (+ running-statistics update) followed by ReLU, the class-similarity softmax, the layer norms.
// Row reductions / elementwise work over at most a few hundred MB per step: HBM-bound, one or two passes each.
#include "cvc_common.h"
#include "dropout_rng.h"
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

// ---------------------------------------------------------------- Linear -> ReLU -> Dropout epilogue (backbone.py:55-79)
// y[r, c] = max(x[r, c] + bias[c], 0) * multiplier(site, r * N + c);   float4 granules (N % 4 == 0)
__global__ __launch_bounds__(WG) void relu_dropout_fwd_kernel(const float* x, const float* bias, int N, size_t n4, DropSpec rng, float* y) {
    const size_t i4 = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i4 >= n4) return;
    const size_t i = i4 * 4;
    f32x4 v = ld4(x + i);
    if (bias != nullptr) v += ld4(bias + (int)(i % (size_t)N));
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    if (rng.state != nullptr) { s0 = rng.state[0]; s1 = rng.state[1]; s2 = rng.state[2]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = fmaxf(v[e], 0.f);
        if (rng.state != nullptr) r *= cvc_drop_mult(rng, s0, s1, s2, (uint32_t)(i + e));
        v[e] = r;
    }
    st4(y + i, v);
}

// dx = dy * multiplier * [y > 0]   (y > 0 exactly where the element was kept and its pre-activation positive)
__global__ __launch_bounds__(WG) void relu_dropout_bwd_kernel(const float* dy, const float* y, size_t n4, DropSpec rng, float* dx) {
    const size_t i4 = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i4 >= n4) return;
    const size_t i = i4 * 4;
    const f32x4 g = ld4(dy + i), o = ld4(y + i);
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    if (rng.state != nullptr) { s0 = rng.state[0]; s1 = rng.state[1]; s2 = rng.state[2]; }
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float m = o[e] > 0.f ? 1.f : 0.f;
        if (rng.state != nullptr) m *= cvc_drop_mult(rng, s0, s1, s2, (uint32_t)(i + e));
        r[e] = g[e] * m;
    }
    st4(dx + i, r);
}

// ---------------------------------------------------------------- BatchNorm1d (batch statistics) + ReLU (backbone.py:81, 332)
// Column reductions over `rows` rows of [rows, C]: a workgroup = 256 columns x one chunk of rows (coalesced along the columns),
// partials [chunks][C] combined in chunk order by the finalize kernels (deterministic).
//   what = 0: sum x        1: sum (x - mean)^2        2: sum dy', sum dy' * xhat  (dy' = dy * [y > 0], xhat = (x - mean) * invstd)
constexpr int BN_CHUNK = 256;
__global__ __launch_bounds__(WG) void bn_colsum_kernel(int what, const float* x, const float* dy, const float* y, const float* mean,
                                                       const float* invstd, long long rows, int C, float* part0, float* part1) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= C) return;
    const long long r0 = (long long)blockIdx.y * BN_CHUNK, r1 = r0 + BN_CHUNK < rows ? r0 + BN_CHUNK : rows;
    float a0 = 0.f, a1 = 0.f;
    const float m = what >= 1 ? mean[c] : 0.f, is = what == 2 ? invstd[c] : 0.f;
    for (long long r = r0; r < r1; ++r) {
        const size_t o = (size_t)r * C + c;
        if (what == 0) a0 += x[o];
        else if (what == 1) { const float d = x[o] - m; a0 += d * d; }
        else {
            const float g = y[o] > 0.f ? dy[o] : 0.f;
            a0 += g;
            a1 += g * ((x[o] - m) * is);
        }
    }
    part0[(size_t)blockIdx.y * C + c] = a0;
    if (what == 2) part1[(size_t)blockIdx.y * C + c] = a1;
}

// mean[c] = sum over chunks / rows
__global__ __launch_bounds__(WG) void bn_mean_kernel(const float* part, int nchunk, int C, long long rows, float* mean) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * C + c];
    mean[c] = s / (float)rows;
}

// biased variance -> invstd; running statistics as nn.BatchNorm1d updates them (momentum, UNBIASED variance)
__global__ __launch_bounds__(WG) void bn_var_kernel(const float* part, int nchunk, int C, long long rows, float eps, float momentum,
                                                    const float* mean, float* invstd, float* running_mean, float* running_var) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * C + c];
    const float var = s / (float)rows;
    invstd[c] = rsqrtf(var + eps);
    if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    if (running_var != nullptr) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (rows > 1 ? s / (float)(rows - 1) : var);
}

__global__ __launch_bounds__(WG) void bn_apply_relu_kernel(const float* x, const float* mean, const float* invstd, const float* gamma,
                                                           const float* beta, size_t n4, int C, float* y) {
    const size_t i4 = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i4 >= n4) return;
    const size_t i = i4 * 4;
    const int c = (int)(i % (size_t)C);
    const f32x4 v = ld4(x + i), m = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf((v[e] - m[e]) * is[e] * g[e] + b[e], 0.f);
    st4(y + i, r);
}

// dgamma[c] = sum dy' xhat, dbeta[c] = sum dy' (chunk order)
__global__ __launch_bounds__(WG) void bn_bwd_sums_kernel(const float* part0, const float* part1, int nchunk, int C, float* dbeta, float* dgamma) {
    const int c = blockIdx.x * WG + threadIdx.x;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < nchunk; ++k) { s0 += part0[(size_t)k * C + c]; s1 += part1[(size_t)k * C + c]; }
    dbeta[c] = s0;
    dgamma[c] = s1;
}

// dx = gamma * invstd * (dy' - dbeta / rows - xhat * dgamma / rows)
__global__ __launch_bounds__(WG) void bn_bwd_apply_kernel(const float* x, const float* dy, const float* y, const float* mean, const float* invstd,
                                                          const float* gamma, const float* dbeta, const float* dgamma, size_t n4, int C,
                                                          float inv_rows, float* dx) {
    const size_t i4 = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i4 >= n4) return;
    const size_t i = i4 * 4;
    const int c = (int)(i % (size_t)C);
    const f32x4 v = ld4(x + i), g = ld4(dy + i), o = ld4(y + i), m = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c);
    const f32x4 db = ld4(dbeta + c), dg = ld4(dgamma + c);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gp = o[e] > 0.f ? g[e] : 0.f, xh = (v[e] - m[e]) * is[e];
        r[e] = ga[e] * is[e] * (gp - db[e] * inv_rows - xh * dg[e] * inv_rows);
    }
    st4(dx + i, r);
}

// ---------------------------------------------------------------- class-similarity softmax backward (backbone.py:222-235)
// one wave per region row: d_logits[b, n, c] = p[c] * (d[c] - sum_c p d),  d = d_rows[b, n, c] + d_sim[b, c, n]; padded regions: 0
__global__ __launch_bounds__(WG) void class_softmax_bwd_kernel(const float* p_rows, const float* d_rows, const float* d_sim, const uint8_t* pad,
                                                               int B, int N, int C, float* d_logits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= (long long)B * N) return;
    const int b = (int)(row / N), n = (int)(row - (long long)b * N);
    float* out = d_logits + (size_t)row * C;
    if (pad != nullptr && pad[row] != 0) {
        for (int c = lane; c < C; c += 64) out[c] = 0.f;
        return;
    }
    const float* p = p_rows + (size_t)row * C;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = (d_rows != nullptr ? d_rows[(size_t)row * C + c] : 0.f) + (d_sim != nullptr ? d_sim[((size_t)b * C + c) * N + n] : 0.f);
        dot += p[c] * d;
    }
    dot = wave_sum(dot);
    for (int c = lane; c < C; c += 64) {
        const float d = (d_rows != nullptr ? d_rows[(size_t)row * C + c] : 0.f) + (d_sim != nullptr ? d_sim[((size_t)b * C + c) * N + n] : 0.f);
        out[c] = p[c] * (d - dot);
    }
}

// ---------------------------------------------------------------- layer norms + concat, backward (backbone.py:215-216, 274-277)
struct LnBwdArgs {
    const float* x[3];
    float* dx[3];
    long long ldx[3], lddx[3];
    int d[3];
    int nseg;
    const float* dout;
    long long ldo;
    float eps;
};
// per row and segment: dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)); mean / rstd recomputed from x (no saved statistics)
__global__ __launch_bounds__(WG) void layernorm_cat_bwd_kernel(LnBwdArgs a) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    int off = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const int d = a.d[s];
        if (a.dx[s] != nullptr) {
            const float* x = a.x[s] + row * a.ldx[s];
            const float* dy = a.dout + row * a.ldo + off;
            float sum = 0.f;
            for (int i = threadIdx.x; i < d; i += WG) sum += x[i];
            const float mean = block_sum4(sum, red) / d;
            float var = 0.f;
            for (int i = threadIdx.x; i < d; i += WG) { const float c = x[i] - mean; var += c * c; }
            const float rstd = rsqrtf(block_sum4(var, red) / d + a.eps);
            float s1 = 0.f, s2 = 0.f;
            for (int i = threadIdx.x; i < d; i += WG) { const float g = dy[i]; s1 += g; s2 += g * ((x[i] - mean) * rstd); }
            const float m1 = block_sum4(s1, red) / d, m2 = block_sum4(s2, red) / d;
            float* o = a.dx[s] + row * a.lddx[s];
            for (int i = threadIdx.x; i < d; i += WG) o[i] = rstd * (dy[i] - m1 - (x[i] - mean) * rstd * m2);
        }
        off += d;
    }
}

}  // namespace

extern "C" int cvc_relu_dropout_fwd(const float* x, const float* bias, long long rows, int N, const uint32_t* rng_state, unsigned site,
                                    float p, float* y, cvc_stream_t stream) {
    if (!x || !y || rows < 1 || N < 4 || (N & 3) || p < 0.f || p >= 1.f || rows * N > 0xffffffffll) return CVC_E_BADARG;
    const size_t n4 = (size_t)rows * N / 4;
    hipLaunchKernelGGL(relu_dropout_fwd_kernel, dim3((unsigned)((n4 + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, x, bias, N, n4,
                       cvc_drop_spec(rng_state, site, p), y);
    return cvc_launch_status();
}

extern "C" int cvc_relu_dropout_bwd(const float* dy, const float* y, long long n, const uint32_t* rng_state, unsigned site, float p, float* dx,
                                    cvc_stream_t stream) {
    if (!dy || !y || !dx || n < 4 || (n & 3) || p < 0.f || p >= 1.f || n > 0xffffffffll) return CVC_E_BADARG;
    const size_t n4 = (size_t)n / 4;
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3((unsigned)((n4 + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, dy, y, n4,
                       cvc_drop_spec(rng_state, site, p), dx);
    return cvc_launch_status();
}

extern "C" long long cvc_bn_workspace(long long rows, int C) {
    if (rows < 1 || C < 1) return 0;
    return 2 * ((rows + BN_CHUNK - 1) / BN_CHUNK) * (long long)C;
}

extern "C" int cvc_bn_relu_train_fwd(const float* x, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, long long rows, int C, float* y, float* save_mean, float* save_invstd,
                                     float* workspace, cvc_stream_t stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || rows < 1 || C < 4 || (C & 3)) return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = (int)((rows + BN_CHUNK - 1) / BN_CHUNK);
    const dim3 gc((C + WG - 1) / WG, nchunk), g1((C + WG - 1) / WG);
    hipLaunchKernelGGL(bn_colsum_kernel, gc, dim3(WG), 0, st, 0, x, nullptr, nullptr, nullptr, nullptr, rows, C, workspace, nullptr);
    hipLaunchKernelGGL(bn_mean_kernel, g1, dim3(WG), 0, st, workspace, nchunk, C, rows, save_mean);
    hipLaunchKernelGGL(bn_colsum_kernel, gc, dim3(WG), 0, st, 1, x, nullptr, nullptr, save_mean, nullptr, rows, C, workspace, nullptr);
    hipLaunchKernelGGL(bn_var_kernel, g1, dim3(WG), 0, st, workspace, nchunk, C, rows, eps, momentum, save_mean, save_invstd, running_mean,
                       running_var);
    const size_t n4 = (size_t)rows * C / 4;
    hipLaunchKernelGGL(bn_apply_relu_kernel, dim3((unsigned)((n4 + WG - 1) / WG)), dim3(WG), 0, st, x, save_mean, save_invstd, gamma, beta, n4,
                       C, y);
    return cvc_launch_status();
}

extern "C" int cvc_bn_relu_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* save_mean,
                                     const float* save_invstd, long long rows, int C, float* dx, float* dgamma, float* dbeta,
                                     float* workspace, cvc_stream_t stream) {
    if (!x || !dy || !y || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace || rows < 1 || C < 4 || (C & 3))
        return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = (int)((rows + BN_CHUNK - 1) / BN_CHUNK);
    float* p0 = workspace;
    float* p1 = workspace + (size_t)nchunk * C;
    hipLaunchKernelGGL(bn_colsum_kernel, dim3((C + WG - 1) / WG, nchunk), dim3(WG), 0, st, 2, x, dy, y, save_mean, save_invstd, rows, C, p0, p1);
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3((C + WG - 1) / WG), dim3(WG), 0, st, p0, p1, nchunk, C, dbeta, dgamma);
    const size_t n4 = (size_t)rows * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + WG - 1) / WG)), dim3(WG), 0, st, x, dy, y, save_mean, save_invstd, gamma, dbeta,
                       dgamma, n4, C, 1.0f / (float)rows, dx);
    return cvc_launch_status();
}

extern "C" int cvc_class_softmax_bwd(const float* p_rows, const float* d_rows, const float* d_sim, const uint8_t* pad, int B, int N, int C,
                                     float* d_logits, cvc_stream_t stream) {
    if (!p_rows || (!d_rows && !d_sim) || !d_logits || B < 1 || N < 1 || C < 1) return CVC_E_BADARG;
    const long long rows = (long long)B * N;
    hipLaunchKernelGGL(class_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(WG), 0, (hipStream_t)stream, p_rows, d_rows, d_sim, pad,
                       B, N, C, d_logits);
    return cvc_launch_status();
}

extern "C" int cvc_layernorm_cat_bwd(const float* const* xs, const long long* ldx, const int* widths, int nseg, long long rows, float eps,
                                     const float* d_out, long long ld_out, float* const* dxs, const long long* lddx, cvc_stream_t stream) {
    if (!xs || !ldx || !widths || !d_out || !dxs || !lddx || nseg < 1 || nseg > 3 || rows < 1 || rows > 0x7fffffffll) return CVC_E_BADARG;
    LnBwdArgs a{};
    int tot = 0;
    for (int s = 0; s < nseg; ++s) {
        if (!xs[s] || widths[s] < 1 || ldx[s] < widths[s] || (dxs[s] && lddx[s] < widths[s])) return CVC_E_BADARG;
        a.x[s] = xs[s]; a.ldx[s] = ldx[s]; a.d[s] = widths[s]; a.dx[s] = dxs[s]; a.lddx[s] = lddx[s];
        tot += widths[s];
    }
    if (ld_out < tot) return CVC_E_BADARG;
    a.nseg = nseg; a.dout = d_out; a.ldo = ld_out; a.eps = eps;
    hipLaunchKernelGGL(layernorm_cat_bwd_kernel, dim3((unsigned)rows), dim3(WG), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}
