// Small fused kernels of the once-per-clip encoder (model/backbone.py:189-351): what stays
// between its dense products (tile GEMM, cvc/dense.py) and its GRU (cvc/gru.py) in inference.  All of it is a few MB per 64 clips
// -- row reductions and elementwise work -- so the point is to have the encoder's forward on the build's own kernels end to end,
// one launch per fused group instead of a handful of library kernels each.
//   class_softmax : class-similarity logits [B*N, C] (a tile-GEMM product) + class bias, padded regions filled with -1e8,
//                   softmax over the C classes, written transposed as [B, C, N]      (backbone.py:222-235)
//   layernorm_cat : up to three F.layer_norm(x_s, [d_s]) (no affine, eps 1e-5) written side by side into one row
//                   (backbone.py:215-216: fc | seg_info; :274-277: region | location | class-probability features)
//   frame_embed   : cat(relu(y0 + b0), relu(y1 + b1)) -> BatchNorm1d in eval mode (running statistics folded into a per-channel
//                   scale / shift on the host) -> ReLU                              (backbone.py:325-333)
#include "cvc_common.h"
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
__device__ __forceinline__ float block_max4(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one workgroup = RB consecutive regions of one clip: rows are read contiguously ([row][C]), the softmax of a row is taken by
// one wave (4 rows per pass), results go through LDS and leave as runs of RB floats along n for every class
constexpr int RB = 16;
__global__ __launch_bounds__(WG) void class_softmax_kernel(const float* logits, long long ldl, const float* bias, const uint8_t* pad,
                                                           int N, int C, float* out, float* out_rows) {
    extern __shared__ float sm[];                 // [RB][C + 1]
    const int b = blockIdx.y, n0 = blockIdx.x * RB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ld = C + 1;
    for (int r = wave; r < RB; r += 4) {
        const int n = n0 + r;
        if (n >= N) break;
        const float* x = logits + ((size_t)b * N + n) * ldl;
        const bool masked = pad != nullptr && pad[(size_t)b * N + n] != 0;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) {
            const float v = masked ? CVC_MIN_VALUE : x[c] + (bias != nullptr ? bias[c] : 0.f);
            sm[r * ld + c] = v;
            m = fmaxf(m, v);
        }
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float e = expf(sm[r * ld + c] - m);
            sm[r * ld + c] = e;
            s += e;
        }
        s = wave_sum(s);
        const float inv = 1.0f / s;
        for (int c = lane; c < C; c += 64) {
            const float p = sm[r * ld + c] * inv;
            sm[r * ld + c] = p;
            if (out_rows != nullptr) out_rows[((size_t)b * N + n) * C + c] = p;        // [B*N, C]: the layout the region features take it in
        }
    }
    __syncthreads();
    const int nr = min(RB, N - n0);
    for (int i = threadIdx.x; i < C * RB; i += WG) {
        const int c = i / RB, r = i - c * RB;
        if (r < nr) out[((size_t)b * C + c) * N + n0 + r] = sm[r * ld + c];
    }
}

struct LnArgs {
    const float* x[3];
    long long ldx[3];
    int d[3];
    int nseg;
    float* out;
    long long ldo;
    float eps;
};

// one workgroup per row; every segment normalised on its own (biased variance, as F.layer_norm)
__global__ __launch_bounds__(WG) void layernorm_cat_kernel(LnArgs a) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    int off = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const float* x = a.x[s] + row * a.ldx[s];
        const int d = a.d[s];
        float sum = 0.f;
        for (int i = threadIdx.x; i < d; i += WG) sum += x[i];
        const float mean = block_sum4(sum, red) / d;
        float var = 0.f;
        for (int i = threadIdx.x; i < d; i += WG) {
            const float c = x[i] - mean;
            var += c * c;
        }
        const float rstd = rsqrtf(block_sum4(var, red) / d + a.eps);
        float* o = a.out + row * a.ldo + off;
        for (int i = threadIdx.x; i < d; i += WG) o[i] = (x[i] - mean) * rstd;
        off += d;
    }
}

__global__ __launch_bounds__(WG) void frame_embed_kernel(const float* y0, const float* b0, int c0, const float* y1, const float* b1, int c1,
                                                         const float* scale, const float* shift, size_t rows, float* out) {
    const int C = c0 + c1;
    const size_t i = ((size_t)blockIdx.x * WG + threadIdx.x) * 4;
    if (i >= rows * C) return;
    const size_t row = i / C;
    const int c = (int)(i - row * C);
    f32x4 v = c < c0 ? ld4(y0 + row * c0 + c) + ld4(b0 + c) : ld4(y1 + row * c1 + (c - c0)) + ld4(b1 + (c - c0));
    const f32x4 sc = ld4(scale + c), sh = ld4(shift + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(v[e], 0.f) * sc[e] + sh[e], 0.f);
    st4(out + i, v);
}

}  // namespace

extern "C" int cvc_class_softmax_fwd(const float* logits, long long ld_logits, const float* bias, const uint8_t* pad, int B, int N,
                                     int C, float* out, float* out_rows, cvc_stream_t stream) {
    if (!logits || !out || B < 1 || N < 1 || C < 1 || ld_logits < C) return CVC_E_BADARG;
    const size_t lds = (size_t)RB * (C + 1) * sizeof(float);
    if (lds > 64 * 1024) return CVC_E_TOOBIG;
    hipLaunchKernelGGL(class_softmax_kernel, dim3((N + RB - 1) / RB, B), dim3(WG), lds, (hipStream_t)stream, logits, ld_logits, bias, pad, N, C,
                       out, out_rows);
    return cvc_launch_status();
}

extern "C" int cvc_layernorm_cat_fwd(const float* const* xs, const long long* ldx, const int* widths, int nseg, long long rows, float eps,
                                     float* out, long long ld_out, cvc_stream_t stream) {
    if (!xs || !ldx || !widths || !out || nseg < 1 || nseg > 3 || rows < 1) return CVC_E_BADARG;
    LnArgs a{};
    int tot = 0;
    for (int s = 0; s < nseg; ++s) {
        if (!xs[s] || widths[s] < 1 || ldx[s] < widths[s]) return CVC_E_BADARG;
        a.x[s] = xs[s]; a.ldx[s] = ldx[s]; a.d[s] = widths[s];
        tot += widths[s];
    }
    if (ld_out < tot || rows > 0x7fffffffll) return CVC_E_BADARG;
    a.nseg = nseg; a.out = out; a.ldo = ld_out; a.eps = eps;
    hipLaunchKernelGGL(layernorm_cat_kernel, dim3((unsigned)rows), dim3(WG), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}

extern "C" int cvc_frame_embed_fwd(const float* y0, const float* b0, int c0, const float* y1, const float* b1, int c1, const float* scale,
                                   const float* shift, long long rows, float* out, cvc_stream_t stream) {
    if (!y0 || !b0 || !y1 || !b1 || !scale || !shift || !out || c0 < 4 || c1 < 4 || (c0 & 3) || (c1 & 3) || rows < 1) return CVC_E_BADARG;
    const size_t n4 = (size_t)rows * (c0 + c1) / 4;
    hipLaunchKernelGGL(frame_embed_kernel, dim3((unsigned)((n4 + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, y0, b0, c0, y1, b1, c1, scale,
                       shift, (size_t)rows, out);
    return cvc_launch_status();
}
