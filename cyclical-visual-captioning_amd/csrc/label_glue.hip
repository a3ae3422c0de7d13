// Label glue and the supervised attention criteria of the cyclical training pass, for all T steps
// in one launch each:
//   cvc_bbox_overlaps_fwd : misc/utils.py:335-338 -> misc/bbox_transform.py:224-272 -- IoU of proposals vs ground-truth boxes
//   cvc_label_glue_fwd    : misc/utils.py:351-373 (bbox_target, labels only) + model/captioner.py:246-260 (frame mask on proposals),
//                           which the reference evaluates once per decode step
//   cvc_attn_nll_fwd/bwd  : misc/utils.py:150-162 -- att2_loss / ground_loss = -mean(log_softmax(w, 2)[att2_target])
// The first two are bool / integer results: BIT-EXACT against the oracle (the IoU is evaluated with the reference's operation
// order, every operation rounded on its own -- no fused multiply-add).
#include "cvc_common.h"
#include <math.h>

namespace {

constexpr int WG = 256;

// IoU with the +1-pixel convention, operation by operation as torch evaluates misc/bbox_transform.py:240-268.  Contraction into
// fused multiply-adds is switched OFF for this function (hipcc's default for device code is -ffp-contract=fast, and HIP's
// __fmul_rn / __fadd_rn are plain operators that contract like any other): every operation rounds on its own, as on the host.
__device__ __noinline__ float iou_ref(const float* a, const float* g, bool frame_differs) {
#pragma clang fp contract(off)
    const float gx = (g[2] - g[0]) + 1.f, gy = (g[3] - g[1]) + 1.f;
    const float ax = (a[2] - a[0]) + 1.f, ay = (a[3] - a[1]) + 1.f;
    const float g_area = gx * gy, a_area = ax * ay;
    float iw = (fminf(a[2], g[2]) - fmaxf(a[0], g[0])) + 1.f;
    float ih = (fminf(a[3], g[3]) - fmaxf(a[1], g[1])) + 1.f;
    iw = iw < 0.f ? 0.f : iw;                                  // clamp(min=0) (NaN stays NaN, as in torch)
    ih = ih < 0.f ? 0.f : ih;
    const float inter = iw * ih;
    const float den = (a_area + g_area) - inter;
    float ov = inter / den;
    ov = ov * (frame_differs ? 0.f : 1.f);                     // ov * (~frm_mask)
    if (gx == 1.f && gy == 1.f) ov = 0.f;                      // degenerate ground-truth box
    if (ax == 1.f && ay == 1.f) ov = -1.f;                     // degenerate proposal
    return ov;
}

__global__ __launch_bounds__(WG) void bbox_overlaps_kernel(const float* rois, int ld_roi, const float* gt, int ld_gt, const uint8_t* frm_mask,
                                                           const uint8_t* pnt, int ld_pnt, int N, int K, float* ov) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= N * K) return;
    const int n = i / K, k = i - n * K;
    const bool differs = frm_mask[((size_t)b * N + n) * K + k] != 0 || (pnt != nullptr && pnt[(size_t)b * ld_pnt + n] != 0);
    ov[((size_t)b * N + n) * K + k] = iou_ref(rois + ((size_t)b * N + n) * ld_roi, gt + ((size_t)b * K + k) * ld_gt, differs);
}

// one thread per (b, t, n): labels = max_k (box_mask ? 0 : ov) > 0.5;  on_prop = no k with !(box_mask | frm_mask)
__global__ __launch_bounds__(WG) void label_glue_kernel(const float* ov, const uint8_t* box_mask, long long bm_b, long long bm_k, long long bm_t,
                                                        const uint8_t* frm_mask, const uint8_t* pnt, int B, int N, int K, int T,
                                                        uint8_t* labels, uint8_t* fmo, uint8_t* step_fmask) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= T * N) return;
    const int t = i / N, n = i - t * N;
    const float* o = ov + ((size_t)b * N + n) * K;
    const uint8_t* fmk = frm_mask + ((size_t)b * N + n) * K;
    const uint8_t* bm = box_mask + (size_t)b * bm_b + (size_t)t * bm_t;
    // torch.max over K of the masked overlaps: NaN propagates (a NaN overlap makes the comparison false, as in torch)
    float m = -INFINITY;
    bool any_nan = false, all_blocked = true;
    for (int k = 0; k < K; ++k) {
        const bool masked = bm[(size_t)k * bm_k] != 0;
        const float v = masked ? 0.f : o[k];
        any_nan |= v != v;
        m = fmaxf(m, v);
        all_blocked &= masked || fmk[k] != 0;
    }
    const bool lab = !any_nan && m > 0.5f;
    labels[((size_t)b * T + t) * N + n] = lab ? 1 : 0;
    const bool f = all_blocked || pnt[(size_t)b * (N + 1) + 1 + n] != 0;
    fmo[((size_t)b * T + t) * (N + 1) + 1 + n] = f ? 1 : 0;
    if (n == 0) fmo[((size_t)b * T + t) * (N + 1)] = pnt[(size_t)b * (N + 1)] != 0 ? 1 : 0;      // sentinel column: 0 | pnt_mask[:, 0]
    if (step_fmask != nullptr) step_fmask[((size_t)t * B + b) * N + n] = f ? 1 : 0;
}

__device__ __forceinline__ float block_sum_all(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max_all(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

struct AttnNllArgs {
    const float* x[2];            // up to two [B, T, N] score tensors (att2_weights, ground_weights), element (b, t, n) at b*sb + t*st + n
    long long sb[2], st[2];
    const uint8_t* target;        // [B, T, N] contiguous
    int B, T, N, nx;
    float* row_part;              // [nx][B*T] sum_n target * log_softmax(x)
    float* row_lse;               // [nx][B*T]
    float* row_cnt;               // [B*T] sum_n target
};

// grid (B*T, nx): one workgroup per row and tensor
__global__ __launch_bounds__(WG) void attn_nll_rows_kernel(AttnNllArgs a) {
    __shared__ float red[4];
    const int row = blockIdx.x, which = blockIdx.y;
    const int b = row / a.T, t = row - b * a.T;
    const float* x = (which ? a.x[1] : a.x[0]) + (size_t)b * (which ? a.sb[1] : a.sb[0]) + (size_t)t * (which ? a.st[1] : a.st[0]);
    const uint8_t* tg = a.target + (size_t)row * a.N;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < a.N; i += WG) m = fmaxf(m, x[i]);
    m = block_max_all(m, red);
    float s = 0.f, tx = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < a.N; i += WG) {
        s += expf(x[i] - m);
        if (tg[i]) { tx += x[i]; c += 1.f; }
    }
    s = block_sum_all(s, red);
    tx = block_sum_all(tx, red);
    c = block_sum_all(c, red);
    if (threadIdx.x == 0) {
        const float lse = m + logf(s);
        a.row_lse[(size_t)which * a.B * a.T + row] = lse;
        a.row_part[(size_t)which * a.B * a.T + row] = tx - c * lse;
        if (which == 0) a.row_cnt[row] = c;
    }
}

// one workgroup: loss[w] = -sum_rows row_part[w] / max(count, 1) (fixed order), count_out = max(count, 1)
__global__ __launch_bounds__(WG) void attn_nll_final_kernel(const float* row_part, const float* row_cnt, int rows, int nx, float* loss,
                                                            float* count_out) {
    __shared__ float red[4];
    float c = 0.f;
    for (int i = threadIdx.x; i < rows; i += WG) c += row_cnt[i];
    c = block_sum_all(c, red);
    const float cnt = fmaxf(c, 1.f);
    for (int w = 0; w < nx; ++w) {
        float p = 0.f;
        for (int i = threadIdx.x; i < rows; i += WG) p += row_part[(size_t)w * rows + i];
        p = block_sum_all(p, red);
        if (threadIdx.x == 0) loss[w] = -p / cnt;
    }
    if (threadIdx.x == 0) count_out[0] = cnt;
}

// d_x[b, t, n] = g / count * (softmax(x)[n] * nt - target[n]),  nt = targets in the row; grid (B*T, nx), contiguous outputs
__global__ __launch_bounds__(WG) void attn_nll_bwd_kernel(AttnNllArgs a, const float* g0, const float* g1, const float* count, float* d0,
                                                          float* d1) {
    const int row = blockIdx.x, which = blockIdx.y;
    const float* g = which ? g1 : g0;
    float* d = which ? d1 : d0;
    if (d == nullptr) return;
    const int b = row / a.T, t = row - b * a.T;
    const float* x = (which ? a.x[1] : a.x[0]) + (size_t)b * (which ? a.sb[1] : a.sb[0]) + (size_t)t * (which ? a.st[1] : a.st[0]);
    const uint8_t* tg = a.target + (size_t)row * a.N;
    const float lse = a.row_lse[(size_t)which * a.B * a.T + row], nt = a.row_cnt[row];
    const float scale = (g != nullptr ? g[0] : 0.f) / count[0];
    for (int i = threadIdx.x; i < a.N; i += WG)
        d[(size_t)row * a.N + i] = scale * (expf(x[i] - lse) * nt - (tg[i] ? 1.f : 0.f));
}

}  // namespace

extern "C" int cvc_bbox_overlaps_fwd(const float* rois, int ld_roi, const float* gt, int ld_gt, const uint8_t* frm_mask,
                                     const uint8_t* pnt_mask, int ld_pnt, int B, int N, int K, float* ov, cvc_stream_t stream) {
    if (!rois || !gt || !frm_mask || !ov || B < 1 || N < 1 || K < 1 || ld_roi < 4 || ld_gt < 4) return CVC_E_BADARG;
    hipLaunchKernelGGL(bbox_overlaps_kernel, dim3((N * K + WG - 1) / WG, B), dim3(WG), 0, (hipStream_t)stream, rois, ld_roi, gt, ld_gt,
                       frm_mask, pnt_mask, ld_pnt, N, K, ov);
    return cvc_launch_status();
}

extern "C" int cvc_label_glue_fwd(const float* ov, const uint8_t* box_mask, long long bm_stride_b, long long bm_stride_k,
                                  long long bm_stride_t, const uint8_t* frm_mask, const uint8_t* pnt_mask, int B, int N, int K, int T,
                                  uint8_t* labels, uint8_t* frm_mask_output, uint8_t* step_fmask, cvc_stream_t stream) {
    if (!ov || !box_mask || !frm_mask || !pnt_mask || !labels || !frm_mask_output || B < 1 || N < 1 || K < 1 || T < 1) return CVC_E_BADARG;
    hipLaunchKernelGGL(label_glue_kernel, dim3((T * N + WG - 1) / WG, B), dim3(WG), 0, (hipStream_t)stream, ov, box_mask, bm_stride_b,
                       bm_stride_k, bm_stride_t, frm_mask, pnt_mask, B, N, K, T, labels, frm_mask_output, step_fmask);
    return cvc_launch_status();
}

extern "C" int cvc_attn_nll_fwd(const float* x0, long long x0_stride_b, long long x0_stride_t, const float* x1, long long x1_stride_b,
                                long long x1_stride_t, const uint8_t* target, int B, int T, int N, float* workspace, float* loss,
                                cvc_stream_t stream) {
    if (!x0 || !target || !workspace || !loss || B < 1 || T < 1 || N < 1) return CVC_E_BADARG;
    AttnNllArgs a{};
    a.x[0] = x0; a.sb[0] = x0_stride_b; a.st[0] = x0_stride_t;
    a.x[1] = x1 ? x1 : x0; a.sb[1] = x1 ? x1_stride_b : x0_stride_b; a.st[1] = x1 ? x1_stride_t : x0_stride_t;
    a.target = target; a.B = B; a.T = T; a.N = N; a.nx = x1 ? 2 : 1;
    const int rows = B * T;
    a.row_part = workspace; a.row_lse = workspace + 2 * (size_t)rows; a.row_cnt = workspace + 4 * (size_t)rows;
    hipLaunchKernelGGL(attn_nll_rows_kernel, dim3(rows, a.nx), dim3(WG), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(attn_nll_final_kernel, dim3(1), dim3(WG), 0, (hipStream_t)stream, a.row_part, a.row_cnt, rows, a.nx, loss,
                       workspace + 5 * (size_t)rows);
    return cvc_launch_status();
}

extern "C" int cvc_attn_nll_bwd(const float* x0, long long x0_stride_b, long long x0_stride_t, const float* x1, long long x1_stride_b,
                                long long x1_stride_t, const uint8_t* target, int B, int T, int N, const float* workspace,
                                const float* g0, const float* g1, float* d_x0, float* d_x1, cvc_stream_t stream) {
    if (!x0 || !target || !workspace || B < 1 || T < 1 || N < 1 || (!d_x0 && !d_x1) || (d_x1 && !x1)) return CVC_E_BADARG;
    AttnNllArgs a{};
    a.x[0] = x0; a.sb[0] = x0_stride_b; a.st[0] = x0_stride_t;
    a.x[1] = x1 ? x1 : x0; a.sb[1] = x1 ? x1_stride_b : x0_stride_b; a.st[1] = x1 ? x1_stride_t : x0_stride_t;
    a.target = target; a.B = B; a.T = T; a.N = N; a.nx = x1 ? 2 : 1;
    const int rows = B * T;
    float* ws = const_cast<float*>(workspace);
    a.row_part = ws; a.row_lse = ws + 2 * (size_t)rows; a.row_cnt = ws + 4 * (size_t)rows;
    hipLaunchKernelGGL(attn_nll_bwd_kernel, dim3(rows, a.nx), dim3(WG), 0, (hipStream_t)stream, a, g0, g1, workspace + 5 * (size_t)rows,
                       d_x0, d_x1);
    return cvc_launch_status();
}
