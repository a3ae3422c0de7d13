// Backward of the GRU recurrence (autograd of nn.GRU as the encoder's frame context uses it, reference backbone.py:103-106,
// 335-338), the counterpart of cvc_gru_seq_persistent_train_fwd.  Per time step and direction, walking the sequence backwards:
//   dh      = dY_t + z_{t+1}-scaled carry + (dgh_{t+1} W_hh)           (the two carried parts arrive as separate buffers)
//   dn_pre  = dh (1 - z)(1 - n^2),  dz_pre = dh (h_{t-1} - n) z (1 - z),  dr_pre = dn_pre hn r (1 - r)      hn = W_hn h + b_hn
//   dgi_t   = (dr_pre, dz_pre, dn_pre)            -> rows of dGI (input side: dW_ih, db_ih, dX come from it in dense GEMMs)
//   dgh_t   = (dr_pre, dz_pre, dn_pre r)          -> rows of dGH (dW_hh, db_hh) and, in the quad layout, the operand of
//   dgh_t W_hh  on the backward-data kernel of the LSTM cells (cvc_linear_nn_fwd, weights in checkpoint layout).
// The dense products over all steps (dW = dG^T X on the tile GEMM) are issued by the host side (cvc/gru.py).
#include "cvc_common.h"
#include "../../include/cvc_hip.h"

namespace {

struct GruStepBwd {
    const float* dy; long long dy_ld;          // dL/dh_t rows (row m at + m * dy_ld), H columns of this direction
    const float* dh_a;                         // carried gradient dh z of the previous processed step, [M, H] row-major, nullable
    const float* dh_planes; int nplanes;       // ... and dgh W_hh of that step as K-slice planes [nplanes][M][plane_ld], nullable
    long long plane_stride, plane_ld;
    const float* gates; long long g_ld;        // (r, z, n, hn) of this step: row m at + m * g_ld, columns [4][H]
    const float* h_prev; long long hp_ld;      // h_{t-1} rows, nullable (= 0)
    float* dgi; long long dgi_ld;              // [M, 3H] at + m * dgi_ld
    float* dgh; long long dgh_ld;
    float* dgh_q;                              // [3H/4][64][4]
    float* dh_part;                            // [M, H]
    int M, H;
};

struct GruStepBwd2 { GruStepBwd d[2]; };

// both directions in one launch (blockIdx.y); the K-slice planes of the previous step's dgh W_hh are summed here, in slice order
__global__ __launch_bounds__(256) void gru_pointwise_bwd_kernel(GruStepBwd2 both) {
    const GruStepBwd& a = both.d[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = t & 63, jq = t >> 6;                         // batch row fastest: the quad-layout stores are contiguous
    const int j = jq * 4;
    if (j >= a.H) return;
    const int H = a.H;
    if (m >= a.M) {                                             // rows beyond M of the quad operand must be zero
#pragma unroll
        for (int g = 0; g < 3; ++g) st4(a.dgh_q + ((size_t)((g * H + j) >> 2) * 64 + m) * 4, f32x4{0, 0, 0, 0});
        return;
    }
    f32x4 dh = ld4(a.dy + (size_t)m * a.dy_ld + j);
    if (a.dh_a != nullptr) dh += ld4(a.dh_a + (size_t)m * H + j);
    if (a.dh_planes != nullptr) {
        const float* pl = a.dh_planes + (size_t)m * a.plane_ld + j;
        f32x4 acc = ld4(pl);
        for (int k = 1; k < a.nplanes; ++k) acc += ld4(pl + (size_t)k * a.plane_stride);
        dh += acc;
    }
    const float* gp = a.gates + (size_t)m * a.g_ld + j;
    const f32x4 r = ld4(gp), z = ld4(gp + H), n = ld4(gp + 2 * H), hn = ld4(gp + 3 * H);
    const f32x4 hp = a.h_prev != nullptr ? ld4(a.h_prev + (size_t)m * a.hp_ld + j) : f32x4{0, 0, 0, 0};
    const f32x4 dn = dh * (1.f - z) * (1.f - n * n);
    const f32x4 dz = dh * (hp - n) * z * (1.f - z);
    const f32x4 dr = dn * hn * r * (1.f - r);
    const f32x4 dnr = dn * r;
    float* gi = a.dgi + (size_t)m * a.dgi_ld + j;
    st4(gi, dr); st4(gi + H, dz); st4(gi + 2 * H, dn);
    float* gh = a.dgh + (size_t)m * a.dgh_ld + j;
    st4(gh, dr); st4(gh + H, dz); st4(gh + 2 * H, dnr);
    st4(a.dgh_q + ((size_t)((0 * H + j) >> 2) * 64 + m) * 4, dr);
    st4(a.dgh_q + ((size_t)((1 * H + j) >> 2) * 64 + m) * 4, dz);
    st4(a.dgh_q + ((size_t)((2 * H + j) >> 2) * 64 + m) * 4, dnr);
    st4(a.dh_part + (size_t)m * H + j, dh * z);
}

}  // namespace

// dy, gates, y: row of (clip m, step t) at base + m * ld_m + t * ld_t, columns [ndir][H] (dy, y) / [ndir][4][H] (gates).
// w_hh: [ndir][3H, H] row-major (the checkpoint layout).  dgi, dgh: [F * M rows (t * M + m), ndir * 3H] outputs.
// work: ndir * (2 * M * H + 3H * 64 + ksplit * M * ceil(H / 128) * 128) floats, ksplit = cvc_gru_seq_bwd_ksplit(H).
// M <= 64, H % 8 == 0.
extern "C" int cvc_gru_seq_bwd_ksplit(int H) {
    const int slabs = (H + 127) / 128;
    int ks = 256 / slabs;
    const int kmax = 3 * H / 8 / 16;       // >= 16 rows (two double groups) per wave and slice: the weights are only 12 H^2 bytes, the
                                            // launch is latency-bound and wants the whole chip (measured: 6 slices 22 us, 24 slices below)
    if (ks > kmax) ks = kmax;
    return ks < 1 ? 1 : ks;
}

extern "C" int cvc_gru_seq_bwd(const float* dy, long long dy_ld_m, long long dy_ld_t, const float* gates, long long g_ld_m,
                               long long g_ld_t, const float* y, long long y_ld_m, long long y_ld_t, const float* w_hh, int M,
                               int F, int H, int ndir, float* dgi, float* dgh, float* work, cvc_stream_t stream) {
    if (!dy || !gates || !y || !w_hh || !dgi || !dgh || !work || M < 1 || M > 64 || F < 1 || H < 8 || (H & 7) || ndir < 1 || ndir > 2)
        return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int ks = cvc_gru_seq_bwd_ksplit(H);
    const long long ntot = (long long)((H + 127) / 128) * 128;     // plane row length of the backward-data product
    const size_t mh = (size_t)M * H, per_dir = 2 * mh + (size_t)3 * H * 64 + (size_t)ks * M * ntot;
    const long long row_ld = (long long)ndir * 3 * H;            // dgi / dgh row stride
    for (int s = 0; s < F; ++s) {
        GruStepBwd2 both{};
        for (int d = 0; d < ndir; ++d) {
            const long long t = d == 0 ? F - 1 - s : s;           // the forward direction is walked back from the end
            const long long tp = d == 0 ? t - 1 : t + 1;          // where this direction's h_{t-1} lives
            float* base = work + per_dir * d;
            float* dh_part[2] = {base, base + mh};
            float* dgh_q = base + 2 * mh;
            float* planes = dgh_q + (size_t)3 * H * 64;
            GruStepBwd& a = both.d[d];
            a.dy = dy + t * dy_ld_t + (long long)d * H; a.dy_ld = dy_ld_m;
            a.dh_a = s > 0 ? dh_part[(s - 1) & 1] : nullptr;
            a.dh_planes = s > 0 ? planes : nullptr; a.nplanes = ks; a.plane_stride = (long long)M * ntot; a.plane_ld = ntot;
            a.gates = gates + t * g_ld_t + (long long)d * 4 * H; a.g_ld = g_ld_m;
            const bool has_prev = tp >= 0 && tp < F;
            a.h_prev = has_prev ? y + tp * y_ld_t + (long long)d * H : nullptr; a.hp_ld = y_ld_m;
            a.dgi = dgi + (t * M) * row_ld + (long long)d * 3 * H; a.dgi_ld = row_ld;
            a.dgh = dgh + (t * M) * row_ld + (long long)d * 3 * H; a.dgh_ld = row_ld;
            a.dgh_q = dgh_q; a.dh_part = dh_part[s & 1]; a.M = M; a.H = H;
        }
        hipLaunchKernelGGL(gru_pointwise_bwd_kernel, dim3((H / 4 * 64 + 255) / 256, ndir), dim3(256), 0, st, both);
        if (s + 1 < F) {                                          // the carry of the last processed step is not needed
            for (int d = 0; d < ndir; ++d) {
                float* base = work + per_dir * d;
                float* dgh_q = base + 2 * mh;
                float* planes = dgh_q + (size_t)3 * H * 64;
                // one K slice: the product goes straight into "plane 0" (row stride ntot); several: planes, summed by the next step
                cvc_nn_seg seg{w_hh + (size_t)d * 3 * H * H, planes, H, H, (int)ntot};
                int rc = cvc_linear_nn_planes_fwd(dgh_q, 3 * H, M, &seg, 1, ks, planes, stream);
                if (rc) return rc;
            }
        }
    }
    return cvc_launch_status();
}
