// Weight-streaming skinny GEMM (M <= 64 rows) on fp32 MFMA, with the nn.LSTMCell pointwise
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
    float* y;                 // linear: y [M, ldy]; lstm: h_out [M, R]
    float* c_out;
    float* gates_out;
    int ldy;
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

// MT = number of 32-row batch tiles (M <= 32*MT); NW = waves per workgroup (K split)
template <int MT, int NW, bool LSTM>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_kernel(GemmArgs a) {
    constexpr int LDM = MT * 32 + 1;
    constexpr int NRED = NW > 4 ? 4 : NW;               // tiles resident in LDS at once
    __shared__ float red[NRED * 32 * LDM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const int M = a.M, R = a.R;

    // this lane's weight row and batch rows
    int wrow;
    if (LSTM) wrow = (i >> 3) * R + blockIdx.x * 8 + (i & 7);
    else wrow = min((int)blockIdx.x * 32 + i, a.Nout - 1);
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
        // unit u -> (hidden jj in 0..7, batch row m); reads along m are conflict-free
        const int j0 = blockIdx.x * 8;
        for (int u = tid; u < 8 * MT * 32; u += NW * 64) {
            const int m = u % (MT * 32), jj = u / (MT * 32);
            if (m >= M) continue;
            const int j = j0 + jj;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NRED; ++w) v += red[(w * 32 + g * 8 + jj) * LDM + m];
                pre[g] = v + a.bias[g * R + j] + a.bias2[g * R + j];
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
        for (int u = tid; u < 32 * MT * 32; u += NW * 64) {
            const int nl = u & 31, m = u >> 5;
            const int n = n0 + nl;
            if (m >= M || n >= a.Nout) continue;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NRED; ++w) v += red[(w * 32 + nl) * LDM + m];
            if (a.bias != nullptr) v += a.bias[n];
            if (a.bias2 != nullptr) v += a.bias2[n];
            a.y[(size_t)m * a.ldy + n] = v;
        }
    }
}

int fill_args(GemmArgs& a, const cvc_gemm_seg* segs, int nsegs) {
    if (nsegs < 1 || nsegs > MAXSEG) return CVC_E_BADARG;
    a.nsegs = nsegs;
    a.prefix[0] = 0;
    for (int s = 0; s < MAXSEG; ++s) {
        if (s < nsegs) {
            const cvc_gemm_seg& g = segs[s];
            if (g.k < 4 || (g.k & 3) || (g.ldx & 3) || (g.ldw & 3) || !g.x || !g.w) return CVC_E_BADARG;
            if (((uintptr_t)g.x & 15) || ((uintptr_t)g.w & 15)) return CVC_E_BADARG;
            a.seg[s] = g;
            a.prefix[s + 1] = a.prefix[s] + (g.k + KC - 1) / KC;
        } else {
            a.seg[s] = segs[0];
            a.prefix[s + 1] = a.prefix[s];
        }
    }
    a.total_chunks = a.prefix[nsegs];
    return 0;
}

template <bool LSTM>
int launch(const GemmArgs& a, int blocks, hipStream_t st) {
    if (a.M <= 32) hipLaunchKernelGGL((skinny_gemm_kernel<1, 8, LSTM>), dim3(blocks), dim3(512), 0, st, a);
    else if (a.M <= 64) hipLaunchKernelGGL((skinny_gemm_kernel<2, 8, LSTM>), dim3(blocks), dim3(512), 0, st, a);
    else return CVC_E_TOOBIG;
    return cvc_launch_status();
}

}  // namespace

extern "C" int cvc_linear_fwd(const cvc_gemm_seg* segs, int nsegs, const float* bias, const float* bias2,
                              int M, int Nout, float* y, int ldy, cvc_stream_t stream) {
    if (M < 1 || Nout < 1 || y == nullptr || ldy < Nout) return CVC_E_BADARG;
    // M > 64: walk the batch in 64-row slabs (weights are re-streamed per slab; the decode
    // path never gets here, beam/training shapes do until the wide kernel lands)
    for (int m0 = 0; m0 < M; m0 += 64) {
        GemmArgs a{};
        cvc_gemm_seg tmp[MAXSEG];
        if (nsegs < 1 || nsegs > MAXSEG) return CVC_E_BADARG;
        for (int s = 0; s < nsegs; ++s) {
            tmp[s] = segs[s];
            if (tmp[s].idx) tmp[s].idx += m0; else tmp[s].x += (size_t)m0 * tmp[s].ldx;
        }
        int rc = fill_args(a, tmp, nsegs);
        if (rc) return rc;
        a.M = M - m0 < 64 ? M - m0 : 64; a.Nout = Nout; a.R = 0;
        a.bias = bias; a.bias2 = bias2; a.c_prev = nullptr;
        a.y = y + (size_t)m0 * ldy; a.c_out = nullptr; a.gates_out = nullptr; a.ldy = ldy;
        rc = launch<false>(a, (Nout + 31) / 32, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cvc_lstm_cell_fwd(const cvc_gemm_seg* segs, int nsegs, const float* b_ih, const float* b_hh,
                                 const float* c_prev, int M, int R, float* h_out, float* c_out,
                                 float* gates_out, cvc_stream_t stream) {
    if (M < 1 || R < 8 || (R & 7) || !b_ih || !b_hh || !c_prev || !h_out || !c_out) return CVC_E_BADARG;
    for (int m0 = 0; m0 < M; m0 += 64) {
        GemmArgs a{};
        cvc_gemm_seg tmp[MAXSEG];
        if (nsegs < 1 || nsegs > MAXSEG) return CVC_E_BADARG;
        for (int s = 0; s < nsegs; ++s) {
            tmp[s] = segs[s];
            if (tmp[s].idx) tmp[s].idx += m0; else tmp[s].x += (size_t)m0 * tmp[s].ldx;
        }
        int rc = fill_args(a, tmp, nsegs);
        if (rc) return rc;
        a.M = M - m0 < 64 ? M - m0 : 64; a.Nout = 4 * R; a.R = R;
        a.bias = b_ih; a.bias2 = b_hh; a.c_prev = c_prev + (size_t)m0 * R;
        a.y = h_out + (size_t)m0 * R; a.c_out = c_out + (size_t)m0 * R;
        a.gates_out = gates_out ? gates_out + (size_t)m0 * 4 * R : nullptr; a.ldy = R;
        rc = launch<true>(a, R / 8, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}
