// Host-side drivers of the cyclical training pass's two recurrent loops and their back-propagation through time
// (model/captioner.py:242-270 + decoder_core.py:30-66, captioner.py:348-362 + decoder_core.py:86-113): cvc_train_loop_fwd /
// cvc_train_loop_bwd enqueue every launch of the T steps on the caller's stream from one descriptor -- what cvc/functional.py
// used to drive step by step through autograd (one Function per cell / attention / linear, ~45 launches and ~17 framework
// kernels per step), now one call per loop and direction, capturable into a HIP graph.  Nothing here touches the device except
// through the C-ABI entry points of this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cvc_hip.h"
#include "../../include/cvc_hip_blocks.h"
#include "../../include/cvc_hip_experimental.h"

#ifndef CVC_TRAIN_ATT_W_CACHED_DEFAULT
#define CVC_TRAIN_ATT_W_CACHED_DEFAULT 2
#endif

namespace {

// ---- measurement aid (bench.py): per-launch HIP-event pairs around every entry point the drivers call, on the launch stream.
// Off by default; the product path never enables it.  Kinds name the role of a launch inside a step.
enum LaunchKind { K_ZERO = 0, K_ATT_CELL, K_H2ATTN, K_SCORES, K_WSUM, K_LANG_CELL, K_PW_LANG, K_NN_LANG, K_ATTN_BWD, K_NN_H2ATTN, K_PW_ATT,
                  K_NN_ATT, K_NKINDS };
struct Prof {
    bool on = false;
    int n = 0, cap = 0;
    hipEvent_t* ev = nullptr;      // 2 per launch
    int* kind = nullptr;
    int* loop = nullptr;           // 0: loop A forward, 1: loop C forward, 2: loop A backward, 3: loop C backward
} g_prof;
int g_prof_loop = 0;

// Cache policy of the four feature tensors inside the training loops (cvc_attn_set.stream).  When they exceed the 256 MiB Infinity
// Cache (config 3: 456 MB) nothing of them survives from one step to the next and non-temporal reads leave the L2 to the small
// operands (queries, scores): score pass 39.5 -> 35.3 us, weighted sum 57.3 -> 52.5, attention backward 102 -> 87 us per step, the
// step 19.46 -> 18.94 ms.  When they fit (config 4's 32-clip share: 228 MB) cacheable reads win (12.34 against 12.42 ms).
// CVC_TRAIN_ATTN_STREAM=0 / 3 forces either (A/B).
inline int feat_stream(const cvc_train_loop& L) {
    static const int forced = getenv("CVC_TRAIN_ATTN_STREAM") ? atoi(getenv("CVC_TRAIN_ATTN_STREAM")) : -1;
    if (forced >= 0) return forced & 3;
    const unsigned long long bytes = 4ull * L.B * (L.N + L.F) * ((unsigned long long)L.A + L.R);
    return bytes > (256ull << 20) ? 3 : 0;
}

// Cache policy of the attention cell's gate matrix in the forward loops (134 MB at D = 2048: it fits the 256 MiB Infinity Cache).
// Loop C launches nothing but the two cells per step -- the language cell streams its weights non-temporally -- so the attention
// cell's matrix survives from step to step: 44.7 -> 40.3 us per launch at config 3, 33.1 -> 29.3 at config 4's share.  In loop A the
// attention passes run in between: when they read the features non-temporally (the features exceed the cache, config 3:
// feat_stream) the matrix survives them as well (44.7 -> 41.0 us); when the features are cacheable themselves (config 4's share:
// 228 MB) the two compete and everything in the loop gets slower (attention cell 33.3 -> 36.6, weighted sum 29.7 -> 33.8 us), so
// there loop A keeps streaming its weights.  Step: 17.72 -> 17.52 ms (config 3), 12.00 -> 11.91 ms (config 4's share).
// CVC_TRAIN_ATT_W_CACHED: 0 = never, 1 = loop C only, 2 = this policy (default), 3 = both loops always (A/B).
inline int att_w_cached(const cvc_train_loop& L) {
    static const int mode = getenv("CVC_TRAIN_ATT_W_CACHED") ? atoi(getenv("CVC_TRAIN_ATT_W_CACHED")) : CVC_TRAIN_ATT_W_CACHED_DEFAULT;
    const unsigned long long bytes = 4ull * 4 * L.R * 2 * L.R;
    if (bytes > (200ull << 20) || mode <= 0) return 0;
    if (L.kind == 1) return 1;
    return mode >= 3 || (mode == 2 && feat_stream(L) != 0);
}

struct ProfScope {
    hipStream_t st;
    int slot = -1;
    ProfScope(int kind, hipStream_t s) : st(s) {
        if (!g_prof.on || g_prof.n >= g_prof.cap) return;
        slot = g_prof.n++;
        g_prof.kind[slot] = kind;
        g_prof.loop[slot] = g_prof_loop;
        (void)hipEventRecord(g_prof.ev[2 * slot], st);
    }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
    }
};

#define CVC_TRY_K(kind, expr)          \
    do {                               \
        int rc_;                       \
        {                              \
            ProfScope ps_((kind), st); \
            rc_ = (expr);              \
        }                              \
        ++n;                           \
        if (rc_ != 0) return rc_;      \
    } while (0)
#define CVC_TRY(expr) CVC_TRY_K(K_ZERO, expr)

inline float* quad_off(float* buf, int k0) { return buf + (size_t)(k0 / 4) * 64 * 4; }

__global__ __launch_bounds__(256) void zero16_kernel(uint4* dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = uint4{0, 0, 0, 0};
}
// zero fill as an ordinary kernel node (see decode_driver.hip: memset nodes raced with the first GEMM inside captured graphs)
int zero(float* dst, size_t floats, hipStream_t st) {
    if (floats == 0) return 0;
    if ((floats & 3) || ((uintptr_t)dst & 15)) return CVC_E_BADARG;
    const size_t n16 = floats / 4;
    hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (uint4*)dst, n16);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int validate(const cvc_train_loop& L, bool backward) {
    if (L.kind != 0 && L.kind != 1) return CVC_E_BADARG;
    if (L.B < 1 || L.B > 64 || L.T < 1 || L.R < 32 || (L.R & 31)) return CVC_E_BADARG;
    if (!L.wp_att || !L.wp_lang || !L.gpre_att || !L.out || !L.h_att || !L.h_att_prev || !L.h_lang_prev || !L.c_att || !L.c_lang ||
        !L.g_att || !L.g_lang || !L.xa[0] || !L.xa[1] || !L.xl[0] || !L.xl[1])
        return CVC_E_BADARG;
    if ((L.row_bias != nullptr) != (L.row_index != nullptr)) return CVC_E_BADARG;
    if (L.p < 0.f || L.p >= 1.f || (L.p > 0.f && !L.rng_state)) return CVC_E_BADARG;
    if (L.kind == 0) {
        if (L.A < 4 || (L.A & 3) || L.N < 1 || L.F < 1) return CVC_E_BADARG;
        if (L.attn_kind != CVC_ATTN_ADDITIVE && L.attn_kind != CVC_ATTN_DOT) return CVC_E_BADARG;
        if (L.attn_kind == CVC_ATTN_ADDITIVE && !L.w_a) return CVC_E_BADARG;
        if (!L.w_h || !L.b_h || !L.pool || !L.ppool || !L.conv || !L.pconv || !L.ctx || !L.q || !L.attn_r || !L.attn_f || !L.scores_ws)
            return CVC_E_BADARG;
        if ((L.frame_mask != nullptr) != (L.fm != nullptr)) return CVC_E_BADARG;
        if (L.wp_h != nullptr && (L.q_split < 1 || L.q_split > 16 || (L.A & 31))) return CVC_E_BADARG;
    } else if (!L.gpre_lang) {
        return CVC_E_BADARG;
    }
    if (backward) {
        if (!L.d_out || !L.dg_att || !L.dg_lang || !L.bwd_ws || !L.w_ih_att || !L.w_hh_att || !L.w_ih_lang || !L.w_hh_lang) return CVC_E_BADARG;
        if (L.ld_ih_att < L.R || L.ld_ih_lang < 2 * L.R || (L.ld_ih_att & 3) || (L.ld_ih_lang & 3)) return CVC_E_BADARG;
        if (L.kind == 0 && (!L.dq || !L.ds_r || !L.ds_f || (L.attn_kind == CVC_ATTN_ADDITIVE && !L.dwa_part) || (L.A & 7))) return CVC_E_BADARG;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- forward
int run_fwd(const cvc_train_loop& L, hipStream_t st, int* launches) {
    const int B = L.B, T = L.T, R = L.R, A = L.A, N = L.N, F = L.F;
    const size_t BR = (size_t)B * R, BG = (size_t)B * 4 * R;
    const int k_lang = L.kind == 0 ? 3 * R : 2 * R;      // K of the language cell's per-step GEMM
    const int hl_off = L.kind == 0 ? 2 * R : R;          // h_lang(t-1)'s K offset inside XL
    const int ha_off = L.kind == 0 ? R : 0;              // h_att(t)'s K offset inside XL
    int n = 0;
    // zero state: h_lang(-1) = h_att(-1) = 0 in the operand rings and the row-major rows, c(-1) = 0
    CVC_TRY(zero(L.xa[0], (size_t)(2 * R / 4) * 256, st));
    CVC_TRY(zero(quad_off(L.xl[0], hl_off), (size_t)(R / 4) * 256, st));
    CVC_TRY(zero(L.c_att, BR, st));
    CVC_TRY(zero(L.c_lang, BR, st));
    CVC_TRY(zero(L.h_att_prev, BR, st));
    CVC_TRY(zero(L.h_lang_prev, BR, st));
    for (int t = 0; t < T; ++t) {
        const int rd = t & 1, wr = (t + 1) & 1;
        const bool last = t + 1 == T;
        // ---- attention LSTM (decoder_core.py:45-50 / :99-104): recurrent columns streamed, fc + word terms hoisted
        cvc_lstm_step a{};
        a.wp = L.wp_att; a.xq = L.xa[rd]; a.K = 2 * R; a.M = B; a.R = R;
        a.b_ih = L.b_ih_att; a.b_hh = L.b_hh_att;
        a.gate_pre = L.gpre_att + (size_t)t * BG;
        a.row_bias = L.row_bias; a.row_index = L.row_index;
        a.c_prev = L.c_att + (size_t)t * BR; a.c_out = L.c_att + (size_t)(t + 1) * BR; a.gates_out = L.g_att + (size_t)t * BG;
        a.h_out = L.h_att + (size_t)t * BR;
        a.h_out2 = last ? nullptr : L.h_att_prev + (size_t)(t + 1) * BR;
        a.h_dst1_q = quad_off(L.xl[rd], ha_off);
        a.h_dst2_q = quad_off(L.xa[wr], R);
        a.w_cached = att_w_cached(L);
        CVC_TRY_K(K_ATT_CELL, cvc_packed_lstm_step_fwd(&a, st));
        if (L.kind == 0) {
            // ---- additive / dot attention over regions + frames with one query (decoder_core.py:54-56, modules.py:100-159)
            const int qs = L.wp_h ? L.q_split : 1;
            float* q = L.q + (size_t)t * qs * B * A;
            if (L.wp_h) {       // split-K over the chip on the packed kernel, planes summed (+ bias) while the score pass loads the query
                CVC_TRY_K(K_H2ATTN, cvc_packed_linear_fwd(L.wp_h, quad_off(L.xl[rd], ha_off), R, nullptr, B, A, qs, q, A, nullptr, st));
            } else {
                cvc_gemm_seg seg{L.h_att + (size_t)t * BR, nullptr, L.w_h, R, R, R, 0};
                CVC_TRY_K(K_H2ATTN, cvc_linear_fwd(&seg, 1, L.b_h, nullptr, B, A, q, A, st));
            }
            cvc_attn_set sets[2];
            sets[0] = cvc_attn_set{L.ppool, L.pool, L.mask, L.frame_mask ? L.frame_mask + (size_t)t * B * N : nullptr, L.scores_ws,
                                   L.fm ? L.fm + (size_t)t * B * N : nullptr, L.attn_r + (size_t)t * B * N, nullptr, N, feat_stream(L)};
            sets[1] = cvc_attn_set{L.pconv, L.conv, nullptr, nullptr, L.scores_ws + (size_t)B * N, nullptr,
                                   L.attn_f + (size_t)t * B * F, nullptr, F, feat_stream(L)};
            if (L.wp_h) CVC_TRY_K(K_SCORES, cvc_attn_scores_qparts(L.attn_kind, q, qs, L.b_h, L.w_a, L.b_a, L.inv_temp, sets, 2, B, 1, A, st));
            else CVC_TRY_K(K_SCORES, cvc_attn_scores(L.attn_kind, q, L.w_a, L.b_a, L.inv_temp, sets, 2, B, 1, A, st));
            CVC_TRY_K(K_WSUM, cvc_attn_wsum_quad_rm(sets, 2, B, R, L.xl[rd], L.ctx + (size_t)t * BR, st));
        }
        // ---- language LSTM (decoder_core.py:59-62 / :106-109) + output dropout
        cvc_lstm_step l{};
        l.wp = L.wp_lang; l.xq = L.xl[rd]; l.K = k_lang; l.M = B; l.R = R;
        l.b_ih = L.b_ih_lang; l.b_hh = L.b_hh_lang;
        l.gate_pre = L.kind == 1 ? L.gpre_lang + (size_t)t * BG : nullptr;
        l.c_prev = L.c_lang + (size_t)t * BR; l.c_out = L.c_lang + (size_t)(t + 1) * BR; l.gates_out = L.g_lang + (size_t)t * BG;
        l.h_out = last ? nullptr : L.h_lang_prev + (size_t)(t + 1) * BR;
        l.h_drop_out = L.out + (size_t)t * BR;
        l.rng_state = L.p > 0.f ? L.rng_state : nullptr; l.site = L.site0 + (unsigned)t; l.p = L.p;
        l.h_dst1_q = last ? nullptr : L.xa[wr];
        l.h_dst2_q = last ? nullptr : quad_off(L.xl[wr], hl_off);
        CVC_TRY_K(K_LANG_CELL, cvc_packed_lstm_step_fwd(&l, st));
    }
    if (launches) *launches = n;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- backward
struct BwdWs {
    float *dgq, *dgq2, *dqq, *d_ctx, *d_ha_a, *d_ha_b, *d_ha_prev, *d_hl_a, *d_hl_b, *d_c_att, *d_c_lang;
    float *nn_l, *nn_h, *nn_a;       // K-slice planes of the three backward-data products of a step (each lives until its consumers ran)
    long long total;
};
BwdWs carve(float* base, int B, int R, int A) {
    BwdWs w{};
    size_t o = 0;
    auto take = [&](size_t floats) { float* p = base ? base + o : nullptr; o += (floats + 63) / 64 * 64; return p; };
    // B > 64: the 128-row joint form (two 64-row operand groups, loop C's rows from row 64 on): a second quad operand, 128-row
    // gradient rows and planes
    const size_t rows = B > 64 ? 128 : 64;
    w.dgq = take((size_t)R * 256);                        // d_gates, quad layout [4R/4][64][4]
    w.dgq2 = rows > 64 ? take((size_t)R * 256) : w.dgq;   // ... of the second group
    w.dqq = take((size_t)(A > 0 ? A : 4) / 4 * 256);      // d_q, quad layout [A/4][64][4]
    const size_t br = rows * R;
    w.d_ctx = take(br); w.d_ha_a = take(br); w.d_ha_b = take(br); w.d_ha_prev = take(br);
    w.d_hl_a = take(br); w.d_hl_b = take(br); w.d_c_att = take(br); w.d_c_lang = take(br);
    // planes of cvc_linear_nn_planes_fwd: ksplit * M * ntot floats with ksplit * slabs <= max(256, slabs)
    const size_t slabs_max = (size_t)(3 * R + 127) / 128;
    const size_t plane = rows * 128 * (slabs_max > 512 ? slabs_max : 512);
    w.nn_l = take(plane); w.nn_h = take(plane); w.nn_a = take(plane);
    w.total = (long long)o;
    return w;
}

// Backward-data product with the K split cvc/hip.py::linear_nn chooses (one resident round of workgroups, >= 16 K rows per wave),
// WITHOUT the summing launch: out[s] describes where segment s's gradient lives -- the K-slice planes in `ws` (summed by the
// consumer) or, unsplit, the segment's own dst.
int nn(const float* dy_q, int K, int M, const cvc_nn_seg* segs, int nsegs, float* ws, cvc_grad_src* out, bool finished, hipStream_t st,
       const float* dy_q2 = nullptr, int M2 = 0) {
    int slabs = 0;
    for (int s = 0; s < nsegs; ++s) slabs += (segs[s].ncols + 127) / 128;
    const int resident = cvc_gemm_packed_split(-1) != 0 ? 256 : 512;
    int ks = resident / (slabs > 0 ? slabs : 1);
    if (ks > K / 8 / 16) ks = K / 8 / 16;
    if (ks < 1) ks = 1;
    const long long ntot = (long long)slabs * 128;
    int slab = 0;
    const int mrows = dy_q2 ? 128 : M;                  // rows of a plane (128-row form: the second group's rows start at row 64)
    for (int s = 0; s < nsegs; ++s) {
        if (ks == 1 || finished) out[s] = cvc_grad_src{segs[s].dst, segs[s].ld_dst, 0, 1};
        else out[s] = cvc_grad_src{ws + (size_t)slab * 128, ntot, (long long)mrows * ntot, ks};
        slab += (segs[s].ncols + 127) / 128;
    }
    if (dy_q2) return cvc_linear_nn_planes2_fwd(dy_q, dy_q2, K, M, M2, segs, nsegs, ks, ws, finished ? 1 : 0, st);
    return finished ? cvc_linear_nn_fwd(dy_q, K, M, segs, nsegs, ks, ws, st) : cvc_linear_nn_planes_fwd(dy_q, K, M, segs, nsegs, ks, ws, st);
}

inline cvc_grad_src rows_from(cvc_grad_src g, int row0) {
    if (g.p != nullptr) g.p += (size_t)row0 * g.ld;
    return g;
}

// the gate-gradient kernels of a step: both loops' in ONE launch when both are there (cvc_lstm_pointwise_bwd4_pair; 80 launches of
// ~10 us per training step become 40), else (or when an operand is not 16-byte aligned) one launch per loop
int gate_grads(const cvc_train_loop* const* loops, const cvc_pw_bwd_args* pw, int R, int kind, hipStream_t st, int* count) {
    int n = 0;
    if (loops[0] && loops[1]) {
        g_prof_loop = 4;
        int rc;
        {
            ProfScope ps_(kind, st);
            rc = cvc_lstm_pointwise_bwd4_pair(&pw[0], &pw[1], R, st);
        }
        if (rc == 0) { *count += 1; return 0; }
        if (rc != CVC_E_BADARG) return rc;
    }
    for (int s = 0; s < 2; ++s) {
        if (!loops[s]) continue;
        g_prof_loop = 2 + loops[s]->kind;
        const cvc_pw_bwd_args& a = pw[s];
        CVC_TRY_K(kind, cvc_lstm_pointwise_bwd4(a.d_h, a.d_hd, a.rng_state, a.site, a.p, a.d_c, a.gates, a.c_prev, a.c_new, a.M, R, a.d_gates,
                                                a.d_c_prev, a.d_gates_q, a.dg_sum, a.q_row0, st));
    }
    *count += n;
    return 0;
}

// Back-propagation through time of one loop, or of both loops at once (LA = loop A or null, LC = loop C or null): the loops share the
// LSTM cells, so with both present every backward-data product takes the two loops' gate gradients in one launch and streams its
// weights once -- as ONE 64-row operand (loop A's rows first) when B_A + B_C <= 64, as two 64-row operand groups on the 128-row
// form of the product otherwise (config 3: 64 + 64 rows; 3 products per step instead of 5).  The attention backward and the h2attn product belong to loop A alone.
int run_bwd(const cvc_train_loop* LA, const cvc_train_loop* LC, hipStream_t st, int* launches) {
    const cvc_train_loop& L0 = LA ? *LA : *LC;           // shared: T, R, weights, workspace
    const int T = L0.T, R = L0.R;
    const int BA = LA ? LA->B : 0, BC = LC ? LC->B : 0;
    // both loops, more rows than ONE 64-row operand holds: two operand groups against one stream of the weights (the 128-row
    // form of the backward-data product, gemm_nn.hip) -- loop A's rows are rows 0 .., loop C's rows 64 .. of every gradient
    const bool two = LA && LC && BA + BC > 64;
    const int M = two ? BA : BA + BC;
    const int A = LA ? LA->A : 0, N = LA ? LA->N : 0, F = LA ? LA->F : 0;
    const BwdWs w = carve(L0.bwd_ws, two ? 128 : 64, R, A);
    const bool feat_grads = LA != nullptr && (LA->d_pool || LA->d_conv);     // the context-feature gradient reads d_ctx as a finished tensor
    // ... of every step, kept in LA->d_ctx_all and turned into d_pool / d_conv by ONE pass after the loop (each feature row read and
    // written once) -- per step it is a read-modify-write of both feature tensors: 2 x 304 MB x T at config 3
    const bool feat_batched = feat_grads && LA->d_ctx_all != nullptr && T <= 32;
    // the projected-feature gradients likewise (additive attention): everything their one pass reads -- the queries q_t, the score
    // gradients d_s_t, the projected features -- is kept by the loops anyway
    const bool dproj_batched = LA != nullptr && (LA->d_ppool || LA->d_pconv) && LA->attn_kind == CVC_ATTN_ADDITIVE && T <= 32;
    const cvc_grad_src none{nullptr, 0, 0, 0};
    cvc_grad_src g_hl_a = none, g_hl_b = none, g_ha_prev = none;       // what step t + 1 left for step t (all M rows)
    const cvc_train_loop* loops[2] = {LA, LC};
    const int row0[2] = {0, two ? 64 : BA};
    const int qrow0[2] = {0, two ? 0 : BA};              // row inside the loop's quad operand
    float* const dgq[2] = {w.dgq, two ? w.dgq2 : w.dgq};
    int n = 0;
    for (int s = 0; s < 2; ++s) {
        if (!loops[s]) continue;
        g_prof_loop = 2 + loops[s]->kind;
        const size_t BG = (size_t)loops[s]->B * 4 * R;
        if (loops[s]->dgsum_att) CVC_TRY(zero(loops[s]->dgsum_att, BG, st));
        if (loops[s]->dgsum_lang) CVC_TRY(zero(loops[s]->dgsum_lang, BG, st));
    }
    for (int t = T - 1; t >= 0; --t) {
        const bool last = t + 1 == T;
        // ---- language cells: d_h = dropout'(d_out[t]) + the next step's two uses of h_lang(t)
        {
            cvc_pw_bwd_args pw[2];
            for (int s = 0; s < 2; ++s) {
                if (!loops[s]) continue;
                const cvc_train_loop& L = *loops[s];
                const size_t BR = (size_t)L.B * R, BG = (size_t)L.B * 4 * R;
                float* d_c = w.d_c_lang + (size_t)row0[s] * R;
                pw[s] = cvc_pw_bwd_args{{rows_from(g_hl_a, row0[s]), rows_from(g_hl_b, row0[s]), none}, L.d_out + (size_t)t * BR,
                                        L.p > 0.f ? L.rng_state : nullptr, L.site0 + (unsigned)t, L.p, last ? nullptr : d_c,
                                        L.g_lang + (size_t)t * BG, L.c_lang + (size_t)t * BR, L.c_lang + (size_t)(t + 1) * BR, L.B,
                                        L.dg_lang + (size_t)t * BG, d_c, dgq[s], L.dgsum_lang, qrow0[s]};
            }
            if (int rc_ = gate_grads(loops, pw, R, K_PW_LANG, st, &n)) return rc_;
        }
        const int shared = LA && LC ? 4 : 2 + L0.kind;        // profile label of the products both loops share
        g_prof_loop = shared;
        cvc_grad_src g_ctx = none, g_ha_a = none, g_ha_b = none;
        {
            cvc_nn_seg segs[3];
            cvc_grad_src out[3];
            int ns = 0, i_ctx = -1, i_ha, i_hl = -1;
            float* d_ctx_t = feat_batched ? LA->d_ctx_all + (size_t)t * 128 * R : w.d_ctx;
            if (LA) { i_ctx = ns; segs[ns++] = cvc_nn_seg{L0.w_ih_lang, d_ctx_t, L0.ld_ih_lang, R, R}; }
            i_ha = ns; segs[ns++] = cvc_nn_seg{L0.w_ih_lang + R, w.d_ha_a, L0.ld_ih_lang, R, R};
            if (t > 0) { i_hl = ns; segs[ns++] = cvc_nn_seg{L0.w_hh_lang, w.d_hl_a, R, R, R}; }
            CVC_TRY_K(K_NN_LANG, nn(w.dgq, 4 * R, M, segs, ns, w.nn_l, out, feat_grads, st, two ? w.dgq2 : nullptr, BC));
            if (i_ctx >= 0) g_ctx = out[i_ctx];
            g_ha_a = out[i_ha];
            g_hl_a = i_hl >= 0 ? out[i_hl] : none;
        }
        if (LA) {
            // ---- attention (both feature sets) and h2attn: loop A's rows (the first BA of every operand)
            const cvc_train_loop& L = *LA;
            g_prof_loop = 2;
            const int B = BA;
            cvc_attn_set sets[2]{};
            sets[0].proj = L.ppool; sets[0].ctx = L.pool; sets[0].attn = L.attn_r + (size_t)t * B * N; sets[0].n = N;
            sets[0].scores = L.ds_r + (size_t)t * B * N;
            sets[0].frame_masked = L.d_fm ? const_cast<float*>(L.d_fm) + (size_t)t * B * N : nullptr;
            sets[1].proj = L.pconv; sets[1].ctx = L.conv; sets[1].attn = L.attn_f + (size_t)t * B * F; sets[1].n = F;
            sets[1].scores = L.ds_f + (size_t)t * B * F;
            sets[0].stream = sets[1].stream = feat_stream(L);
            float* d_proj[2] = {L.d_ppool, L.d_pconv};
            float* d_cf[2] = {L.d_pool, L.d_conv};
            const bool any_dp = L.d_ppool || L.d_pconv;
            const int qs = L.wp_h ? L.q_split : 1;
            const cvc_grad_src qsrc{L.q + (size_t)t * qs * B * A, A, (long long)B * A, qs};
            CVC_TRY_K(K_ATTN_BWD, cvc_attn_bwd_pair(L.attn_kind, &qsrc, L.wp_h ? L.b_h : nullptr, L.w_a, L.inv_temp, sets, 2, &g_ctx, B, 1, A, R,
                                      L.dq + (size_t)t * B * A, w.dqq, L.dwa_part ? L.dwa_part + (size_t)t * B * A : nullptr,
                                      (any_dp && !dproj_batched) ? d_proj : nullptr, (feat_grads && !feat_batched) ? d_cf : nullptr, st));
            cvc_nn_seg seg{L.w_h, w.d_ha_b, R, R, R};
            CVC_TRY_K(K_NN_H2ATTN, nn(w.dqq, A, B, &seg, 1, w.nn_h, &g_ha_b, false, st));
        }
        // ---- attention cells: d_h = language cell's input + attention query (loop A) + next step's recurrence
        {
            cvc_pw_bwd_args pw[2];
            for (int s = 0; s < 2; ++s) {
                if (!loops[s]) continue;
                const cvc_train_loop& L = *loops[s];
                const size_t BR = (size_t)L.B * R, BG = (size_t)L.B * 4 * R;
                float* d_c = w.d_c_att + (size_t)row0[s] * R;
                pw[s] = cvc_pw_bwd_args{{rows_from(g_ha_a, row0[s]), L.kind == 0 ? g_ha_b : none, rows_from(g_ha_prev, row0[s])}, nullptr,
                                        nullptr, 0, 0.f, last ? nullptr : d_c, L.g_att + (size_t)t * BG, L.c_att + (size_t)t * BR,
                                        L.c_att + (size_t)(t + 1) * BR, L.B, L.dg_att + (size_t)t * BG, d_c, dgq[s], L.dgsum_att, qrow0[s]};
            }
            if (int rc_ = gate_grads(loops, pw, R, K_PW_ATT, st, &n)) return rc_;
        }
        g_prof_loop = shared;
        if (t > 0) {
            cvc_nn_seg segs[2] = {cvc_nn_seg{L0.w_ih_att, w.d_hl_b, L0.ld_ih_att, R, R}, cvc_nn_seg{L0.w_hh_att, w.d_ha_prev, R, R, R}};
            cvc_grad_src out[2];
            CVC_TRY_K(K_NN_ATT, nn(w.dgq, 4 * R, M, segs, 2, w.nn_a, out, false, st, two ? w.dgq2 : nullptr, BC));
            g_hl_b = out[0];
            g_ha_prev = out[1];
        }
    }
    if (dproj_batched) {
        const cvc_train_loop& L = *LA;
        g_prof_loop = 2;
        const int qs = L.wp_h ? L.q_split : 1;
        const long long q_step = (long long)qs * BA * A, q_plane = (long long)BA * A;
        const float* qb = L.wp_h ? L.b_h : nullptr;
        if (L.d_ppool) CVC_TRY_K(K_ATTN_BWD, cvc_dproj_bwd_steps(L.q, q_step, q_plane, qs, qb, L.w_a, L.ppool, L.ds_r, T, BA, N, A, L.d_ppool, st));
        if (L.d_pconv) CVC_TRY_K(K_ATTN_BWD, cvc_dproj_bwd_steps(L.q, q_step, q_plane, qs, qb, L.w_a, L.pconv, L.ds_f, T, BA, F, A, L.d_pconv, st));
    }
    if (feat_batched) {
        g_prof_loop = 2;
        if (LA->d_pool) CVC_TRY_K(K_ATTN_BWD, cvc_ctxfeat_bwd_steps(LA->attn_r, LA->d_ctx_all, T, BA, N, R, LA->d_pool, st));
        if (LA->d_conv) CVC_TRY_K(K_ATTN_BWD, cvc_ctxfeat_bwd_steps(LA->attn_f, LA->d_ctx_all, T, BA, F, R, LA->d_conv, st));
    }
    if (launches) *launches = n;
    return 0;
}

}  // namespace

extern "C" long long cvc_train_loop_bwd_ws(int B, int R, int A) {
    if (B < 1 || B > 128 || R < 32 || A < 0) return 0;      /* B > 64: the joint pass of two loops with B_A + B_C rows */
    return carve(nullptr, B, R, A).total;
}

extern "C" int cvc_train_loop_fwd(const cvc_train_loop* loop, cvc_stream_t stream) {
    if (!loop) return CVC_E_BADARG;
    int rc = validate(*loop, false);
    if (rc) return rc;
    g_prof_loop = loop->kind;
    return run_fwd(*loop, (hipStream_t)stream, nullptr);
}

extern "C" int cvc_train_loop_bwd(const cvc_train_loop* loop, cvc_stream_t stream) {
    if (!loop) return CVC_E_BADARG;
    int rc = validate(*loop, true);
    if (rc) return rc;
    return loop->kind == 0 ? run_bwd(loop, nullptr, (hipStream_t)stream, nullptr) : run_bwd(nullptr, loop, (hipStream_t)stream, nullptr);
}

extern "C" int cvc_train_loops_bwd_joint(const cvc_train_loop* loop_a, const cvc_train_loop* loop_c, cvc_stream_t stream) {
    if (!loop_a || !loop_c || loop_a->kind != 0 || loop_c->kind != 1) return CVC_E_BADARG;
    int rc = validate(*loop_a, true);
    if (!rc) rc = validate(*loop_c, true);
    if (rc) return rc;
    if (loop_a->B + loop_c->B > 128 || (loop_a->B + loop_c->B > 64 && cvc_gemm_packed_split(-1) == 0) || loop_a->T != loop_c->T || loop_a->R != loop_c->R || loop_a->w_ih_att != loop_c->w_ih_att ||
        loop_a->w_hh_att != loop_c->w_hh_att || loop_a->w_ih_lang != loop_c->w_ih_lang || loop_a->w_hh_lang != loop_c->w_hh_lang)
        return CVC_E_BADARG;
    return run_bwd(loop_a, loop_c, (hipStream_t)stream, nullptr);
}

// ---- measurement aid: per-launch timing of the drivers' entry points (bench.py --mode train).  enable > 0: room for `enable`
// launches (event pairs created here, recording starts); 0: stop and free.  Not thread-safe, not for captured streams.
extern "C" int cvc_train_loop_profile(int enable) {
    if (g_prof.ev != nullptr) {
        for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        delete[] g_prof.ev; delete[] g_prof.kind; delete[] g_prof.loop;
        g_prof = Prof{};
    }
    if (enable <= 0) return 0;
    g_prof.cap = enable;
    g_prof.ev = new hipEvent_t[2 * (size_t)enable];
    g_prof.kind = new int[enable];
    g_prof.loop = new int[enable];
    for (int i = 0; i < 2 * enable; ++i)
        if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return CVC_E_BADARG;
    g_prof.n = 0;
    g_prof.on = true;
    return 0;
}

// -> number of launches recorded since cvc_train_loop_profile(n); kind[i] (LaunchKind), loop[i] (0 / 1: forward of loop A / C,
// 2 / 3: backward), ms[i].  Waits for the recorded launches to finish.
extern "C" int cvc_train_loop_profile_read(int* kind, int* loop, float* ms, int cap) {
    if (!g_prof.on || !kind || !loop || !ms) return 0;
    const int n = g_prof.n < cap ? g_prof.n : cap;
    for (int i = 0; i < n; ++i) {
        (void)hipEventSynchronize(g_prof.ev[2 * i + 1]);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        kind[i] = g_prof.kind[i]; loop[i] = g_prof.loop[i]; ms[i] = t;
    }
    g_prof.n = 0;
    return n;
}

// launches one call enqueues (what the drivers above count): the zero fills + per step 2 cells (+ h2attn, 2 attention passes),
// backward per step 2 gate-gradient kernels, the backward-data products (+ the attention backward's 3 launches)
extern "C" int cvc_train_loop_launches(const cvc_train_loop* loop, int backward) {
    if (!loop || loop->T < 1) return 0;
    const int T = loop->T;
    if (!backward) return 6 + T * (loop->kind == 0 ? 5 : 2);
    // backward-data products leave their K-slice planes to the readers (no summing launch); attention backward = score pass + softmax
    // backward + score backward
    const int per = loop->kind == 0 ? (1 + 1 + 3 + 1 + 1 + 1) : (1 + 1 + 1 + 1);
    return T * per - 1 + (loop->dgsum_att ? 1 : 0) + (loop->dgsum_lang ? 1 : 0);          // (step 0 has no recurrent product of the attention cell)
}
