// Host-side decode drivers: cvc_decode_greedy / cvc_decode_beam enqueue the whole T-step caption decode
// (model/captioner.py:384-443) on the caller's stream from a bound descriptor -- the launch list that cvc/decode.py used to walk
// in Python (one ctypes call per kernel), now one call per decode.  Nothing here touches the device except through the C-ABI
// entry points of this library and stream-ordered memcpy / memset, so the call is capturable into a HIP graph.
#include <hip/hip_runtime.h>
#include <new>
#include <stdint.h>
#include "../../include/cvc_hip.h"
#include "../../include/cvc_hip_blocks.h"
#include "../../include/cvc_hip_experimental.h"

// stream-K launch shapes of the packed path (include/cvc_hip.h, "Grouped stream-K form"): host arithmetic done once per plan
struct gsk_launch {
    int U;
    int unit0[2], maxseg[2];
};
struct cvc_decode_plan {
    cvc_decode_desc d;
    int launches;
    gsk_launch ga, go, gl;      // {att-early, logits}, {logits}, {lang-early, h2attn}
};

namespace {

#define CVC_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        ++n;                     \
        if (rc_ != 0) return rc_; \
    } while (0)

inline float* quad_off(float* buf, int k0) { return buf + (size_t)(k0 / 4) * 64 * 4; }
inline void* frag_off(void* xb, int k0) { return (char*)xb + (size_t)(k0 / 16) * 3 * 1024; }

int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// State reset as ordinary kernel launches (16-byte granules): inside a captured graph they are plain kernel nodes, ordered like
// every other launch of the decode (memcpy / memset nodes may be served by a copy engine).
__global__ __launch_bounds__(256) void reset_kernel(uint4* dst, const uint4* src, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src != nullptr ? src[i] : uint4{0, 0, 0, 0};
}
int reset(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if ((bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) {       // odd sizes (the BOS row of a ragged batch)
        return hip_rc(src ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) : hipMemsetAsync(dst, 0, bytes, st));
    }
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(reset_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n16);
    return hip_rc(hipGetLastError());
}

void attn_sets(const cvc_decode_desc& d, int t, int rows, cvc_attn_set* sets) {
    sets[0] = cvc_attn_set{d.ppool, d.pool, d.mask, nullptr, d.scores_r, nullptr, d.att_steps + (size_t)t * rows * d.N, nullptr,
                           d.N, d.stream_r};
    sets[1] = cvc_attn_set{d.pconv, d.conv, nullptr, nullptr, d.scores_f, nullptr, d.attn_f, nullptr, d.F, d.stream_f};
}

// ---- packed path: greedy, <= 64 rows (7 launches per step)
int run_packed(cvc_decode_plan* p, hipStream_t st) {
    const cvc_decode_desc& d = p->d;
    const int rows = d.B, R = d.R, E = d.E, A = d.A, V = d.V, N = d.N;
    const size_t qbytes = (size_t)64 * 4 * sizeof(float);          // one quad of the activation layout
    int n = 0;
    // reset: XA of step 0, zero XL / cell states / BOS words
    CVC_TRY(reset(d.xa[0], d.xa0_init, (size_t)((2 * R + E) / 4) * qbytes, st));
    CVC_TRY(reset(d.xl[0], nullptr, (size_t)(3 * R / 4) * qbytes, st));
    CVC_TRY(reset(d.ca[0], nullptr, (size_t)(R / 4) * qbytes, st));
    CVC_TRY(reset(d.cl[0], nullptr, (size_t)(R / 4) * qbytes, st));
    CVC_TRY(reset(d.words, nullptr, (size_t)rows * sizeof(int64_t), st));
    // hoisted fc gate term + both biases (decoder_core.py:46)
    cvc_gemm_seg seg{d.fc, nullptr, d.w_fc, R, R, d.ld_w_fc, 0};
    CVC_TRY(cvc_linear_fwd(&seg, 1, d.b_ih_att, d.b_hh_att, rows, 4 * R, d.gate_fc, 4 * R, st));
    const int nblk_v = (V + 31) / 32;
    for (int t = 0; t < d.T; ++t) {
        const int rd = t & 1, wr = (t + 1) & 1;
        float *XA_r = d.xa[rd], *XA_w = d.xa[wr], *XL_r = d.xl[rd], *XL_w = d.xl[wr];
        CVC_TRY(cvc_packed_lstm_fwd((const float*)d.w_att, XA_r, 2 * R + E, nullptr, nullptr, d.gate_fc, d.ca[rd], rows, R,
                                    quad_off(XL_r, R), quad_off(XA_w, R + E), d.ca[wr], st));
        CVC_TRY(cvc_packed_linear_fwd((const float*)d.w_h, quad_off(XL_r, R), R, nullptr, rows, A, d.qsplit, d.q_parts, A, nullptr, st));
        cvc_attn_set sets[2];
        attn_sets(d, t, rows, sets);
        CVC_TRY(cvc_attn_scores_qparts(d.attn_kind, d.q_parts, d.qsplit, d.b_h, d.w_a, d.b_a, d.inv_temp, sets, 2, d.B, 1, A, st));
        CVC_TRY(cvc_attn_wsum_quad(sets, 2, d.B, 1, R, XL_r, st));
#ifdef CVC_EXPERIMENTAL
        if (d.lang_ksx)      // K-split gate GEMM, every K slice of a tile finishing one of its blocks after the in-launch exchange
            CVC_TRY(cvc_packed_lstm_ksx_fwd((const float*)d.w_lang, XL_r, 3 * R, d.b_ih_lang, d.b_hh_lang, nullptr, nullptr, nullptr, d.cl[rd],
                                            rows, R, XA_w, quad_off(XL_w, 2 * R), d.cl[wr], d.ksx_slab, d.ksx_flags, (unsigned)(t + 1), st));
        else
#endif
        CVC_TRY(cvc_packed_lstm_fwd((const float*)d.w_lang, XL_r, 3 * R, d.b_ih_lang, d.b_hh_lang, nullptr, d.cl[rd], rows, R, XA_w,
                                    quad_off(XL_w, 2 * R), d.cl[wr], st));
        CVC_TRY(cvc_packed_linear_fwd((const float*)d.w_o, XA_w, R, d.b_o, rows, V, 1, nullptr, V, d.top2_part, st));
        CVC_TRY(cvc_top2_final(d.top2_part, nblk_v, rows, d.unk_idx, d.words + (size_t)(t + 1) * rows, 1,
                               d.logprob ? d.logprob + (size_t)t * rows : nullptr, d.embed, E, quad_off(XA_w, R), 0, st));
    }
    p->launches = n;
    return 0;
}

// ---- packed path, embedding-gate schedule: the embedded word's share of the att-LSTM gates is a row of a per-checkpoint table
// (cvc_packed_lstm_embgate_fwd); XA = [h_lang | h_att], K = 2R; word selection writes only the word (no embedded-word buffer)
int run_packed_eg(cvc_decode_plan* p, hipStream_t st) {
    const cvc_decode_desc& d = p->d;
    const int rows = d.B, R = d.R, A = d.A, V = d.V;
    const size_t qbytes = (size_t)64 * 4 * sizeof(float);
    int n = 0;
    CVC_TRY(reset(d.xa[0], nullptr, (size_t)(2 * R / 4) * qbytes, st));
    CVC_TRY(reset(d.xl[0], nullptr, (size_t)(3 * R / 4) * qbytes, st));
    CVC_TRY(reset(d.ca[0], nullptr, (size_t)(R / 4) * qbytes, st));
    CVC_TRY(reset(d.cl[0], nullptr, (size_t)(R / 4) * qbytes, st));
    CVC_TRY(reset(d.words, nullptr, (size_t)rows * sizeof(int64_t), st));
    cvc_gemm_seg seg{d.fc, nullptr, d.w_fc, R, R, d.ld_w_fc, 0};
    CVC_TRY(cvc_linear_fwd(&seg, 1, d.b_ih_att, d.b_hh_att, rows, 4 * R, d.gate_fc, 4 * R, st));
    const long long ws_att = (long long)(2 * R / 4) * 128, ws_lang = (long long)(3 * R / 4) * 128;     // block strides of the packs
    const bool split2 = cvc_gemm_packed_split(-1) == 2;
    for (int t = 0; t < d.T; ++t) {
        const int rd = t & 1, wr = (t + 1) & 1;
        float *XA_r = d.xa[rd], *XA_w = d.xa[wr], *XL_r = d.xl[rd], *XL_w = d.xl[wr];
        // step 0 multiplies the all-zero initial state (h_lang = h_att = 0): an exact zero times a finite weight adds nothing, so the
        // attention cell contracts over one chunk only (its gates are the table row + the hoisted fc term) and the language cell
        // below stops before its h_lang columns
        const bool first = t == 0 && split2;
        CVC_TRY(cvc_packed_lstm_embgate_ex_fwd((const float*)d.w_att, ws_att, XA_r, first ? 32 : 2 * R, nullptr, nullptr, d.gate_fc,
                                               d.emb_gate, d.words + (size_t)t * rows, d.ca[rd], rows, R, quad_off(XL_r, R),
                                               quad_off(XA_w, R), d.ca[wr], d.att_w_cached, st));
        CVC_TRY(cvc_packed_linear_fwd((const float*)d.w_h, quad_off(XL_r, R), R, nullptr, rows, A, d.qsplit, d.q_parts, A, nullptr, st));
        cvc_attn_set sets[2];
        attn_sets(d, t, rows, sets);
        CVC_TRY(cvc_attn_scores_qparts(d.attn_kind, d.q_parts, d.qsplit, d.b_h, d.w_a, d.b_a, d.inv_temp, sets, 2, d.B, 1, A, st));
        CVC_TRY(cvc_attn_wsum_quad(sets, 2, d.B, 1, R, XL_r, st));
        if (first)           // (h_lang(-1) = 0 sits in the last third of XL: K = 2R of the 3R packed columns)
            CVC_TRY(cvc_packed_lstm_late_fwd((const float*)d.w_lang, ws_lang, XL_r, 2 * R, d.b_ih_lang, d.b_hh_lang, nullptr, d.cl[rd], rows,
                                             R, XA_w, quad_off(XL_w, 2 * R), d.cl[wr], nullptr, st));
#ifdef CVC_EXPERIMENTAL
        else if (d.lang_ksx) // K-split gate GEMM, every K slice of a tile finishing one of its blocks after the in-launch exchange
            CVC_TRY(cvc_packed_lstm_ksx_fwd((const float*)d.w_lang, XL_r, 3 * R, d.b_ih_lang, d.b_hh_lang, nullptr, nullptr, nullptr, d.cl[rd],
                                            rows, R, XA_w, quad_off(XL_w, 2 * R), d.cl[wr], d.ksx_slab, d.ksx_flags, (unsigned)(t + 1), st));
#endif
        else
        CVC_TRY(cvc_packed_lstm_fwd((const float*)d.w_lang, XL_r, 3 * R, d.b_ih_lang, d.b_hh_lang, nullptr, d.cl[rd], rows, R, XA_w,
                                    quad_off(XL_w, 2 * R), d.cl[wr], st));
        CVC_TRY(cvc_packed_linear_fwd((const float*)d.w_o, XA_w, R, d.b_o, rows, V, 1, nullptr, V, d.top2_part, st));
        CVC_TRY(cvc_top2_final(d.top2_part, (V + 31) / 32, rows, d.unk_idx, d.words + (size_t)(t + 1) * rows, 1,
                               d.logprob ? d.logprob + (size_t)t * rows : nullptr, nullptr, 0, nullptr, 0, st));
    }
    p->launches = n;
    return 0;
}

#ifdef CVC_EXPERIMENTAL
// ---- packed path, grouped stream-K schedule (7 launches per step, every one of them chip-filling):
//   att-late    : W_att[:, emb] x relu(Emb[word]) + partial tiles of (h_lang, h_att) from the previous step's launch 6 + hoisted fc
//                 term, cell update -> h_att(t)
//   stream-K    : lang-early (h_att(t), h_lang(t-1))  ||  h2attn (h_att(t))
//   scores      : query summed from the h2attn partial tiles
//   weighted sum
//   lang-late   : W_lang[:, ctx] x ctx + partial tiles of launch 2 + biases, cell update -> h_lang(t)
//   stream-K    : att-early of step t+1 (h_lang(t), h_att(t))  ||  vocabulary logits (h_lang(t))
//   word select : logits summed from the partial tiles, top-2 / UNK rule, next step's embedded word
int run_packed_gsk(cvc_decode_plan* p, hipStream_t st) {
    const cvc_decode_desc& d = p->d;
    const int rows = d.B, R = d.R, E = d.E, A = d.A, V = d.V;
    const size_t qbytes = (size_t)64 * 4 * sizeof(float);
    int n = 0;
    CVC_TRY(reset(d.xa[0], d.xa0_init, (size_t)((2 * R + E) / 4) * qbytes, st));
    CVC_TRY(reset(d.xl[0], nullptr, (size_t)(3 * R / 4) * qbytes, st));
    CVC_TRY(reset(d.ca[0], nullptr, (size_t)(R / 4) * qbytes, st));
    CVC_TRY(reset(d.cl[0], nullptr, (size_t)(R / 4) * qbytes, st));
    CVC_TRY(reset(d.words, nullptr, (size_t)rows * sizeof(int64_t), st));
    cvc_gemm_seg seg{d.fc, nullptr, d.w_fc, R, R, d.ld_w_fc, 0};
    CVC_TRY(cvc_linear_fwd(&seg, 1, d.b_ih_att, d.b_hh_att, rows, 4 * R, d.gate_fc, 4 * R, st));
    const long long ws_att = (long long)((2 * R + E) / 4) * 128, ws_lang = (long long)(3 * R / 4) * 128, ws_r = (long long)(R / 4) * 128;
    const float* w_att = (const float*)d.w_att;
    const float* w_lang = (const float*)d.w_lang;
    const int nblk_v = (V + 31) / 32;
    const cvc_gsk_segs seg_att{d.slab_att, p->ga.unit0[0], 2 * R / 32, p->ga.U, p->ga.maxseg[0]};
    const cvc_gsk_segs seg_o_a{d.slab_o, p->ga.unit0[1], R / 32, p->ga.U, p->ga.maxseg[1]};
    const cvc_gsk_segs seg_o_o{d.slab_o, 0, R / 32, p->go.U, p->go.maxseg[0]};
    const cvc_gsk_segs seg_lang{d.slab_lang, p->gl.unit0[0], 2 * R / 32, p->gl.U, p->gl.maxseg[0]};
    const cvc_gsk_segs seg_q{d.slab_q, p->gl.unit0[1], R / 32, p->gl.U, p->gl.maxseg[1]};
    for (int t = 0; t < d.T; ++t) {
        const int rd = t & 1, wr = (t + 1) & 1;
        float *XA_r = d.xa[rd], *XA_w = d.xa[wr], *XL_r = d.xl[rd], *XL_w = d.xl[wr];
        // step 0 starts from the zero state: the early K range contributes nothing
        CVC_TRY(cvc_packed_lstm_late_fwd(w_att + (size_t)(R / 4) * 128, ws_att, quad_off(XA_r, R), E, nullptr, nullptr, d.gate_fc,
                                         d.ca[rd], rows, R, quad_off(XL_r, R), quad_off(XA_w, R + E), d.ca[wr],
                                         t == 0 ? nullptr : &seg_att, st));
        cvc_gsk_group gl[2] = {
            {w_lang, ws_lang, XL_r, R / 8, 2 * R / 32, 0, R / 32, d.slab_lang, p->gl.maxseg[0]},
            {(const float*)d.w_h, ws_r, quad_off(XL_r, R), A / 32, R / 32, 0, 0, d.slab_q, p->gl.maxseg[1]}};
        CVC_TRY(cvc_gsk_gemm(gl, 2, p->gl.U, st));
        cvc_attn_set sets[2];
        attn_sets(d, t, rows, sets);
        CVC_TRY(cvc_attn_scores_qslab(d.attn_kind, &seg_q, d.b_h, d.w_a, d.b_a, d.inv_temp, sets, 2, d.B, 1, A, st));
        CVC_TRY(cvc_attn_wsum_quad(sets, 2, d.B, 1, R, XL_r, st));
        CVC_TRY(cvc_packed_lstm_late_fwd(w_lang, ws_lang, XL_r, R, d.b_ih_lang, d.b_hh_lang, nullptr, d.cl[rd], rows, R, XA_w,
                                         quad_off(XL_w, 2 * R), d.cl[wr], &seg_lang, st));
        const bool last = t + 1 == d.T;
        cvc_gsk_group ga[2] = {
            {w_att, ws_att, XA_w, R / 8, 2 * R / 32, R / 32, E / 32, d.slab_att, p->ga.maxseg[0]},
            {(const float*)d.w_o, ws_r, XA_w, nblk_v, R / 32, 0, 0, d.slab_o, last ? p->go.maxseg[0] : p->ga.maxseg[1]}};
        if (last) CVC_TRY(cvc_gsk_gemm(ga + 1, 1, p->go.U, st));        // no next step: the vocabulary projection alone
        else CVC_TRY(cvc_gsk_gemm(ga, 2, p->ga.U, st));
        CVC_TRY(cvc_top2_slab(last ? &seg_o_o : &seg_o_a, d.b_o, V, rows, d.unk_idx, d.words + (size_t)(t + 1) * rows, 1,
                              d.logprob ? d.logprob + (size_t)t * rows : nullptr, d.embed, E, quad_off(XA_w, R), 0, st));
    }
    p->launches = n;
    return 0;
}

#endif  // CVC_EXPERIMENTAL

// ---- tile path: beam search or more than 64 rows (12 launches per step)
int run_tile(cvc_decode_plan* p, hipStream_t st) {
    const cvc_decode_desc& d = p->d;
    const int B = d.B, beam = d.beam, rows = B * beam, R = d.R, E = d.E, A = d.A, V = d.V;
    int n = 0;
    CVC_TRY(reset(d.words, nullptr, (size_t)rows * sizeof(int64_t), st));
    if (beam > 1) {
        CVC_TRY(reset(d.score, nullptr, (size_t)2 * rows * sizeof(float), st));
        CVC_TRY(reset(d.done, nullptr, (size_t)2 * rows, st));
    }
    // once per decode: hoisted fc gate term (+ both biases), one row per clip; step-0 operands from the zero state
    CVC_TRY(cvc_tile_pack_rows(d.fc, R, nullptr, 0, B, R, d.xff, d.xff_stride, st));
    CVC_TRY(cvc_tile_gemm(d.w_fc_frag, d.xff, d.xff_stride, R, B, 4 * R, d.ks_fc, d.parts_fc, 4 * R, (long long)B * 4 * R, st));
    CVC_TRY(cvc_tile_linear_finish(d.parts_fc, d.ks_fc, (long long)B * 4 * R, 4 * R, d.b_ih_att, d.b_hh_att, B, 4 * R, d.gate_fc,
                                   4 * R, st));
    void* xl_hatt = frag_off(d.xlf, R);
    void* xl_hlang = frag_off(d.xlf, 2 * R);
    // embedding-gate form (d.emb_gate): XA = [h_lang | h_att], the word's share of the gates is added by the finishing launch
    const bool eg = d.emb_gate != nullptr;
    const int E_pack = eg ? 0 : E, ka = 2 * R + E_pack;
    CVC_TRY(cvc_tile_reorder_pack(nullptr, d.words, beam, d.zero_state, d.zero_state, d.zero_state, d.zero_state, d.embed, E_pack, V,
                                  d.c_att_prev, d.c_lang_prev, d.xaf, d.xaf_stride, xl_hlang, d.xlf_stride, rows, R, st));
    const long long gs = (long long)rows * 4 * R;
    for (int t = 0; t < d.T; ++t) {
        CVC_TRY(cvc_tile_gemm(d.w_att, d.xaf, d.xaf_stride, ka, rows, 4 * R, d.ks_gate, d.parts_gate, 4 * R, gs, st));
        if (eg)
            CVC_TRY(cvc_tile_lstm_finish_embgate(d.parts_gate, d.ks_gate, gs, nullptr, nullptr, d.gate_fc, beam, d.emb_gate,
                                                 d.words + (size_t)t * rows, V, d.c_att_prev, rows, R, d.c_att, d.h_att, xl_hatt,
                                                 d.xlf_stride, nullptr, 0, st));
        else
            CVC_TRY(cvc_tile_lstm_finish(d.parts_gate, d.ks_gate, gs, nullptr, nullptr, d.gate_fc, beam, d.c_att_prev, rows, R, d.c_att,
                                         d.h_att, xl_hatt, d.xlf_stride, nullptr, 0, st));
        CVC_TRY(cvc_tile_gemm(d.w_h, xl_hatt, d.xlf_stride, R, rows, A, d.ks_q, d.q_parts, A, (long long)rows * A, st));
        CVC_TRY(cvc_tile_linear_finish(d.q_parts, d.ks_q, (long long)rows * A, A, d.b_h, nullptr, rows, A, d.q, A, st));
        cvc_attn_set sets[2];
        attn_sets(d, t, rows, sets);
        CVC_TRY(cvc_attn_scores(d.attn_kind, d.q, d.w_a, d.b_a, d.inv_temp, sets, 2, B, beam, A, st));
        CVC_TRY(cvc_attn_wsum_frag(sets, 2, B, beam, R, d.xlf, d.xlf_stride, st));
        CVC_TRY(cvc_tile_gemm(d.w_lang, d.xlf, d.xlf_stride, 3 * R, rows, 4 * R, d.ks_gate, d.parts_gate, 4 * R, gs, st));
        CVC_TRY(cvc_tile_lstm_finish(d.parts_gate, d.ks_gate, gs, d.b_ih_lang, d.b_hh_lang, nullptr, 1, d.c_lang_prev, rows, R,
                                     d.c_lang, d.h_lang, d.xhf, d.xhf_stride, nullptr, 0, st));
        CVC_TRY(cvc_tile_gemm(d.w_o, d.xhf, d.xhf_stride, R, rows, V, d.ks_o, d.parts_o, V, (long long)rows * V, st));
        // beams: the selection's row scan sums the K-slice slabs itself (round 6: its float4 form, slab 0 + ... + bias in the finishing
        // pass's order -- the same logits bit for bit; the GENERAL scan doing that had measured 54 us against 12 + 24 us in round 4);
        // the finished matrix is neither written nor read back.  CVC_BEAM_FINISH=1 keeps the separate pass (A/B).
        static const bool keep_finish = [] { const char* e = getenv("CVC_BEAM_FINISH"); return e && e[0] == '1'; }();
        const bool fused_sel = beam > 1 && !keep_finish && (V & 3) == 0 && (d.ks_o == 2 || d.ks_o == 4 || d.ks_o == 6 || d.ks_o == 8);
        if (!fused_sel)
            CVC_TRY(cvc_tile_linear_finish(d.parts_o, d.ks_o, (long long)rows * V, V, d.b_o, nullptr, rows, V, d.logits, V, st));
        int64_t* word_next = d.words + (size_t)(t + 1) * rows;
        const int64_t* parent = nullptr;
        if (beam == 1) {
            CVC_TRY(cvc_top2_unk(d.logits, rows, V, d.unk_idx, word_next, 1, d.logprob ? d.logprob + (size_t)t * rows : nullptr, st));
        } else {
            const int srd = t & 1, swr = (t + 1) & 1;
            int64_t* par = d.parent + (size_t)t * rows;
            CVC_TRY(cvc_beam_select_parts(fused_sel ? d.parts_o : d.logits, fused_sel ? d.ks_o : 1, fused_sel ? (long long)rows * V : 0,
                                          fused_sel ? d.b_o : nullptr, d.score + (size_t)srd * rows, d.done + (size_t)srd * rows, B, beam, V,
                                          d.unk_idx, t == 0 ? 1 : 0, par, word_next, d.score + (size_t)swr * rows,
                                          d.done + (size_t)swr * rows, d.beam_ws, st));
            parent = par;
        }
        if (t + 1 < d.T)
            CVC_TRY(cvc_tile_reorder_pack(parent, word_next, beam, d.h_att, d.c_att, d.h_lang, d.c_lang, d.embed, E_pack, V, d.c_att_prev,
                                          d.c_lang_prev, d.xaf, d.xaf_stride, xl_hlang, d.xlf_stride, rows, R, st));
    }
    p->launches = n;
    return 0;
}

int validate(const cvc_decode_desc& d) {
    if (d.B < 1 || d.beam < 1 || d.beam > 8 || d.T < 1 || d.N < 1 || d.F < 1 || d.V < 2) return CVC_E_BADARG;
    if (d.attn_kind != CVC_ATTN_ADDITIVE && d.attn_kind != CVC_ATTN_DOT) return CVC_E_BADARG;
    if (!d.fc || !d.conv || !d.pconv || !d.pool || !d.ppool || !d.words || !d.att_steps || !d.embed || !d.b_o || !d.b_h) return CVC_E_BADARG;
    if (d.attn_kind == CVC_ATTN_ADDITIVE && !d.w_a) return CVC_E_BADARG;
    if (!d.w_att || !d.w_lang || !d.w_h || !d.w_o || !d.gate_fc || !d.scores_r || !d.scores_f || !d.attn_f || !d.q_parts) return CVC_E_BADARG;
#ifndef CVC_EXPERIMENTAL
    if (d.gsk_nwg > 0 || d.lang_ksx) return CVC_E_BADARG;     // schedules of cvc_hip_experimental.h: not in this build
#endif
    if (d.path == 0) {
        if (d.beam != 1 || d.B > 64 || (d.R & 31) || (d.E & 31) || (d.A & 31) || d.qsplit < 1) return CVC_E_BADARG;
        if (!d.w_fc || !d.top2_part || !d.xa[0] || !d.xa[1] || !d.xl[0] || !d.xl[1] || !d.ca[0] || !d.ca[1] || !d.cl[0] || !d.cl[1] ||
            (!d.xa0_init && !d.emb_gate))
            return CVC_E_BADARG;
        if (d.emb_gate != nullptr && d.gsk_nwg > 0) return CVC_E_BADARG;
        if (d.lang_ksx && (!d.ksx_slab || !d.ksx_flags || d.R != 2048 || d.T < 2 || d.gsk_nwg > 0)) return CVC_E_BADARG;
        if (d.gsk_nwg < 0 || (d.gsk_nwg > 0 && ((d.R & 63) || !d.slab_att || !d.slab_lang || !d.slab_q || !d.slab_o))) return CVC_E_BADARG;
    } else if (d.path == 1) {
        if ((d.R & 15) || (d.E & 15) || d.ks_gate < 1 || d.ks_q < 1 || d.ks_o < 1 || d.ks_fc < 1) return CVC_E_BADARG;
        if (!d.w_fc_frag || !d.xaf || !d.xlf || !d.xhf || !d.xff || !d.parts_gate || !d.parts_o || !d.parts_fc || !d.logits || !d.q ||
            !d.h_att || !d.c_att || !d.h_lang || !d.c_lang || !d.c_att_prev || !d.c_lang_prev || !d.zero_state)
            return CVC_E_BADARG;
        if (d.beam > 1 && (!d.score || !d.done || !d.parent || !d.beam_ws)) return CVC_E_BADARG;
    } else {
        return CVC_E_BADARG;
    }
    return 0;
}

}  // namespace

extern "C" int cvc_decode_plan_create(const cvc_decode_desc* desc, cvc_decode_plan** plan) {
    if (!desc || !plan) return CVC_E_BADARG;
    int rc = validate(*desc);
    if (rc) return rc;
    cvc_decode_plan* p = new (std::nothrow) cvc_decode_plan;
    if (!p) return CVC_E_BADARG;
    p->d = *desc;
    p->launches = 0;
#ifdef CVC_EXPERIMENTAL
    if (desc->path == 0 && desc->gsk_nwg > 0) {
        const int R = desc->R, nt_r = R / 64, nt_v = ((desc->V + 31) / 32 + 7) / 8, nt_a = (desc->A / 32 + 7) / 8;
        const int nta[2] = {nt_r, nt_v}, nca[2] = {2 * R / 32, R / 32};
        const int ntl[2] = {nt_r, nt_a};
        rc = cvc_gsk_plan(nta, nca, 2, desc->gsk_nwg, &p->ga.U, p->ga.unit0, p->ga.maxseg);
        if (!rc) rc = cvc_gsk_plan(nta + 1, nca + 1, 1, desc->gsk_nwg, &p->go.U, p->go.unit0, p->go.maxseg);
        if (!rc) rc = cvc_gsk_plan(ntl, nca, 2, desc->gsk_nwg, &p->gl.U, p->gl.unit0, p->gl.maxseg);
        if (rc) { delete p; return rc; }
    }
#endif
    *plan = p;
    return 0;
}

extern "C" int cvc_decode_plan_set_features(cvc_decode_plan* plan, const float* fc, const float* conv, const float* pconv,
                                            const float* pool, const float* ppool, const uint8_t* mask) {
    if (!plan || !fc || !conv || !pconv || !pool || !ppool) return CVC_E_BADARG;
    plan->d.fc = fc; plan->d.conv = conv; plan->d.pconv = pconv; plan->d.pool = pool; plan->d.ppool = ppool; plan->d.mask = mask;
    return 0;
}

extern "C" void cvc_decode_plan_destroy(cvc_decode_plan* plan) { delete plan; }

extern "C" int cvc_decode_num_launches(const cvc_decode_plan* plan) { return plan ? plan->launches : 0; }

extern "C" int cvc_decode_greedy(cvc_decode_plan* plan, cvc_stream_t stream) {
    if (!plan || plan->d.beam != 1) return CVC_E_BADARG;
    if (plan->d.path == 0) {
        if (plan->d.emb_gate != nullptr) return run_packed_eg(plan, (hipStream_t)stream);
#ifdef CVC_EXPERIMENTAL
        if (plan->d.gsk_nwg > 0) return run_packed_gsk(plan, (hipStream_t)stream);
#endif
        return run_packed(plan, (hipStream_t)stream);
    }
    return run_tile(plan, (hipStream_t)stream);
}

extern "C" int cvc_decode_beam(cvc_decode_plan* plan, cvc_stream_t stream) {
    if (!plan || plan->d.beam < 2 || plan->d.path != 1) return CVC_E_BADARG;
    return run_tile(plan, (hipStream_t)stream);
}
