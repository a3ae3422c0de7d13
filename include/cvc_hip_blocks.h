/*
 * cvc_hip_blocks.h -- the per-kernel building blocks of libcvc_hip.so: what the whole-decode drivers (cvc_decode_greedy / _beam),
 * the training-loop drivers (cvc_train_loop_fwd / _bwd) and the host mirror's eager launch lists are composed of.  NOT part of the
 * exported drop-in ABI (hidden visibility): unit tests and cvc/decode.py reach them through cvc_block("name") (cvc_hip.h).
 * Contracts: the section of cvc_hip.h that describes the corresponding driver or core entry point (same argument conventions:
 * raw device pointers, sizes, the launch stream; 0 / hipError_t / CVC_E_*).
 */
#ifndef CVC_HIP_BLOCKS_H
#define CVC_HIP_BLOCKS_H
#include "cvc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

int cvc_attn_scores(int kind, const float* q, const float* w_a, const float* b_a, float inv_temp,
                    const cvc_attn_set* sets, int nsets, int nclip, int nq, int A, cvc_stream_t stream);
int cvc_attn_wsum(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, float* ctx_sum,
                  cvc_stream_t stream);
int cvc_attn_scores_qparts(int kind, const float* q_parts, int q_nparts, const float* q_bias, const float* w_a,
                           const float* b_a, float inv_temp, const cvc_attn_set* sets, int nsets, int nclip,
                           int nq, int A, cvc_stream_t stream);
int cvc_attn_wsum_quad(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, float* ctx_sum_q,
                       cvc_stream_t stream);
int cvc_attn_wsum_frag(const cvc_attn_set* sets, int nsets, int nclip, int nq, int R, void* ctx_frag,
                       long long frag_mblk_stride, cvc_stream_t stream);
int cvc_attn_wsum_quad_rm(const cvc_attn_set* sets, int nsets, int nclip, int R, float* ctx_sum_q, float* ctx_sum_rm,
                          cvc_stream_t stream);            /* one query per clip (nq = 1), nclip <= 64 */
int cvc_attn_bwd_pair(int kind, const cvc_grad_src* q, const float* q_bias, const float* w_a, float inv_temp,
                      const cvc_attn_set* sets, int nsets, const cvc_grad_src* d_ctx, int nclip, int nq, int A, int R,
                      float* d_q, float* d_q_q, float* d_w_part, float* const* d_proj, float* const* d_ctxfeat,
                      cvc_stream_t stream);
/* d_feat[b, i, :] += sum_t attn[t][b][i] * d_ctx_all[t][b (of 128 rows)][:]  (t < T <= 32): the context-feature gradient of all T
 * steps of the training loop in one pass over d_feat [B, n, R] (cvc_train_loop.d_ctx_all; reference: the `bmm(att, context)` of
 * modules.py:66-69 / 150-153 under autograd, accumulated over the T decoder steps of captioner.py:242-270) */
int cvc_ctxfeat_bwd_steps(const float* attn, const float* d_ctx_all, int T, int B, int n, int R, float* d_feat, cvc_stream_t stream);
/* d_proj[b, i, :] += sum_t d_s[t][b][i] * w_a * (1 - tanh^2(proj[b, i, :] + q_t[b, :]))  (additive attention; t < T <= 32): the
 * projected-feature gradient of all T steps in one pass.  q: [T] blocks of q_step floats, each q_nplanes planes ([B, A], q_plane
 * floats apart) summed on load (+ q_bias): the h2attn outputs the forward kept (cvc_train_loop.q). */
int cvc_dproj_bwd_steps(const float* q, long long q_step, long long q_plane, int q_nplanes, const float* q_bias, const float* w_a,
                        const float* proj, const float* ds, int T, int B, int n, int A, float* d_proj, cvc_stream_t stream);
/* test / A-B hook of cvc_tile_gemm's 256 x 256 form (round 6, OFF by default: measured no gain inside the training step; applies to
 * M % 256 == 0, N % 256 == 0, M >= 512 and min_wgs .. 256 workgroups, any grid with min_wgs = 1): on = 1 / 0 / -1 (query only),
 * min_wgs > 0 sets the threshold.  Returns the previous on / off setting. */
int cvc_tile_gemm_big(int on, int min_wgs);
/* The K split to pass as `ksplit` for a dense product (callers without a split of their own: the backward pass's dense products) and
 * the grid cvc_tile_gemm launches for it: the largest split with >= 8 k steps per slice whose grid is at most one round of
 * workgroups (one per compute unit), else 1.  chunk_rows = rows one workgroup walks, workgroups = the grid size.  Host only; any of
 * the three outputs may be NULL.  CVC_E_BADARG for M, N < 1 or K not a positive multiple of 16. */
int cvc_tile_gemm_plan(int M, int N, int K, int* ksplit, int* chunk_rows, int* workgroups);
int cvc_linear_splitk_fwd(const cvc_gemm_seg* segs, int nsegs, const float* bias, int M, int Nout,
                          int ksplit, float* y_parts, cvc_stream_t stream);
int cvc_linear_top2_fwd(const cvc_gemm_seg* segs, int nsegs, const float* bias, int M, int Nout,
                        float* y_or_null, float* top2_part, cvc_stream_t stream);
int cvc_top2_final(const float* part, int nblocks, int M, int unk_idx, int64_t* word, int word_stride,
                   float* logprob, const float* table, int E, float* emb_out, int emb_ld,
                   cvc_stream_t stream);
int cvc_packed_lstm_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                        const float* gate_bias, const float* c_prev_q, int M, int R, float* h_dst1_q,
                        float* h_dst2_q, float* c_out_q, cvc_stream_t stream);
int cvc_packed_linear_fwd(const float* wp, const float* xq, int K, const float* bias, int M, int Nout,
                          int ksplit, float* y, int ldy, float* top2_part, cvc_stream_t stream);
int cvc_packed_lstm_embgate_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                const float* gate_bias, const float* emb_gate, const int64_t* word, const float* c_prev_q,
                                int M, int R, float* h_dst1_q, float* h_dst2_q, float* c_out_q, cvc_stream_t stream);
int cvc_packed_lstm_embgate_ex_fwd(const float* wp, long long w_blk_stride, const float* xq, int K, const float* b_ih,
                                   const float* b_hh, const float* gate_bias, const float* emb_gate, const int64_t* word,
                                   const float* c_prev_q, int M, int R, float* h_dst1_q, float* h_dst2_q,
                                   float* c_out_q, int w_cached, cvc_stream_t stream);
int cvc_packed_lstm_late_fwd(const float* wp, long long w_blk_stride, const float* xq, int K, const float* b_ih,
                             const float* b_hh, const float* gate_bias, const float* c_prev_q, int M, int R,
                             float* h_dst1_q, float* h_dst2_q, float* c_out_q, const cvc_gsk_segs* early,
                             cvc_stream_t stream);
int cvc_packed_lstm_train_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                              const float* c_prev, int M, int R, float* h_out, float* c_out, float* gates_out,
                              float* h_out2, float* h_out3, cvc_stream_t stream);   /* h_out2/3: further copies of h', nullable */
int cvc_packed_lstm_train_pre_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                  const float* gate_pre, const float* c_prev, int M, int R, float* h_out, float* c_out,
                                  float* gates_out, float* h_out2, float* h_out3, cvc_stream_t stream);
int cvc_packed_lstm_train_drop_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                   const float* gate_pre, const float* c_prev, int M, int R, float* h_out, float* c_out,
                                   float* gates_out, float* h_out2, float* h_drop_out, const uint32_t* rng_state,
                                   unsigned site, float p, cvc_stream_t stream);
int cvc_lstm_pointwise_bwd(const float* d_h, const float* d_c, const float* gates,
                           const float* c_prev, const float* c_new, int M, int R,
                           float* d_gates, float* d_c_prev, float* d_gates_q, cvc_stream_t stream);
int cvc_lstm_pointwise_bwd3(const float* d_h, const float* d_h2, const float* d_h3, const float* d_c,
                            const float* gates, const float* c_prev, const float* c_new, int M, int R,
                            float* d_gates, float* d_c_prev, float* d_gates_q, cvc_stream_t stream);
int cvc_lstm_pointwise_bwd3_drop(const float* d_h, const float* d_h2, const float* d_h3, const uint32_t* rng_state,
                                 unsigned site, float p, const float* d_c, const float* gates, const float* c_prev,
                                 const float* c_new, int M, int R, float* d_gates, float* d_c_prev, float* d_gates_q,
                                 cvc_stream_t stream);
int cvc_pack_lstm_weights(const float* w_ih, int K_ih, const float* w_hh, int K_hh, int R, float* wp,
                          cvc_stream_t stream);
int cvc_linear_nn_planes_fwd(const float* dy_q, int K, int M, const cvc_nn_seg* segs, int nsegs, int ksplit,
                             float* workspace, cvc_stream_t stream);
/* Training form of the per-step recurrence: additionally writes, for every step and direction, what autograd needs --
 * (r, z, n, W_hn h + b_hn) at gates + m * g_ld_m + t * g_ld_t + d * 4H + {0, H, 2H, 3H} (as cvc_gru_seq_persistent_train_fwd).
 * Any H % 8 == 0: the form config 5's encoder width (rnn_size 4096 -> H = 2048, backbone.py:103-104) trains on. */
int cvc_gru_seq_train_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih,
                          const float* b_hh, int M, int F, int H, int ndir, float* hq, float* y, long long y_ld_m,
                          long long y_ld_t, float* gates, long long g_ld_m, long long g_ld_t, cvc_stream_t stream);
/* cvc_lstm_pointwise_bwd4 (include/cvc_hip.h) for TWO independent argument sets in ONE launch: the gate-gradient kernels of the two
 * loops of the cyclical pass at a step of their joint back-propagation (same cell, same shape, ~10 us each and latency-bound).
 * Vector form only (every pointer 16-byte aligned, R % 4 == 0): CVC_E_BADARG otherwise, and the caller launches them one by one. */
typedef struct cvc_pw_bwd_args {
    cvc_grad_src d_h[3];
    const float* d_hd;
    const uint32_t* rng_state;
    unsigned site;
    float p;
    const float *d_c, *gates, *c_prev, *c_new;
    int M;
    float *d_gates, *d_c_prev, *d_gates_q, *dg_sum;
    int q_row0;
} cvc_pw_bwd_args;
int cvc_lstm_pointwise_bwd4_pair(const cvc_pw_bwd_args* a, const cvc_pw_bwd_args* b, int R, cvc_stream_t stream);
/* cvc_linear_nn_planes_fwd for TWO 64-row operand groups against one stream of the weights (the two loops of the cyclical pass at
 * B = 64 each share the LSTM cells, captioner.py:86-87): rows 0 .. M - 1 from dy_q, rows 64 .. 64 + M2 - 1 from dy_q2; planes are
 * [ksplit][128][ntot]; with reduce != 0 (or ksplit == 1) the result goes to the segments' dst [128 rows, ld_dst].  Under the same K
 * split a row gets the bits cvc_linear_nn_planes_fwd gives it. */
int cvc_linear_nn_planes2_fwd(const float* dy_q, const float* dy_q2, int K, int M, int M2, const cvc_nn_seg* segs, int nsegs,
                              int ksplit, float* workspace, int reduce, cvc_stream_t stream);
int cvc_beam_select_parts(const float* parts, int nparts, long long part_stride, const float* bias,
                          const float* score_in, const uint8_t* done_in, int B, int beam, int V, int unk_idx,
                          int first_step, int64_t* parent, int64_t* word, float* score_out, uint8_t* done_out,
                          float* workspace, cvc_stream_t stream);
int cvc_tile_lstm_finish(const float* parts, int nparts, long long part_stride, const float* b_ih, const float* b_hh,
                         const float* gate_bias, int gb_div, const float* c_prev, int M, int R, float* c_out,
                         float* h_out, void* frag1, long long frag1_stride, void* frag2, long long frag2_stride,
                         cvc_stream_t stream);
int cvc_tile_lstm_finish_embgate(const float* parts, int nparts, long long part_stride, const float* b_ih, const float* b_hh,
                                 const float* gate_bias, int gb_div, const float* emb_gate, const int64_t* word, int V,
                                 const float* c_prev, int M, int R, float* c_out, float* h_out, void* frag1,
                                 long long frag1_stride, void* frag2, long long frag2_stride, cvc_stream_t stream);
int cvc_tile_reorder_pack(const int64_t* parent, const int64_t* word, int beam, const float* h_att, const float* c_att,
                          const float* h_lang, const float* c_lang, const float* table, int E, int V,
                          float* c_att_prev, float* c_lang_prev, void* xa, long long xa_stride, void* xl_hlang,
                          long long xl_stride, int rows, int R, cvc_stream_t stream);
int cvc_decode_num_launches(const cvc_decode_plan* plan);                            /* kernels + copies per decode           */
/* test hook: every concat-GEMM on the generic direct-load kernel (see cvc_hip.h, cvc_linear_fwd) */
int cvc_gemm_force_generic(int on);
/* coverage hooks: select among kernel forms that the default path picks per shape (identical results) */
int cvc_tile_gemm_loaders(int on);
int cvc_gru_persistent_waves8(int on);   /* A/B + test hook: 1 (default) = 8 waves per workgroup where H % 256 == 0, 0 = always 4 */
/* ---- training forms of the once-per-clip encoder's small pieces (csrc/encoder_train.hip; model/backbone.py:55-81, 215-235,
 * 274-277, 325-333) -- used by the host mirror's encoder in train() mode (cvc/encoder_ops.py):
 *   cvc_relu_dropout_fwd / _bwd : y = max(x + bias, 0) * keep-mask multiplier of (site, flat index) -- the ReLU -> Dropout tail of the
 *       reference's Linear -> ReLU -> Dropout blocks, mask generated in the kernel (rng_state NULL or p == 0: plain ReLU);
 *       dx = dy * multiplier * [y > 0].  N % 4 == 0.
 *   cvc_bn_relu_train_fwd / _bwd : nn.BatchNorm1d on BATCH statistics over the rows of x [rows, C] (biased variance; running
 *       statistics updated with `momentum` and the unbiased variance, as the module does) followed by ReLU; save_mean / save_invstd [C]
 *       are kept for the backward, which also returns dgamma / dbeta.  workspace: cvc_bn_workspace(rows, C) floats.  C % 4 == 0.
 *   cvc_class_softmax_bwd : backward of cvc_class_softmax_fwd: p_rows [B*N, C] its softmax output, d_rows [B*N, C] / d_sim [B, C, N]
 *       (either nullable) the gradients of its two output layouts -> d_logits [B*N, C] (zero for padded regions).
 *   cvc_layernorm_cat_bwd : backward of cvc_layernorm_cat_fwd: d_out [rows, ld_out] -> dxs[s] [rows, lddx[s]] (nullable per input). */
int cvc_relu_dropout_fwd(const float* x, const float* bias, long long rows, int N, const uint32_t* rng_state, unsigned site,
                         float p, float* y, cvc_stream_t stream);
int cvc_relu_dropout_bwd(const float* dy, const float* y, long long n, const uint32_t* rng_state, unsigned site, float p, float* dx,
                         cvc_stream_t stream);
long long cvc_bn_workspace(long long rows, int C);
int cvc_bn_relu_train_fwd(const float* x, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, long long rows, int C, float* y, float* save_mean, float* save_invstd,
                          float* workspace, cvc_stream_t stream);
int cvc_bn_relu_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* save_mean,
                          const float* save_invstd, long long rows, int C, float* dx, float* dgamma, float* dbeta,
                          float* workspace, cvc_stream_t stream);
int cvc_class_softmax_bwd(const float* p_rows, const float* d_rows, const float* d_sim, const uint8_t* pad, int B, int N, int C,
                          float* d_logits, cvc_stream_t stream);
int cvc_layernorm_cat_bwd(const float* const* xs, const long long* ldx, const int* widths, int nseg, long long rows, float eps,
                          const float* d_out, long long ld_out, float* const* dxs, const long long* lddx, cvc_stream_t stream);

/* out[row, :] = scale * sum_n w[row, n] X[row / nq, n, :] (w [nclip * nq, n], X [nclip, n, R], nq >= 2): the several-queries
 * weighted sum of cvc_attn_wsum without its softmax -- d_q of dot-product attention (model/modules.py:24-76 under autograd) with
 * w = d_scores, X = proj_context, scale = 1 / temp.  CVC_E_TOOBIG when no group of queries fits the LDS. */
int cvc_attn_weighted_rows(const float* w, const float* X, int nclip, int nq, int n, int R, float scale, float* out, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Small helpers of the training step (deterministic; were library launches):
 * cvc_stable_order -- order[r] = index of the r-th key in a STABLE ascending sort of key[n] (n <= 7168): the row grouping of the
 *   embedding backward (torch.argsort(stable=True) in the host mirror);
 * cvc_col_sum -- out[c] (and out2[c] when given) = sum_s x[s * ld + c], s < S, c < n: bias gradients (nn.Linear / nn.LSTMCell
 *   bias_ih and bias_hh receive the same sum).  ws: cvc_col_sum_ws(S, n) floats (0 -> may be null): row chunks are summed by
 *   separate workgroups and combined by a second launch, fixed order. */
int cvc_stable_order(const int64_t* key, int n, int64_t* order, cvc_stream_t stream);
long long cvc_col_sum_ws(int S, int n);
int cvc_col_sum(const float* x, long long ld, int S, int n, float* out, float* out2, float* ws, cvc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CVC_HIP_BLOCKS_H */
