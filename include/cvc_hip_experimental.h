/*
 * cvc_hip_experimental.h -- forms that were built, measured and found not to pay ("Experiments that did not
 * pay"), kept selectable so that the measurements can be repeated: the grouped stream-K schedule of the decode step's skinny GEMMs
 * (csrc/gemm_gsk.hip), the K-split gate GEMM with its three finishes (csrc/gemm_packed_ks.hip), the one-launch vocabulary
 * projection + selection, and A/B switches.  They exist only in a library built with CVC_EXPERIMENTAL=1
 * (`CVC_EXPERIMENTAL=1 python cyclical-visual-captioning_amd/build_hip.py --force`); their tests carry the marker
 * `gpu_experimental`, not `gpu`.  Reached through cvc_block("name").  Contracts: the comments in cvc_hip.h ("Grouped stream-K
 * form", "K-split variant") describe them.
 */
#ifndef CVC_HIP_EXPERIMENTAL_H
#define CVC_HIP_EXPERIMENTAL_H
#include "cvc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

int cvc_gsk_plan(const int* ntile, const int* nchunk, int ngroups, int nwg, int* U, int* unit0, int* maxseg);
int cvc_gsk_gemm(const cvc_gsk_group* groups, int ngroups, int U, cvc_stream_t stream);   /* ngroups <= 3, M <= 64 rows */
int cvc_attn_scores_qslab(int kind, const cvc_gsk_segs* q, const float* q_bias, const float* w_a, const float* b_a,
                          float inv_temp, const cvc_attn_set* sets, int nsets, int nclip, int nq, int A,
                          cvc_stream_t stream);
int cvc_top2_slab(const cvc_gsk_segs* logits, const float* bias, int V, int M, int unk_idx, int64_t* word, int word_stride,
                  float* logprob, const float* table, int E, float* emb_out, int emb_ld, cvc_stream_t stream);
int cvc_packed_lstm_ks_slices(int K, int R);
int cvc_packed_lstm_ks_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                           const float* gate_bias, const float* c_prev_q, int M, int R, float* h_dst1_q,
                           float* h_dst2_q, float* c_out_q, float* slab, long long w_blk_stride, cvc_stream_t stream);
int cvc_packed_lstm_ksf_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                            const float* gate_bias, const float* c_prev_q, int M, int R, float* h_dst1_q,
                            float* h_dst2_q, float* c_out_q, float* slab, unsigned* counters, cvc_stream_t stream);
int cvc_packed_lstm_ksx_local(int on);   /* 1 (default): XCD-local exchange, XCC_ID-checked; 0: system-scope exchange; < 0 queries */
int cvc_packed_lstm_ksx_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                            const float* gate_bias, const float* emb_gate, const int64_t* word, const float* c_prev_q,
                            int M, int R, float* h_dst1_q, float* h_dst2_q, float* c_out_q, float* slab,
                            unsigned* flags, unsigned seq, cvc_stream_t stream);
int cvc_packed_lstm_wg_blocks(int n);
int cvc_packed_linear_select_fwd(const float* wp, const float* xq, int K, const float* bias, int M, int Nout,
                                 float* top2_part, unsigned* counter, int unk_idx, int64_t* word, int word_stride,
                                 float* logprob, cvc_stream_t stream);
int cvc_gru_persistent_halves(int on);   /* A/B + test hook: 1 = more than 32 clips run as two interleaved 32-clip recurrences (measured slower), 0 = default */

#ifdef __cplusplus
}
#endif
#endif /* CVC_HIP_EXPERIMENTAL_H */
