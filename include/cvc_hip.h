/*
 * cvc_hip.h -- C-ABI of libcvc_hip.so: the MI355X (gfx950) kernels behind the cyclical
 * visual-captioning decode / localize / reconstruct hot path.
 *
 * The reference has no FFI: its replaceable units are torch nn.Module classes
 * (SURVEY.md section 8(b)).  The host-side mirrors of those classes
 * (cyclical-visual-captioning_amd/cvc/model/) bind these entry points through ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.  Every entry point
 *   - takes raw DEVICE pointers, plain sizes and the HIP stream to launch on (no torch types),
 *   - allocates nothing, is re-entrant, and is graph-capturable (launches only),
 *   - expects fp32 row-major contiguous data, bool masks as one byte per element, word
 *     indices as int64,
 *   - returns 0 on success, a positive hipError_t from the launch, or a negative CVC_E_* for
 *     argument errors (the Python shim raises RuntimeError on non-zero).
 *
 * Paths below are relative to /root/reference/anet-video-captioning/.
 */
#ifndef CVC_HIP_H
#define CVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cvc_stream_t; /* hipStream_t */

/* The library is built with hidden visibility: only what this header declares CVC_API is exported (the drop-in ABI: what a
 * reference maintainer's stub binds, plus everything the host mirror's default path calls).  The per-kernel BUILDING BLOCKS the
 * whole-decode / whole-loop drivers are composed of are declared in cvc_hip_blocks.h: not exported, reachable for unit tests and
 * the host mirror's eager launch lists through cvc_block("name").  Forms that were measured and do not pay (kept selectable for
 * experiments) are in cvc_hip_experimental.h and only exist in a build made with CVC_EXPERIMENTAL=1. */
#define CVC_API __attribute__((visibility("default")))

#define CVC_E_BADARG (-1)   /* a size/alignment precondition is violated            */
#define CVC_E_TOOBIG (-2)   /* a dimension exceeds what the kernel templates cover  */
#define CVC_E_NORCCL (-3)   /* librccl.so could not be opened (cvc_comm_* / cvc_allreduce_grads only) */

#define CVC_ATTN_ADDITIVE 0 /* model/modules.py:100-159 AdditiveSoftAttention.forward */
#define CVC_ATTN_DOT 1      /* model/modules.py:24-76   SoftAttention.forward         */

/* library / build info: returns a static string "cvc_hip <version> gfx950[ +experimental]" */
CVC_API const char* cvc_version(void);
/* address of a building block (cvc_hip_blocks.h; with CVC_EXPERIMENTAL=1 also of cvc_hip_experimental.h) by name, or NULL */
CVC_API void* cvc_block(const char* name);

/* ---------------------------------------------------------------------------------------
 * Attention over one or two feature sets that share the query (regions + frames,
 * model/decoder_core.py:54-56, model/localizer_core.py:36-39).
 * rows = nclip * nq query rows; row r belongs to clip r / nq (nq > 1: beams of a clip or
 * the T localizer queries of a clip share the clip's features).
 */
typedef struct {
    const float* proj;          /* [nclip, n, A]   proj_context                                  */
    const float* ctx;           /* [nclip, n, R]   context (== proj with R == A when context=None) */
    const uint8_t* mask;        /* [nclip, n] 1 = masked (-1e8, modules.py:129) or NULL          */
    const uint8_t* frame_mask;  /* [rows, n]  proposal_frame_mask (modules.py:131-144) or NULL   */
    float* scores;              /* [rows, n]  out: masked PRE-softmax scores (workspace)         */
    float* frame_masked;        /* [rows, n]  out: pre-softmax copy filled at frame_mask, or NULL */
    float* attn;                /* [rows, n]  out: softmax over n                                */
    float* ctx_out;             /* [rows, R]  out: sum_n attn * ctx, or NULL                     */
    int n;
    int stream;                 /* cache policy of the feature reads: bit 0 = proj, bit 1 = ctx read with
                                 * non-temporal loads (a stream that should not displace what is re-read
                                 * every step from the 256 MB Infinity Cache); 0 = default, cacheable.
                                 * bit 2 = with_sentinel (modules.py:40-41, 123-124): masked positions are
                                 * filled with -inf instead of -1e8 (a fully masked row then softmaxes to
                                 * NaN, as in the reference)                                               */
} cvc_attn_set;

/* q [rows, A] is h2attn(h) (bias included).  kind ADDITIVE: s = w_a . tanh(proj_n + q) + b_a[0]
 * (b_a: DEVICE pointer to alpha_net.bias, nullable = 0; a host float would force a sync)
 * (no temperature, modules.py:120); kind DOT: s = (proj_n . q) * inv_temp (modules.py:34-37).
 * ctx_sum [rows, R] (nullable) receives the sum of the sets' contexts
 * (weighted_pool_feat + attn_conv, decoder_core.py:59).  Requires A % 4 == 0, R % 4 == 0. */
CVC_API int cvc_attn_fwd(int kind, const float* q, const float* w_a, const float* b_a, float inv_temp,
                 const cvc_attn_set* sets, int nsets, int nclip, int nq, int A, int R,
                 float* ctx_sum, cvc_stream_t stream);

/* The two passes of cvc_attn_fwd as separate entry points (cvc_attn_fwd == scores then wsum):
 * pass 1 streams proj [nclip,n,A] once and writes sets[].scores / frame_masked;
 * pass 2 softmaxes sets[].scores into sets[].attn and streams ctx [nclip,n,R] once. */
/* Pass 2 writing the summed context in the packed-GEMM activation layout [R/4][64][4] (rows <= 64) */
/* Pass 1 with the query given as q_nparts partial sums [q_nparts][rows, A] of a split-K h2attn GEMM
 * (cvc_linear_splitk_fwd) plus its bias q_bias [A] (nullable): the partials are summed while the
 * query is loaded into LDS, so the small query GEMM can spread over the whole chip. */

/* Backward of cvc_attn_fwd for ONE set, scores recomputed from proj (nothing but attn is
 * saved).  Inputs: d_ctx [rows,R] (nullable), d_fm [rows,n] gradient of the frame_masked
 * output (nullable).  Outputs: d_scores [rows,n] (gradient of the pre-softmax scores; its sum
 * is d_b_alpha); d_q [rows,A] (overwritten); d_w_part [rows,A] per-row partials of d_w_alpha
 * (additive only, nullable; the caller sums over rows -- keeps the reduction ordered);
 * d_proj [nclip,n,A] / d_ctxfeat [nclip,n,R] (ACCUMULATED into: caller zero-fills; nullable). */
CVC_API int cvc_attn_bwd(int kind, const float* q, const float* w_a, float inv_temp,
                 const float* proj, const float* ctx, const float* attn,
                 const float* d_ctx, const float* d_fm,
                 int nclip, int nq, int n, int A, int R,
                 float* d_scores, float* d_q, float* d_w_part,
                 float* d_proj, float* d_ctxfeat, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Skinny GEMM over a virtual concat of K-segments:  Y[m, n] = sum_s X_s[m, :] . W_s[n, :]
 * (torch.cat + nn.Linear / nn.LSTMCell input GEMMs, decoder_core.py:45-50,59-61,
 * captioner.py:266).  Weights stream once from HBM through the matrix cores: fp32 in, fp32
 * accumulate, products either on the fp32 MFMA or (default) as exact three-way bf16 splits on the
 * bf16 MFMA with fp32-grade error -- see cvc_gemm_packed_split; M <= 64.
 */
typedef struct {
    const float* x;      /* [M, k] leading dim ldx; with idx: table [*, k] read at row idx[m] */
    const int64_t* idx;  /* optional row gather (nn.Embedding, captioner.py:53-68) or NULL     */
    const float* w;      /* [Nout, >=k] leading dim ldw, pointing at this segment's first column */
    int k, ldx, ldw;
    int relu;            /* max(0, x) on load (the embedding's ReLU)                            */
} cvc_gemm_seg;

/* y[M, Nout] (ld ldy) = concat-GEMM + bias[Nout] (nullable) + bias2[Nout] (nullable) */
CVC_API int cvc_linear_fwd(const cvc_gemm_seg* segs, int nsegs, const float* bias, const float* bias2,
                   int M, int Nout, float* y, int ldy, cvc_stream_t stream);

/* Split-K variant for small Nout (h2attn: Nout = A): K slice s of every segment writes its partial
 * product to y_parts + s * M * Nout (slice 0 carries the bias); the consumer sums the slices.
 * M <= 64, no gather segments. */

/* Vocabulary projection with the word-selection partials fused into the epilogue
 * (captioner.py:437 + :415-422): every 32-column block writes, per row, {top-1 value, index, top-2
 * value, index, max, sum exp} to top2_part [ceil(Nout/32)][64][6]; y (nullable) receives the
 * logits only if asked.  cvc_top2_final merges the partials: UNK rule, word (int64, strided),
 * log-prob (nullable), and optionally next step's embedded word emb_out[row,:E] =
 * relu(table[word]) (captioner.py:424; emb_ld == 0 selects the quad layout [E/4][64][4]).  M <= 64. */

/* Packed path of the decode engine: both operands stored MFMA-fragment-native so that waves load
 * them straight into registers (no LDS) and keep CVC_PACKED_DEPTH x 4 KB of weights in flight each.
 *   wp : weights packed [ceil(Nout/32)][K/4][32][4] (LSTM: block b holds rows (i>>3)*R + 8b + (i&7));
 *        built once per checkpoint by cvc.decode.pack_weights -- the K-concat of the reference's
 *        torch.cat inputs is baked into the column order.
 *   xq : activations [K/4][64][4] ("quad" layout, 64 = padded batch rows), pointing at the first
 *        quad of this GEMM's K range; written by the producers (these kernels' epilogues,
 *        cvc_attn_wsum_quad, cvc_top2_final with emb_ld == 0).
 * K % 32 == 0, M <= 64.  LSTM: cell state c in quad layout [R/4][64][4]; h' goes to up to two quad
 * destinations (the next consumers' K ranges).  Linear: y row-major (split-K slices at stride
 * M*ldy, bias in slice 0) and/or the fused word-selection partials (see cvc_linear_top2_fwd). */
/* Embedding-gate table form of the att-LSTM (decoder_core.py:45-50 with xt = relu(Emb[word]), captioner.py:53-68 in eval mode):
 * the embedded word's share of the gates, W_ih[:, emb columns] x relu(Emb[v]), depends on the word alone, so it is tabulated once
 * per checkpoint -- emb_gate [V][4R] fp32, gates in checkpoint order (gate * R + unit; cvc.decode.embgate_table) -- and the
 * step adds row word[m] in the epilogue: wp / xq then cover only the recurrent inputs (K = 2R: h_lang, h_att), 20 % fewer weight
 * bytes per step, and the gate GEMM no longer waits for the word.  Otherwise cvc_packed_lstm_fwd. */
/* ... the general form.  w_cached != 0: the gate weights are read under the default cache policy (a gate matrix the caller's cache
 * plan keeps in the Infinity Cache between steps) instead of streamed non-temporally.  w_blk_stride / K: the contraction may stop
 * short of the packed matrix's K (K a multiple of 32; w_blk_stride = floats between its 32-row blocks, 0 = dense): the first
 * decode step multiplies an all-zero recurrent state, which adds nothing -- the decode driver passes K = 32 there.
 * Same results as cvc_packed_lstm_embgate_fwd over the same K. */
/* Vocabulary projection + word selection in ONE launch (captioner.py:437 + :415-422): cvc_packed_linear_fwd's top-2 records
 * (top2_part [ceil(Nout/32)][64][6], stored write-through) are merged by the last workgroup to arrive -- counter: one word of
 * device memory, zero before the first use, left zero -- which writes word[m * word_stride] (UNK rule, ties -> lowest index) and
 * logprob[m] (nullable).  Same results as cvc_packed_linear_fwd + cvc_top2_final; measured SLOWER than the two launches at
 * config 2 (34.9 vs 20.0 + 7.4 us: arrival atomics, the acquire fence and a serial merge on one CU cost more than a launch
 * boundary), so the decode drivers do not use it.  Kept as a tested entry point. */
/* A/B + test hook: 32-row weight blocks per workgroup of cvc_packed_lstm_fwd (1 = default; 2: two blocks share every
 * activation line through the CU's L1 -- halves the L2 reads, measured 60 % slower because half the CUs then do all the
 * operand splitting; same results up to the fp32 summation order over K).  Returns the previous setting; n < 1 only queries. */

/* Training form of cvc_packed_lstm_fwd (nn.LSTMCell forward under autograd, decoder_core.py:45-50, 59-61): the same GEMM
 * kernel with row-major state -- c_prev / h_out / c_out [M, R] and the activated gates [M, 4R] (i, f, g, o; nullable) that
 * cvc_lstm_pointwise_bwd reads.  Its two operands are rebuilt from the tensors autograd and the optimizer own:
 *   cvc_pack_lstm_weights : w_ih [4R, K_ih], w_hh [4R, K_hh] row-major (the checkpoint layout) -> wp of K = K_ih + K_hh
 *                           (once per optimizer step; K_ih, K_hh % 4 == 0, K % 32 == 0, R % 8 == 0);
 *   cvc_pack_quad_segs    : up to 6 row-major segments [M <= 64, width_s] (the virtual concat of the cell's inputs followed
 *                           by h_prev; widths % 4 == 0, 16-byte aligned) -> xq [K/4][64][4], rows beyond M zero. */
CVC_API int cvc_pack_quad_segs(const float* const* xs, const long long* ldx, const int* widths, int nseg, int M, float* xq,
                       cvc_stream_t stream);
/* Hoisted form of the training cell: input segments whose values are known for all T steps before the loop (the embedded
 * teacher-forced words, fc_feats, the localized context of the reconstruction loop) are multiplied ONCE for all T * B rows (a dense
 * product), and the per-step launch streams only the recurrent columns of the gate matrix:
 *   cvc_pack_lstm_segs : up to 4 column ranges [4R, width_s] (pointer at the first column, leading dimension ld_s) of row-major
 *                        weights -> wp over K = sum width_s (the recurrent ranges of weight_ih, then weight_hh);
 *   cvc_packed_lstm_train_pre_fwd : cvc_packed_lstm_train_fwd + gate_pre [M, 4R] (row-major, checkpoint gate order), the hoisted
 *                        ranges' contribution to this step's pre-activations. */
CVC_API int cvc_pack_lstm_segs(const float* const* ws, const long long* lds, const int* widths, int nseg, int R, float* wp,
                       cvc_stream_t stream);
/* ... with output = nn.Dropout(h') (decoder_core.py:62, 109) fused: h_drop_out [M, R] receives h' times the counter-based
 * keep-mask of element m * R + j (see cvc_dropout_rng); gate_pre nullable; h_out / h_out2 are the plain copies of h'. */

/* GRU over a whole sequence, one or both directions: the recurrent half of nn.GRU(batch_first, h0 = 0) as the encoder's frame
 * context uses it (backbone.py:103-106, 335-338; gate order r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),
 * h' = (1 - z) n + z h).  One launch of the packed GEMM kernel per time step serves both directions (direction 1 walks the
 * sequence backwards); the input projections of ALL steps come from one dense GEMM beforehand (cvc_tile_gemm):
 *   wp : [ndir][H/8][Kp/4][32][4]  W_hh packed like an LSTM gate matrix whose 4th gate is zero (block b = (r, z, n, 0) x
 *        hidden units 8b..8b+7), columns zero-padded to Kp = H rounded up to a multiple of 32;
 *   gi : x W_ih^T WITHOUT bias; row of (clip m, step t) at gi + m * gi_ld_m + t * gi_ld_t, columns [ndir][3H];
 *   b_ih, b_hh : [ndir][3H];  hq : workspace of 2 * ndir * Kp * 64 floats (the two parities of the state, quad layout);
 *   y  : h_t of direction d at y + m * y_ld_m + t * y_ld_t + d * H.
 * M <= 64 clips, H % 8 == 0, strides multiples of 4 floats. */
CVC_API int cvc_gru_seq_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih,
                    const float* b_hh, int M, int F, int H, int ndir, float* hq, float* y, long long y_ld_m,
                    long long y_ld_t, cvc_stream_t stream);
/* Persistent form (csrc/gru_persistent.hip): same operands and results from ONE cooperative launch -- every workgroup keeps its
 * rows of W_hh in registers for the whole sequence and the steps are separated by a barrier in device memory.  hq here holds one
 * state slot per step, (F + 1) * ndir * H * 64 floats (a slot is written once and read only after the barrier, so the XCDs'
 * L2s need no invalidation).  `sync`: cvc_gru_persistent_sync_words()
 * words of device memory (arrival counters spread over memory channels); word 4 is non-zero afterwards when the (bounded) barrier wait timed out: the outputs are then
 * invalid and the caller repeats the sequence with cvc_gru_seq_fwd.  Returns CVC_E_BADARG without launching for shapes
 * outside its range (H % 128 != 0, H > 1024, or more workgroups than the device keeps resident at once). */
CVC_API int cvc_gru_persistent_sync_words(void);
/* Training form of the persistent recurrence: additionally writes, for every step and direction, what autograd needs --
 * (r, z, n, W_hn h + b_hn) at gates + m * g_ld_m + t * g_ld_t + d * 4H + {0, H, 2H, 3H}. */
CVC_API int cvc_gru_seq_persistent_train_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t,
                                     const float* b_ih, const float* b_hh, int M, int F, int H, int ndir, float* hq,
                                     float* y, long long y_ld_m, long long y_ld_t, float* gates, long long g_ld_m,
                                     long long g_ld_t, unsigned* sync, cvc_stream_t stream);
/* Backward of the recurrence (autograd of nn.GRU, backbone.py:103-106, 335-338), walking the sequence backwards: per step and
 * direction the gate arithmetic and dgh_t W_hh (cvc_linear_nn_fwd on the checkpoint-layout weights).  dy / y / gates: row of
 * (clip m, step t) at base + m * ld_m + t * ld_t, columns [ndir][H] / [ndir][4][H]; w_hh [ndir][3H, H] row-major.
 * Outputs dgi, dgh: [F * M rows (t * M + m), ndir * 3H] -- pre-activation gradients of the input / hidden side, from which the
 * caller takes dW_ih, dX, dW_hh and the biases in dense GEMMs over all steps.  work: ndir * (2 M H + 192 H + ksplit * M *
 * ceil(H/128) * 128) floats, ksplit = cvc_gru_seq_bwd_ksplit(H).  M <= 64, H % 8 == 0. */
/* Persistent form of cvc_gru_seq_bwd (csrc/gru_bwd_persistent.hip): one cooperative launch for the whole sequence, W_hh columns
 * in registers, dgh exchanged through per-step slots.  wt = W_hh^T packed [ndir][H/8][3H/8][8 units][8 k]
 * (cvc.gru.pack_gru_weights_t), slots = F * ndir * 3H * 64 floats, sync = cvc_gru_bwd_persistent_sync_words() words (word 4
 * non-zero afterwards = barrier time-out: outputs invalid, repeat with cvc_gru_seq_bwd).  H % 256 == 0, H <= 1024, M <= 64;
 * CVC_E_BADARG without launching otherwise. */
CVC_API int cvc_gru_bwd_persistent_sync_words(void);
CVC_API int cvc_gru_seq_bwd_persistent(const float* dy, long long dy_ld_m, long long dy_ld_t, const float* gates, long long g_ld_m,
                               long long g_ld_t, const float* y, long long y_ld_m, long long y_ld_t, const float* wt,
                               int M, int F, int H, int ndir, float* dgi, float* dgh, float* slots, unsigned* sync,
                               cvc_stream_t stream);
CVC_API int cvc_gru_seq_bwd_ksplit(int H);
CVC_API int cvc_gru_seq_bwd(const float* dy, long long dy_ld_m, long long dy_ld_t, const float* gates, long long g_ld_m,
                    long long g_ld_t, const float* y, long long y_ld_m, long long y_ld_t, const float* w_hh, int M,
                    int F, int H, int ndir, float* dgi, float* dgh, float* work, cvc_stream_t stream);
CVC_API int cvc_gru_seq_persistent_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t,
                               const float* b_ih, const float* b_hh, int M, int F, int H, int ndir, float* hq,
                               float* y, long long y_ld_m, long long y_ld_t, unsigned* sync, cvc_stream_t stream);

/* K-split variant of cvc_packed_lstm_fwd (same operands, same arithmetic): one workgroup = 256 gate rows x K / S, the 32-k
 * activation chunks fetched and split once per workgroup and shared through LDS (cuts the L2 activation reads of the full-K
 * kernel from 2 x the weight bytes to 1/4 of them); partial tiles go to `slab` (>= S * (R/8) * 2048 floats, S =
 * cvc_packed_lstm_ks_slices(K, R)) and a finishing launch sums them in slice order, adds the biases and does the cell update.
 * w_blk_stride: floats between consecutive 32-row blocks of wp (0 = dense, K/4 * 128): a pack whose blocks are 4 KB further
 * apart than dense keeps the 8 waves of a workgroup, which stream the same chunk index of 8 different blocks, off the same
 * HBM channels.
 * cvc_packed_lstm_ks_slices returns 0 when the shape is not covered (R % 64 != 0): use cvc_packed_lstm_fwd then. */
/* The same with the finish fused into the GEMM launch: partial tiles are stored write-through, the K slices of a 256-row tile
 * count their arrivals in counters[tile] and the LAST one sums the tile's slabs in slice order and does the cell update
 * (deterministic: the order of the sum does not depend on who arrives last).  counters: R / 64 words of device memory, zero
 * before the first use; every launch leaves them zero.  Dense weight pack (no block stagger). */
/* ... with the finish shared by ALL K slices of a tile (exchange finish): slice ks of a 256-row tile finishes the tile's block ks
 * once the tile's 8 partial tiles are out (write-through slab rows + one arrival word per slice, system-scope loads on the
 * reading side: correct wherever the workgroups run; tiles are placed on one XCD for speed only).  One launch, no finishing
 * launch.  flags: R / 8 + 1 words of device memory, zero before the first use ([R / 64][8] arrival words + the error word at
 * [R / 8], set to 1 by a slice whose bounded wait ran out); seq: non-zero and different from the previous launch's on these
 * flags; emb_gate / word: the embedding-gate form (cvc_packed_lstm_embgate_fwd), nullable.  R = 2048 (8 slices, 256 workgroups
 * of 512 threads, all resident together). */

/* ---------------------------------------------------------------------------------------
 * Grouped stream-K form of the decode step's skinny GEMMs (csrc/gemm_gsk.hip).  The gate GEMM of an LSTM cell
 * (decoder_core.py:50, 61) is cut along K into an EARLY part -- the K segments whose inputs exist before the step's critical
 * path reaches the cell (att-LSTM: h_lang(t-1), h_att(t-1); lang-LSTM: h_att(t), h_lang(t-1)) -- and a LATE part (the embedded
 * word; the attended context).  The early part of the NEXT cell runs in the same launch as the small GEMM that sits on the
 * critical path at that moment (vocabulary logits, captioner.py:437; h2attn, modules.py:112): one launch = up to three GEMMs
 * ("groups") that read the same kind of operands, flattened into one space of UNITS (256 weight rows x one 32-k chunk) that is
 * dealt out evenly, U units per workgroup, so that every CU streams the same number of weight bytes whatever the shapes are.
 * A workgroup's run of units inside one 256-row tile is a SEGMENT; its partial product goes to the group's slab
 *     slab[tile][seg][8 blocks][64 rows][32 gate rows]  (fp32),   seg = workgroup - first workgroup of the tile,
 * and the consumer (cvc_packed_lstm_late_fwd, cvc_attn_scores_qslab, cvc_top2_slab) sums a tile's segments in segment order:
 * results do not depend on scheduling.  Operand layouts are those of the packed path (wp, xq above); the 32-k activation chunk
 * is fetched and split into bf16 terms once per workgroup and shared by its 8 waves through LDS. */
typedef struct cvc_gsk_group {
    const float* wp;          /* packed weights [nblk][*][32][4], pointing at physical chunk 0 of this GEMM's K space */
    long long w_blk_stride;   /* floats between consecutive 32-row blocks of wp                                       */
    const float* xq;          /* quad-layout activations, physical chunk 0                                            */
    int nblk;                 /* valid 32-row blocks; tiles of 8 blocks (a short last tile is allowed)                */
    int nchunk;               /* 32-k chunks per tile walked by this launch, indexed v = 0 .. nchunk-1                */
    int skip_at, skip_n;      /* physical chunk of v: v < skip_at ? v : v + skip_n  (steps over the late K segment)   */
    float* slab;              /* >= ceil(nblk/8) * maxseg * 16384 floats                                              */
    int maxseg;               /* segments allocated per tile (cvc_gsk_plan)                                           */
} cvc_gsk_group;
typedef struct cvc_gsk_segs { /* what a consumer needs to find and sum one group's partial tiles                       */
    const float* slab;
    int unit0;                /* first unit of the group in the launch's unit space                                   */
    int nchunk;               /* units per tile                                                                       */
    int U;                    /* units per workgroup                                                                  */
    int maxseg;
} cvc_gsk_segs;
/* Host arithmetic of a launch: ntile[g] tiles of nchunk[g] chunks each -> U (units per workgroup for at most nwg workgroups;
 * nwg <= 0: one per CU of the current device), unit0[g] and the segments per tile maxseg[g] the slabs must hold. */
/* LATE part of a cell + finish: gates = wp[:, K range] x xq (full-K packed kernel, K % 32 == 0) + the early part's partial tiles
 * (`early` nullable: none, e.g. step 0 from the zero state) + biases; cell update as cvc_packed_lstm_fwd. */
/* cvc_attn_scores_qparts with the query given as the partial tiles of a stream-K h2attn group (A / 32 blocks) */
/* cvc_top2_final with the logits given as the partial tiles of a stream-K vocabulary group (ceil(V/32) blocks) + bias [V]:
 * sums the segments, top-2 with the UNK rule, log-prob, next step's embedded word (captioner.py:415-424, 437).  M <= 64. */

/* Packed path arithmetic.  mode 2 (default) / 1: every fp32 operand is split exactly into three bf16 terms
 * (v = hi + mid + lo) and each product taken as its six leading cross terms on the bf16 MFMA, fp32
 * accumulate -- the dropped terms are < 2^-23 relative, the level of one fp32 rounding (measured error
 * against fp64 is slightly BELOW the fp32-MFMA path's); 8 (mode 2) or 4 (mode 1) waves per workgroup.
 * mode 0: plain fp32 MFMA.  A negative mode only queries.  Returns the previous mode. */
CVC_API int cvc_gemm_packed_split(int mode);

/* Test hook: route every concat-GEMM to the generic direct-load kernel (on != 0) instead of the
 * LDS-DMA fast path that is taken when all segment widths are multiples of 128.  Returns the
 * previous setting.  Both kernels compute the same sums in different k-orders. */

/* nn.LSTMCell (gate order i,f,g,o; decoder_core.py:14,27,50,61): gates = concat-GEMM with
 * Nout = 4R + b_ih + b_hh; c' = sig(f) c + sig(i) tanh(g); h' = sig(o) tanh(c').
 * c_prev/h_out/c_out are [M, R] contiguous.  gates_out [M, 4R] (nullable) receives the
 * ACTIVATED gates (i,f,g,o) for the backward pass.  b_ih / b_hh are nullable; gate_bias
 * [M, 4R] (nullable) is a per-row additive pre-activation term: the decode loop hoists the
 * step-invariant part of the gate GEMM (fc_feats x W_ih[:, R:2R] + biases, decoder_core.py:46)
 * out of the T loop and passes it here.  Requires R % 8 == 0. */
CVC_API int cvc_lstm_cell_fwd(const cvc_gemm_seg* segs, int nsegs, const float* b_ih, const float* b_hh,
                      const float* gate_bias, const float* c_prev, int M, int R, float* h_out,
                      float* c_out, float* gates_out, cvc_stream_t stream);

/* LSTM pointwise backward: from d_h, d_c (nullable = 0), saved activated gates [M,4R],
 * c_prev, c_new -> d_gates [M,4R] (pre-activation) and d_c_prev [M,R].  d_gates_q (nullable)
 * receives a second copy of d_gates in the quad layout [4R/4][64][4] that cvc_linear_nn_fwd reads
 * (M <= 64, R % 4 == 0 when given). */
/* the same with the gradients of up to three copies of h' (h has several consumers per step -- the other cell, the attention
 * query, the next step -- and a copy per consumer leaves autograd nothing to accumulate: 160 small launches per training step) */
/* ... where the third copy left the cell through the fused dropout of cvc_packed_lstm_train_drop_fwd: d_h3 takes that mask */

/* Backward-data product of the skinny layers, autograd of nn.LSTMCell / nn.Linear
 * (decoder_core.py:45-50, 59-61, 99-108):  dst_s[M, ncols_s] = dY[M, K] x W_s[K, ncols_s]  for up to 6
 * column ranges s of row-major weights (the checkpoint layout, K = output rows of the layer), one
 * launch.  dy_q is dY in the quad layout [K/4][64][4]; M <= 64, K % 8 == 0, ncols / ldw / ld_dst
 * multiples of 4, pointers 16-byte aligned.  ksplit > 1 splits K over workgroups: workspace must
 * hold ksplit * M * sum_s(ceil(ncols_s / 128) * 128) floats; the planes are summed in a fixed order. */
typedef struct cvc_nn_seg {
    const float* w;      /* first column of the range inside a row-major [K, ldw] matrix */
    float* dst;          /* [M, ld_dst] */
    int ldw, ncols, ld_dst;
} cvc_nn_seg;
CVC_API int cvc_linear_nn_fwd(const float* dy_q, int K, int M, const cvc_nn_seg* segs, int nsegs, int ksplit,
                      float* workspace, cvc_stream_t stream);
/* the same without the summing launch: for ksplit > 1 the K-slice planes [ksplit][M][ntot] stay in `workspace` for a consumer
 * that sums them itself (ntot = sum_s ceil(ncols_s / 128) * 128) */
/* x [M <= 64, K] row-major (leading dim ldx) -> the quad layout [K/4][64][4] cvc_linear_nn_fwd reads (rows beyond M zero) */
CVC_API int cvc_pack_quad(const float* x, long long ldx, int M, int K, float* xq, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Word embedding and vocabulary head (captioner.py:53-68, 72-76, 266, 415-422).
 */
/* out[m, :] = relu(table[idx[m], :]) * (drop ? drop[m, :] : 1) */
CVC_API int cvc_embed_relu_fwd(const float* table, const int64_t* idx, const float* drop, int M, int E,
                       float* out, cvc_stream_t stream);
/* d_table[w, :] = (table[w, :] > 0) * sum_{m: idx[m] == w} drop[m, :] * d_out[m, :] for every word w that
 * occurs; no atomics and no host round trip: `order` [M] is a stable argsort of idx, so the rows of a word
 * are contiguous in it; they are summed in a fixed two-level order (16-row pieces, then the pieces of a
 * word), so a word that owns many rows (BOS / padding) does not serialise.  workspace: M * E floats.
 * The rows of the words that occur are ACCUMULATED into d_table (the caller zero-fills for a fresh gradient; several lookups of one
 * table -- the three embeddings of a cyclical pass -- add up in one buffer, call after call); other rows are left untouched. */
CVC_API int cvc_embed_relu_bwd(const float* table, const int64_t* idx, const int64_t* order, const float* drop,
                       const float* d_out, int M, int E, float* d_table, float* workspace,
                       cvc_stream_t stream);

/* Counter-based dropout (nn.Dropout of the training pass: captioner.py:53-68, decoder_core.py:62, 109, backbone.py:55-57) with the
 * keep-masks generated INSIDE the consuming kernels -- no mask tensors, no library RNG launches:
 *   keep(element) = hash(seed, step, site, flat element index) >= p * 2^32,   multiplier = keep ? 1 / (1 - p) : 0
 * (csrc/dropout_rng.h; cvc/synth.py::dropout_keep restates the hash on the host for the train-mode parity tests).
 * rng_state: 3 words of DEVICE memory {seed_lo, seed_hi, step}; the host advances `step` with a device-side add once per
 * training step, so a HIP-graph replay of the step draws fresh masks.  site: which dropout of the pass (cvc/dropout.py).
 *   cvc_embed_relu_rng_fwd / _bwd : cvc_embed_relu_fwd / _bwd with the mask of element (m, e) at flat index m * E + e;
 *   cvc_dropout_rng               : y[i] = x[i] * multiplier(i), n elements (16-byte aligned), for sites no producer fuses
 *                                   (its own backward: the same call on the gradient). */
CVC_API int cvc_embed_relu_rng_fwd(const float* table, const int64_t* idx, const uint32_t* rng_state, unsigned site, float p, int M,
                           int E, float* out, cvc_stream_t stream);
CVC_API int cvc_embed_relu_rng_bwd(const float* table, const int64_t* idx, const int64_t* order, const uint32_t* rng_state,
                           unsigned site, float p, const float* d_out, int M, int E, float* d_table, float* workspace,
                           cvc_stream_t stream);
CVC_API int cvc_dropout_rng(const float* x, long long n, const uint32_t* rng_state, unsigned site, float p, float* y,
                    cvc_stream_t stream);

/* In-place-capable row log-softmax: logp[m, :] = logits[m, :] - logsumexp(logits[m, :]) */
CVC_API int cvc_log_softmax_fwd(const float* logits, int M, int V, float* logp, cvc_stream_t stream);

/* d_logits = d_logp - exp(logp) * rowsum(d_logp) */
CVC_API int cvc_log_softmax_bwd(const float* logp, const float* d_logp, int M, int V, float* d_logits,
                        cvc_stream_t stream);

/* Greedy word selection with UNK suppression (captioner.py:415-422): per row the top-2 of
 * logits; pick #2 iff #1 == unk_idx; ties -> lowest index.  word[m*word_stride] (int64),
 * logprob[m] = logit - logsumexp (nullable). */
CVC_API int cvc_top2_unk(const float* logits, int M, int V, int unk_idx, int64_t* word, int word_stride,
                 float* logprob, cvc_stream_t stream);

/* Masked NLL over a [M, V] log-prob matrix (misc/utils.py:132-146, 181-192), the module-level
 * criterion API that receives log-probs:  loss_sum[0] += sum_m w[m] * -logp[m, target[m]]. */
CVC_API int cvc_nll_fwd(const float* logp, const int64_t* target, const float* w, int M, int V,
                float* loss_sum, cvc_stream_t stream);
/* d_logp[m, v] = -w[m] * g[0] at v == target[m], else 0 (g: device scalar upstream gradient) */
CVC_API int cvc_nll_bwd(const int64_t* target, const float* w, const float* g, int M, int V, float* d_logp,
                cvc_stream_t stream);

/* Fused vocabulary criterion (captioner.py:266 + :313 + misc/utils.py:132-146, 181-192) straight from raw
 * logits [M, V], no log-prob matrix: lse[m] = logsumexp(logits[m]); argmax[m] (nullable; ties -> lowest
 * index) for the cycle's argmax cut; row_loss[m] = w[m] * (lse[m] - logits[m, target[m]]);
 * loss_sum[0] = sum_m row_loss[m] in a fixed order.  Backward:
 * d_logits[m, v] = g[0] * w[m] * (exp(logits[m, v] - lse[m]) - [v == target[m]])  (g: device scalar). */
CVC_API int cvc_vocab_nll_fwd(const float* logits, const int64_t* target, const float* w, int M, int V,
                      float* lse, int64_t* argmax, float* row_loss, float* loss_sum, cvc_stream_t stream);
CVC_API int cvc_vocab_nll_bwd(const float* logits, const float* lse, const int64_t* target, const float* w,
                      const float* g, int M, int V, float* d_logits, cvc_stream_t stream);

/* The same criterion folded into the finishing pass of the vocabulary head's tile GEMM (cvc_tile_gemm): parts = its K-slice slabs
 * [nparts][M, ld] (slab stride part_stride), bias [V] (nullable).  Writes pre[m, v] = w[m] * (softmax(logits[m])[v] - [v ==
 * target[m]]) (leading dimension ld_pre; may alias slab 0), argmax (nullable), row_loss and loss_sum as cvc_vocab_nll_fwd -- the
 * [M, V] logits are never written.  Backward: d_logits = g[0] * pre (cvc_scale_by_scalar: y[i] = g[0] * x[i], n % 4 == 0).
 * V <= 8192. */
CVC_API int cvc_vocab_head_nll_fwd(const float* parts, int nparts, long long part_stride, int ld, const float* bias,
                           const int64_t* target, const float* w, int M, int V, float* pre, int ld_pre, int64_t* argmax,
                           float* row_loss, float* loss_sum, cvc_stream_t stream);
CVC_API int cvc_scale_by_scalar(const float* x, const float* g, long long n, float* y, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Grounder (captioner.py:132-173, dot-product branch): out[b,t,n] = xt[b,t,:] . feats[b,n,:]
 * + bias[b,t,n], filled with -1e8 where mask[b,t,n].
 */
CVC_API int cvc_grounder_fwd(const float* xt, const float* feats, const float* bias, const uint8_t* mask,
                     int B, int T, int N, int G, float* out, cvc_stream_t stream);
/* its backward (autograd of captioner.py:160-171): d [B,T,N] is the upstream gradient with masked slots already zero;
 * d_xt[b,t,:] = sum_n d[b,t,n] feats[b,n,:], d_feats[b,n,:] = sum_t d[b,t,n] xt[b,t,:]; either output may be null. */
CVC_API int cvc_grounder_bwd(const float* d, const float* xt, const float* feats, int B, int T, int N, int G,
                     float* d_xt, float* d_feats, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Label glue and the supervised attention criteria, all T steps per launch.
 *   cvc_bbox_overlaps_fwd : misc/utils.py:335-338 -> misc/bbox_transform.py:224-272.  ov[b, n, k] = IoU(+1-pixel convention) of
 *       proposal n (rois [B, N, ld_roi >= 4]: x1, y1, x2, y2, ...) and ground-truth box k (gt [B, K, ld_gt >= 4]); times 0 where
 *       frm_mask[b, n, k] or pnt_mask[b, n] (uint8, nullable, row stride ld_pnt) is set; 0 for degenerate (1 x 1) ground-truth
 *       boxes, -1 for degenerate proposals.  Evaluated in the reference's operation order, each operation rounded on its own.
 *   cvc_label_glue_fwd : misc/utils.py:351-373 + model/captioner.py:246-260 for t = 0 .. T-1 at once.
 *       box_mask element (b, k, t) at b * bm_stride_b + k * bm_stride_k + t * bm_stride_t (a view of mask_boxes[:, 0, :, 1:T+1]);
 *       labels[b, t, n] = max_k (box_mask ? 0 : ov[b, n, k]) > 0.5;
 *       frm_mask_output[b, t, 0] = pnt_mask[b, 0], [b, t, 1 + n] = (for every k: box_mask | frm_mask[b, n, k]) | pnt_mask[b, 1 + n]
 *       (pnt_mask [B, N + 1] uint8); step_fmask [T, B, N] (nullable) = frm_mask_output[:, :, 1:] with the step leading (what the
 *       decode loop's attention takes).  bool / integer results: bit-exact.
 *   cvc_attn_nll_fwd / _bwd : misc/utils.py:150-162 for one or two score tensors x [B, T, N] (element (b, t, n) at
 *       b * stride_b + t * stride_t + n; att2_weights and ground_weights share the target):
 *       loss[i] = -sum(log_softmax(x_i, 2) * target) / max(sum(target), 1); workspace: 5 * B * T + 1 floats, kept for the backward;
 *       d_x_i[b, t, n] (contiguous) = g_i[0] / count * (softmax(x_i)[n] * targets_in_row - target[n])  (g_i: device scalars). */
CVC_API int cvc_bbox_overlaps_fwd(const float* rois, int ld_roi, const float* gt, int ld_gt, const uint8_t* frm_mask,
                          const uint8_t* pnt_mask, int ld_pnt, int B, int N, int K, float* ov, cvc_stream_t stream);
CVC_API int cvc_label_glue_fwd(const float* ov, const uint8_t* box_mask, long long bm_stride_b, long long bm_stride_k,
                       long long bm_stride_t, const uint8_t* frm_mask, const uint8_t* pnt_mask, int B, int N, int K, int T,
                       uint8_t* labels, uint8_t* frm_mask_output, uint8_t* step_fmask, cvc_stream_t stream);
CVC_API int cvc_attn_nll_fwd(const float* x0, long long x0_stride_b, long long x0_stride_t, const float* x1, long long x1_stride_b,
                     long long x1_stride_t, const uint8_t* target, int B, int T, int N, float* workspace, float* loss,
                     cvc_stream_t stream);
CVC_API int cvc_attn_nll_bwd(const float* x0, long long x0_stride_b, long long x0_stride_t, const float* x1, long long x1_stride_b,
                     long long x1_stride_t, const uint8_t* target, int B, int T, int N, const float* workspace,
                     const float* g0, const float* g1, float* d_x0, float* d_x1, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Beam bookkeeping (build-defined, SURVEY.md section 7 "Beam-search specification"):
 * per clip select the `beam` best of beam*V candidates score[b,k] + logp[b,k,v] with
 * logp[unk] = -inf, finished hypotheses frozen (only v = 0 at +0); beam <= 8, beam < V <= 8192.
 * Two stages: per-row top-`beam` (one workgroup per hypothesis), then a per-clip merge.  Outputs parent[b,k],
 * word[b,k] (int64), new score[b,k]; ties -> lowest flat (k, v) index.
 */
CVC_API int cvc_beam_select(const float* logits, const float* score_in, const uint8_t* done_in,
                    int B, int beam, int V, int unk_idx, int first_step,
                    int64_t* parent, int64_t* word, float* score_out, uint8_t* done_out,
                    float* workspace /* >= 17 * B * beam floats */, cvc_stream_t stream);
/* the same with the logits given as the K-slice slabs of the vocabulary GEMM, parts[s][rows, V] (slab stride part_stride floats)
 * plus bias[V] (nullable): summed in slab order while a row is loaded, so the tile path never writes the logits matrix */
/* Best hypothesis of every clip after a beam decode: words [T, B*beam] (the word chosen for row r at step t), parent
 * [T, B*beam] (its parent beam slot), att [T, B*beam, N] (region attention of the step, computed for the parent row) ->
 * seq [B, T], att_out [B, T, N] of the rank-0 hypothesis (build-defined beam rule).  T <= 256. */
CVC_API int cvc_beam_backtrack(const int64_t* words, const int64_t* parent, const float* att, int B, int beam, int T, int N,
                       int64_t* seq, float* att_out, cvc_stream_t stream);

/* dst[r, :] = src[(r / beam) * beam + parent[r], :] for r in [0, rows)  (state reorder) */
CVC_API int cvc_gather_rows(const float* src, const int64_t* parent, int rows, int beam, int width,
                    float* dst, cvc_stream_t stream);


/* ---------------------------------------------------------------------------------------
 * Tile path of the decode engine: more than 64 live rows per step (beam search: rows = clips x beam, model/captioner.py
 * :410-438 with the build-defined beam rule of SURVEY.md section 7; greedy batches beyond 64 clips).
 * Both GEMM operands live in HBM as "fragments": 32 rows x 16 k of ONE of the three bf16 terms of the split-product
 * arithmetic (v = hi + mid + lo exactly, see cvc_gemm_packed_split), 1 KiB, ordered [k half][row][8 k] -- the lane order
 * of the bf16 MFMA operand.  A block of 32 rows stores, per k step of 16, its three terms back to back (3 KiB):
 *   wb : weights      [ceil(N/128)*4 blocks][K/16][3][512 bf16]  (zero rows beyond N; LSTM gate matrices in the packed row
 *                     order of cvc_packed_lstm_fwd: block b = 4 gates x hidden units 8b..8b+7), packed once per checkpoint
 *                     by cvc.decode.pack_weights_tile;
 *   xb : activations  [cvc_tile_rows_alloc(M)/32 blocks][k steps][3][512], x_mblk_stride = bf16 elements between row
 *                     blocks; a GEMM over a K segment of a wider buffer just points at the segment's first k step.
 *                     Written by the producers below (and cvc_attn_wsum_frag); rows beyond M must be zero-initialised.
 * cvc_tile_gemm: parts[s][m, n] (row-major, leading dim ld, slab stride part_stride floats) = partial product over K slice s
 * of ksplit; the consumer sums the slabs in slab order.  K % 16 == 0; any M (walked in chunks of 320 rows); any N.
 */
/* A/B + test hook, returns the previous setting (< 0 only queries): 0 = every wave issues its share of the LDS-DMA copies,
 * 1 = dedicated loader waves + 8 computing waves, 2 = dedicated loader waves + 4 wide computing waves (2 weight blocks each),
 * 3 (default) = form 2 for long K loops (>= 64 k steps per workgroup), form 1 otherwise.  All forms: identical results. */
CVC_API int cvc_tile_rows_alloc(int M);      /* rows (a multiple of 32) an activation fragment buffer for M live rows must hold */
CVC_API int cvc_tile_gemm(const void* wb, const void* xb, long long x_mblk_stride, int K, int M, int N, int ksplit,
                  float* parts, int ld, long long part_stride, cvc_stream_t stream);
/* nn.LSTMCell epilogue over the slabs of a gate GEMM (packed feature order): + b_ih + b_hh + gate_bias[m / gb_div]
 * (checkpoint order [*, 4R]; the hoisted fc term, one row per clip), cell update with c_prev [M, R]; writes c_out, h_out
 * [M, R] (h_out nullable) and h' as activation fragments at up to two destinations (pointer at the segment's first k step). */
/* the same with the embedded word's share of the gates taken from the embedding-gate table (see cvc_packed_lstm_embgate_fwd):
 * + emb_gate[word[m], :] (checkpoint order [V, 4R]; a word outside [0, V) reads row 0).  The gate GEMM then covers K = 2R
 * (h_lang | h_att) and cvc_tile_reorder_pack is called with E = 0. */
/* y[m, n] = sum_s parts[s][m, n] + bias[n] + bias2[n]  (nn.Linear epilogue: vocabulary logits, hoisted fc gate term) */
CVC_API int cvc_tile_linear_finish(const float* parts, int nparts, long long part_stride, int ld, const float* bias,
                           const float* bias2, int M, int N, float* y, int ldy, cvc_stream_t stream);
/* fp32 rows (optionally gathered through idx, optionally ReLU'd) -> activation fragments */
CVC_API int cvc_tile_pack_rows(const float* x, int ldx, const int64_t* idx, int relu, int M, int K, void* xb,
                       long long x_mblk_stride, cvc_stream_t stream);
/* Operand packers for the dense backward products (autograd of nn.Linear / nn.LSTMCell: dW = dY^T X batched over all T steps,
 * dX = dY W of the vocabulary head; reference decoder_core.py:50,61, captioner.py:266,361): any sizes, zero fill.
 * cvc_tile_pack_rows_any: x [M, K] row-major -> fragments with rows m, contraction k (k padded to a multiple of 16).
 * cvc_tile_pack_cols:     x [S, C] row-major read as its transpose -> fragments with rows c, contraction s. */
CVC_API int cvc_tile_pack_rows_any(const float* x, long long ldx, int M, int K, void* xb, long long x_mblk_stride, cvc_stream_t stream);
CVC_API int cvc_tile_pack_cols(const float* x, long long ldx, int S, int C, void* xb, long long x_mblk_stride, cvc_stream_t stream);
/* Beam-state reorder fused with next step's operand packing: row r continues hypothesis (r / beam) * beam + parent[r]
 * (parent NULL: r).  c_*_prev[r] = c_*[src]; xa = [h_lang[src] | relu(table[word[r]]) | h_att[src]] as fragments (K = 2R + E,
 * decoder_core.py:45-48 without the hoisted fc segment; E = 0: no embedding segment, table may be NULL -- the embedding-gate
 * form); xl_hlang (lang-LSTM input, third K segment) = h_lang[src]. */
/* Pass 2 of the attention writing the summed context as activation fragments (rows of the tile path) */

/* ---------------------------------------------------------------------------------------
 * Once-per-clip encoder, inference (model/backbone.py:189-351): the small fused pieces between its
 * dense products (cvc_tile_gemm) and its frame-context GRU (cvc_gru_seq_*).
 */
/* Class similarity (backbone.py:222-235): logits [B*N, ld_logits] = region features x class table^T (a cvc_tile_gemm product);
 * + bias[C] (nullable); regions with pad[b, n] != 0 filled with -1e8 in every class; softmax over the C classes;
 * out [B, C, N] (the reference's layout) and, if asked, out_rows [B*N, C] (what the region features concatenate, :274-277). */
CVC_API int cvc_class_softmax_fwd(const float* logits, long long ld_logits, const float* bias, const uint8_t* pad, int B, int N,
                          int C, float* out, float* out_rows, cvc_stream_t stream);
/* out[row, :] = [ F.layer_norm(x_0[row], [d_0]) | ... ] for up to three inputs (no affine, biased variance, eps):
 * backbone.py:215-216 (fc | seg_info) and :274-277 (region | location | class-probability features). */
CVC_API int cvc_layernorm_cat_fwd(const float* const* xs, const long long* ldx, const int* widths, int nseg, long long rows, float eps,
                          float* out, long long ld_out, cvc_stream_t stream);
/* Frame embeddings (backbone.py:325-333) after their two dense products y0 [rows, c0], y1 [rows, c1] (bias-free):
 * out [rows, c0 + c1] = relu( scale * cat(relu(y0 + b0), relu(y1 + b1)) + shift ), scale / shift = BatchNorm1d in eval mode
 * folded per channel (gamma / sqrt(running_var + eps), beta - running_mean * scale).  c0, c1 % 4 == 0. */
CVC_API int cvc_frame_embed_fwd(const float* y0, const float* b0, int c0, const float* y1, const float* b1, int c1, const float* scale,
                        const float* shift, long long rows, float* out, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Whole-decode drivers: the T-step sampler of model/captioner.py:384-443 (`_sample`: exactly T decoder steps from BOS,
 * top-2 with UNK suppression, no EOS early exit) and its beam-search counterpart (build-defined) as ONE
 * host call that enqueues every launch of the decode on the caller's stream -- no Python in the loop, no allocation, no host
 * synchronisation, capturable into a HIP graph.  The caller binds device buffers once in a cvc_decode_desc (weights in the
 * packed / fragment layouts above, the clip features, state and output buffers it allocated) and creates a plan; the plan only
 * copies the descriptor (host memory).  Greedy with <= 64 rows runs the packed path (cvc_packed_*), everything else the tile
 * path (cvc_tile_*); which fields a path reads is noted per field.  cvc.decode.DecodeEngine is the reference user.
 */
typedef struct cvc_decode_desc {
    /* dimensions */
    int B, beam, T, N, F, R, A, E, V;     /* clips, beams per clip (1 = greedy), steps, regions, frames, widths, vocabulary   */
    int unk_idx, attn_kind;               /* CVC_ATTN_ADDITIVE / CVC_ATTN_DOT                                                */
    float inv_temp;
    int stream_r, stream_f;               /* cvc_attn_set.stream of the region / frame feature sets                          */
    int path;                             /* 0 = packed (beam == 1, B <= 64), 1 = tile                                       */
    int qsplit;                           /* packed: K slices of the h2attn GEMM                                             */
    int ks_gate, ks_q, ks_o, ks_fc;       /* tile: K slices of the gate / h2attn / vocabulary / fc GEMMs                     */
    /* parameters (checkpoint tensors) */
    const float *b_ih_att, *b_hh_att, *b_ih_lang, *b_hh_lang, *b_h, *w_a, *b_a, *b_o, *embed;
    const float* w_fc; int ld_w_fc;       /* packed: W_ih_att[:, R:2R] row-major (leading dim E + 2R)                        */
    /* packed weights: packed path -> [blk][K/4][32][4] fp32; tile path -> bf16 fragments (cvc_tile_gemm's wb)              */
    const void *w_att, *w_lang, *w_h, *w_o, *w_fc_frag;
    /* clip features (model/backbone.py:350-351 outputs) */
    const float *fc, *conv, *pconv, *pool, *ppool; const uint8_t* mask;
    /* outputs */
    int64_t* words;                       /* [(T + 1), rows]; words[0] = BOS = 0 is written by the driver                    */
    float* att_steps;                     /* [T, rows, N] post-softmax region attention of every step                        */
    float* logprob;                       /* [T, rows] (greedy) or NULL                                                      */
    float* score; uint8_t* done; int64_t* parent;   /* beam: [2, rows], [2, rows], [T, rows]                                 */
    /* workspaces, all caller-allocated */
    float *gate_fc;                       /* packed: [rows, 4R]; tile: [B, 4R]                                               */
    float *scores_r, *scores_f, *attn_f;  /* [rows, N], [rows, F], [rows, F]                                                 */
    float *q;                             /* tile: [rows, A]                                                                 */
    float *q_parts;                       /* packed: [qsplit, rows, A]; tile: [ks_q, rows, A]                                */
    float *top2_part;                     /* packed: [ceil(V/32), 64, 6]                                                     */
    float *xa[2], *xl[2], *ca[2], *cl[2]; /* packed: quad-layout activation / cell-state ping-pong buffers                   */
    const float* xa0_init;                /* packed: XA of step 0 (relu(Emb[BOS]) in its K segment, zeros elsewhere)         */
    void *xaf, *xlf, *xhf, *xff;          /* tile: activation fragments XA [2R+E], XL [3R], XH [R], fc [R]                   */
    long long xaf_stride, xlf_stride, xhf_stride, xff_stride;
    float *parts_gate, *parts_o, *parts_fc, *logits;           /* tile                                                        */
    float *h_att, *c_att, *h_lang, *c_lang, *c_att_prev, *c_lang_prev; const float* zero_state;   /* tile: [rows, R] each     */
    float *beam_ws;                       /* beam: >= 17 * rows floats                                                       */
    /* packed path, grouped stream-K schedule (gsk_nwg > 0; needs R % 64 == 0): workgroups per stream-K launch (the CU count) and
     * the four groups' slabs, each >= ceil(nblk / 8) * maxseg * 16384 floats with maxseg from cvc_gsk_plan over the launches
     * {att-early (R/8 blocks, 2R/32 chunks), logits (ceil(V/32), R/32)}, {logits alone} and {lang-early (R/8, 2R/32), h2attn
     * (A/32, R/32)}.  gsk_nwg == 0: the one-launch-per-GEMM schedule (cvc_packed_lstm_fwd / cvc_packed_linear_fwd).           */
    int gsk_nwg;
    float *slab_att, *slab_lang, *slab_q, *slab_o;
    /* packed path, embedding-gate schedule (emb_gate != NULL; excludes gsk_nwg > 0): emb_gate = the table of
     * cvc_packed_lstm_embgate_fwd, w_att = the gate matrix packed over K = 2R (h_lang | h_att), xa = [2R/4][64][4] buffers,
     * xa0_init is not read; sel_counter is reserved (the one-launch select form, cvc_packed_linear_select_fwd, is not on the
     * default path: measured slower than two launches).                                                                       */
    const float* emb_gate;
    unsigned* sel_counter;
    int att_w_cached;                     /* embedding-gate schedule: 1 = w_att is read with the default cache policy (cvc_packed_lstm_embgate_ex_fwd) */
    /* packed path: lang_ksx = 1 runs the language cell on the K-split gate GEMM with the exchange finish (cvc_packed_lstm_ksx_fwd;
     * R = 2048 shapes, T > 1): ksx_slab >= 8 * (R / 8) * 2048 floats, ksx_flags R / 8 + 1 words, zero before the first decode
     * (the last word is the error word the caller checks after its first decode). */
    int lang_ksx;
    float* ksx_slab;
    unsigned* ksx_flags;
} cvc_decode_desc;
typedef struct cvc_decode_plan cvc_decode_plan;
CVC_API int cvc_decode_plan_create(const cvc_decode_desc* desc, cvc_decode_plan** plan);   /* validates, copies the descriptor       */
CVC_API void cvc_decode_plan_destroy(cvc_decode_plan* plan);
/* next batch of the SAME shape: point the plan at other feature tensors (no copy, no new plan; a graph captured from the plan
 * keeps the old pointers -- re-capture or copy into the bound buffers instead when replaying graphs) */
CVC_API int cvc_decode_plan_set_features(cvc_decode_plan* plan, const float* fc, const float* conv, const float* pconv,
                                 const float* pool, const float* ppool, const uint8_t* mask);
/* enqueue one full decode on `stream`; results land in desc.words / att_steps / logprob (greedy), + score / parent (beam) */
CVC_API int cvc_decode_greedy(cvc_decode_plan* plan, cvc_stream_t stream);
CVC_API int cvc_decode_beam(cvc_decode_plan* plan, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Training loops driven from C (round 4): the two recurrent loops of the cyclical training pass -- loop A, the teacher-forced
 * decode (model/captioner.py:242-270 calling decoder_core.py:30-66), and loop C, the reconstruction from the localized regions
 * (captioner.py:348-362 calling decoder_core.py:86-113) -- and their back-propagation through time, each ONE host call that
 * enqueues every launch of the T steps on `stream` (capturable), like cvc_decode_greedy does for inference.  Nothing of the
 * per-step work is left to the host framework: the cells hand h' to their consumers in the layouts those read (quad operand
 * rings for the next gate GEMMs, row-major rows for the dense weight-gradient products that follow the loop), the gradient
 * fan-in of h (three consumers per step) is summed inside the gate-gradient kernel, both feature sets of the attention go
 * through one backward.
 *
 * What stays outside (dense, once per loop, on the tile GEMM): the hoisted input products -- gpre_att = relu(Emb[w_t]) x
 * W_ih_att[:, emb]^T for all T steps, row_bias = fc_feats x W_ih_att[:, fc]^T + b_ih + b_hh, loop C's gpre_lang = (localized
 * context) x W_ih_lang[:, :R]^T -- the vocabulary head, and after the backward loop every weight gradient as ONE product over the
 * T * B sample rows from the row-major buffers below (dW_ih_att[:, :R] = dg_att^T h_lang_prev, dW_hh_att = dg_att^T h_att_prev,
 * dW_ih_lang = dg_lang^T [ctx | h_att], dW_hh_lang = dg_lang^T h_lang_prev, dW_h2attn = dq^T h_att, ...).
 *
 * Building blocks, also exported on their own:
 *   cvc_packed_lstm_step_fwd : the general training form of the packed gate GEMM + cell update (nn.LSTMCell,
 *                              decoder_core.py:50, 61): every output of cvc_packed_lstm_train_drop_fwd AND the quad destinations
 *                              of cvc_packed_lstm_fwd, plus a second per-row gate term gathered by row index;
 *   cvc_attn_wsum_quad_rm    : cvc_attn_wsum_quad that also writes the summed context row-major [rows, R];
 *   cvc_attn_bwd_pair        : cvc_attn_bwd for both feature sets of a decoder step at once (they share the query and d_ctx):
 *                              one score pass, one softmax backward, one score backward whose d_q is the sum over the sets,
 *                              written row-major and (nullable) in the quad layout cvc_linear_nn_fwd reads;
 *   cvc_lstm_pointwise_bwd4  : cvc_lstm_pointwise_bwd3 with a fourth gradient of h', the one that left through the fused
 *                              dropout (decoder_core.py:62, 109), summed in the order d_h[0], d_h[1], d_h[2], d_hd; each d_h[i]
 *                              may be the K-slice planes of the product that produced it (cvc_grad_src).
 */
/* A gradient that is the sum of `nplanes` partial planes -- what a K-split backward-data product leaves behind
 * (cvc_linear_nn_planes_fwd): element (m, j) = sum_k p[k * plane_stride + m * ld + j], summed in plane order by the consumer, so
 * the planes never need a summing launch of their own.  A finished tensor is nplanes = 1; p == NULL is an absent term (zero).
 * (cvc_attn_bwd_pair with feature gradients d_ctxfeat requires d_ctx to be a finished tensor.) */
typedef struct cvc_grad_src {
    const float* p;
    long long ld, plane_stride;
    int nplanes;
} cvc_grad_src;

typedef struct cvc_lstm_step {
    const float* wp;             /* packed gate weights (cvc_pack_lstm_segs), K columns                                  */
    const float* xq;             /* quad-layout activations [K/4][64][4]                                                 */
    int K, M, R;
    const float *b_ih, *b_hh;    /* [4R], nullable                                                                       */
    const float* gate_pre;       /* [M, 4R] row-major additive pre-activation term, nullable                             */
    const float* row_bias;       /* [*, 4R]: row row_index[m] is added to row m's pre-activations, nullable              */
    const int64_t* row_index;    /* [M] (with row_bias)                                                                  */
    const float* c_prev;         /* [M, R] row-major                                                                     */
    float *c_out, *gates_out;    /* [M, R]; activated gates [M, 4R] (i, f, g, o), nullable                               */
    float *h_out, *h_out2;       /* row-major copies of h' [M, R], nullable                                              */
    float* h_drop_out;           /* [M, R] nn.Dropout(h') with the counter-based mask (rng_state null or p == 0: h'), nullable */
    const uint32_t* rng_state; unsigned site; float p;
    float *h_dst1_q, *h_dst2_q;  /* h' in the quad layout at the consumers' K offsets, nullable                           */
    int w_cached;                /* 1: the gate weights are read under the default cache policy (they fit the Infinity Cache and
                                  * nothing between two launches evicts them) instead of streamed non-temporally */
} cvc_lstm_step;
CVC_API int cvc_packed_lstm_step_fwd(const cvc_lstm_step* s, cvc_stream_t stream);


/* sets[s]: proj / ctx / attn / n as in the forward; scores = d_scores out [rows, n] (the pre-softmax gradient; its sum is
 * d_b_alpha); frame_masked = gradient of the frame-masked output (d_fm, INPUT, nullable); ctx_out unused.  d_ctx [rows, R] is
 * the gradient of the summed context.  d_q [rows, A] row-major, d_q_q (nullable, rows <= 64) the same in the quad layout
 * [A/4][64][4], d_w_part [rows, A] (additive only, nullable) per-row partials of d_w_alpha summed over the sets.
 * d_proj[s] / d_ctxfeat[s] (nullable): ACCUMULATED feature gradients [nclip, n, A] / [nclip, n, R].
 * q: the forward's query h2attn(h), as a finished [rows, A] tensor (nplanes = 1, q_bias null) or as the split-K planes the
 * forward's cvc_packed_linear_fwd left (+ q_bias = h2attn.bias), summed on load. */

CVC_API int cvc_lstm_pointwise_bwd4(const cvc_grad_src* d_h /* [3] */, const float* d_hd,
                            const uint32_t* rng_state, unsigned site, float p, const float* d_c, const float* gates,
                            const float* c_prev, const float* c_new, int M, int R, float* d_gates, float* d_c_prev,
                            float* d_gates_q, float* dg_sum /* [M, 4R] += d_gates, nullable */,
                            int q_row0 /* row of d_gates_q that row 0 goes to: q_row0 + M <= 64 */, cvc_stream_t stream);

typedef struct cvc_train_loop {
    int kind;                    /* 0 = loop A (decode step with attention), 1 = loop C (reconstruction step)            */
    int B, T, R;                 /* B <= 64 clips, R % 32 == 0                                                           */
    int A, N, F, attn_kind;      /* kind 0: attention sizes, CVC_ATTN_*                                                  */
    float inv_temp;
    /* ---- weights.  Packs (cvc_pack_lstm_segs, rebuilt once per optimizer step): wp_att over [W_ih_att[:, :R] | W_hh_att]
     * (K = 2R: h_lang(t-1), h_att(t-1)); wp_lang over [W_ih_lang | W_hh_lang] (kind 0, K = 3R: ctx, h_att(t), h_lang(t-1)) or
     * [W_ih_lang[:, R:] | W_hh_lang] (kind 1, K = 2R).  The row-major checkpoint tensors serve the backward-data products. */
    const float *wp_att, *wp_lang;
    const float *b_ih_att, *b_hh_att, *b_ih_lang, *b_hh_lang;
    const float *w_ih_att, *w_hh_att, *w_ih_lang, *w_hh_lang;
    int ld_ih_att, ld_ih_lang;   /* leading dimensions of weight_ih (R + [R] + E, 2R); weight_hh has R                   */
    const float *w_h, *b_h;      /* kind 0: h2attn [A, R], [A]                                                           */
    const float* wp_h;           /* kind 0, nullable: h2attn packed [A/32][R/4][32][4] (cvc.decode.pack_weights): the query GEMM then
                                  * runs split-K over the chip on the packed kernel (q_split planes summed by the score pass)
                                  * instead of on the row-major ring kernel (q_split = 1)                                */
    int q_split;
    const float *w_a, *b_a;      /* kind 0, additive: alpha_net weight [A], bias [1]                                     */
    /* ---- inputs */
    const float* gpre_att;       /* [T][B][4R] hoisted gates of the attention cell (embedded word of every step)         */
    const float* row_bias;       /* [B][4R] fc_feats term + both biases of the attention cell (then b_*_att are null), nullable */
    const int64_t* row_index;    /* [B] = 0 .. B-1                                                                       */
    const float* gpre_lang;      /* kind 1: [T][B][4R] hoisted gates of the language cell (localized context)            */
    const float *pool, *ppool, *conv, *pconv;      /* kind 0: [B,N,R], [B,N,A], [B,F,R], [B,F,A]                         */
    const uint8_t* mask;         /* kind 0: [B, N], nullable                                                             */
    const uint8_t* frame_mask;   /* kind 0: [T][B][N], nullable                                                          */
    const uint32_t* rng_state; unsigned site0; float p;     /* output dropout of step t: site site0 + t (p == 0: none)    */
    /* ---- forward outputs = what the backward and the dense products read; t-major rows (t * B + b)                     */
    float* out;                  /* [T][B][R] dropout(h_lang(t))                                                         */
    float* h_att;                /* [T][B][R] h_att(t)                                                                   */
    float* h_att_prev;           /* [T][B][R] h_att(t-1), row block 0 zero                                               */
    float* h_lang_prev;          /* [T][B][R] h_lang(t-1), row block 0 zero                                              */
    float *c_att, *c_lang;       /* [T+1][B][R], row block 0 zero                                                        */
    float *g_att, *g_lang;       /* [T][B][4R] activated gates                                                           */
    float* ctx;                  /* kind 0: [T][B][R] attended context (regions + frames)                                */
    float* q;                    /* kind 0: [T][q_split][B][A] h2attn(h_att(t)) (q_split > 1: split-K planes, bias not included) */
    float *attn_r, *attn_f;      /* kind 0: [T][B][N], [T][B][F] softmax weights                                         */
    float* fm;                   /* kind 0: [T][B][N] frame-masked pre-softmax scores (with frame_mask), nullable        */
    float* scores_ws;            /* kind 0: B * (N + F) floats                                                           */
    float *xa[2], *xl[2];        /* quad operand rings: [2R/4][64][4] and [3R/4][64][4] (kind 1: 2R) each                */
    /* ---- backward */
    const float* d_out;          /* [T][B][R]                                                                            */
    const float* d_fm;           /* kind 0: [T][B][N], nullable                                                          */
    float *dg_att, *dg_lang;     /* [T][B][4R] pre-activation gate gradients                                             */
    float *dgsum_att, *dgsum_lang;   /* [B][4R] their sums over the T steps (bias gradients, dY of the fc_feats columns), nullable */
    float* dq;                   /* kind 0: [T][B][A]                                                                    */
    float* dwa_part;             /* kind 0, additive: [T][B][A]                                                          */
    float *ds_r, *ds_f;          /* kind 0: [T][B][N], [T][B][F] gradients of the pre-softmax scores                     */
    float *d_pool, *d_ppool, *d_conv, *d_pconv;    /* kind 0: accumulated feature gradients, nullable                    */
    float* bwd_ws;               /* cvc_train_loop_bwd_ws(B, R, A) floats, ZERO before the first use                     */
    float* d_ctx_all;            /* kind 0, nullable: [T][128][R] floats.  With d_pool / d_conv: the context gradient of EVERY step
                                  * is kept here and the context-feature gradients d_pool[b,n,:] += sum_t attn_t[b,n] d_ctx_t[b,:]
                                  * are taken in ONE pass after the loop (each feature row read and written once) instead of a
                                  * read-modify-write of both feature tensors at every step                               */
} cvc_train_loop;
CVC_API long long cvc_train_loop_bwd_ws(int B, int R, int A);
CVC_API int cvc_train_loop_fwd(const cvc_train_loop* loop, cvc_stream_t stream);
CVC_API int cvc_train_loop_bwd(const cvc_train_loop* loop, cvc_stream_t stream);
/* Back-propagation through BOTH loops in one pass (same T and R): the two loops share the LSTM cells, so at every step every
 * backward-data product takes both loops' gate gradients and streams its weights once -- 3 products per step instead of 5.
 * loop_a->B + loop_c->B <= 64 (the 32-clip shares of the 8-GPU job): ONE 64-row operand, loop A's rows first.  Up to 64 + 64 rows
 * (config 3): two 64-row operand groups on the 128-row form of the product (split-product arithmetic only).  Same results as the two
 * separate calls under the same K split (bwd_ws: cvc_train_loop_bwd_ws(loop_a->B + loop_c->B, R, A) floats, the one of loop_a is used). */
CVC_API int cvc_train_loops_bwd_joint(const cvc_train_loop* loop_a, const cvc_train_loop* loop_c, cvc_stream_t stream);
/* Measurement aid (bench.py --mode train; never enabled by the product path): HIP-event pairs around every entry point the two
 * drivers call, on the launch stream.  cvc_train_loop_profile(n > 0) starts recording with room for n launches, (0) stops and
 * frees; cvc_train_loop_profile_read waits for the recorded launches and returns their count, kind[i] (0 zero fill, 1 attention
 * cell, 2 h2attn, 3 score pass, 4 weighted sum, 5 language cell, 6 / 10 gate gradients of the language / attention cell, 7 / 9 /
 * 11 backward-data product of the language cell / h2attn / the attention cell, 8 attention backward), loop[i] (0 / 1 forward of
 * loop A / C, 2 / 3 their backward, 4 a product of the joint backward that serves both loops) and ms[i]; the record is emptied. */
CVC_API int cvc_train_loop_profile(int enable);
CVC_API int cvc_train_loop_profile_read(int* kind, int* loop, float* ms, int cap);

/* ---------------------------------------------------------------------------------------
 * Optimizer step of the training path (trainer.py:116-122: nn.utils.clip_grad_norm_ -> optimizer.step(); Adam with one group per
 * tensor, main.py:171-191).  cvc_adam_clip_step = global L2 norm of all gradients (chunk sums combined in chunk order:
 * deterministic), clip coefficient min(1, max_norm / (inv_world * norm + 1e-6)) * inv_world (inv_world = 1 / ranks when the
 * gradients are sums over ranks; max_norm <= 0: no clipping), every segment's step count += 1, then torch.optim.Adam's update
 * (amsgrad = False) on (p, g * coef, m, v) in ONE pass; write_grad == 1 also stores the clipped gradient back, as clip_grad_norm_
 * leaves it; write_grad == 2 stores ZERO instead -- the next step's zero_grad() folded into this pass (the gradients of a step
 * are read here for the last time).  Three launches, nothing read back by the host (graph-capturable).
 *   skip   : NULL, or a DEVICE int: when it is non-zero at execution time the step is VOID -- no parameter, moment or step count
 *            changes (write_grad == 2 still leaves the gradients zero).  The training step's status word (cvc.hip.step_status):
 *            a persistent GRU launch whose barrier timed out marks the step, the host re-runs it later (cvc/trainer.py).
 *   segs   : DEVICE array of nseg parameter segments;   chunks : DEVICE array of nchunk (segment, start) pairs covering every
 *            segment in pieces of cvc_optim_chunk_elems() elements (start a multiple of it), in any fixed order;
 *   partial: nchunk floats of scratch;   norm_coef: 2 floats out -- [0] the norm of the (averaged) gradient, [1] the coefficient. */
typedef struct {
    float *p, *g, *m, *v;     /* parameter, gradient, exp_avg, exp_avg_sq: n floats each */
    float* step;              /* the parameter's step count, one float32 on the device (torch's `step` state) */
    long long n;
    float lr, weight_decay;
} cvc_optim_seg;
typedef struct { int seg; int pad; long long start; } cvc_optim_chunk;
CVC_API int cvc_optim_chunk_elems(void);
CVC_API int cvc_adam_clip_step(const cvc_optim_seg* segs, int nseg, const cvc_optim_chunk* chunks, int nchunk, float max_norm,
                       float inv_world, float beta1, float beta2, float eps, int write_grad, float* partial,
                       float* norm_coef, const int* skip, cvc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU training: the one exchange step of the path.  The reference reduces gradients inside nn.DataParallel
 * (main.py:169; per-replica token-mean losses, unweighted mean over replicas, trainer.py:101-122); with one process per GPU
 * that is a SUM all-reduce of the gradients over RCCL / xGMI followed by 1/G, which the caller folds into its clip multiply
 * (cvc.distributed.GradReducer.clip_).  cvc_comm_unique_id (rank 0) -> share the 128 bytes with every rank by any means ->
 * cvc_comm_init on every rank -> cvc_allreduce_grads(comm, flat gradient arena, floats, stream) per bucket, in place, stream
 * ordered, no host synchronisation.  The exchange is the one cvc.distributed.GradReducer issues through torch.distributed: an
 * in-place ncclReduceScatter (every rank sums ITS 1 / G of the arena) + an in-place ncclAllGather, back to back (count divisible
 * by the world size; otherwise one ncclAllReduce).  Return codes: 0, CVC_E_*, or 1000 + ncclResult_t.  librccl is dlopen'ed on
 * first use.
 */
CVC_API int cvc_comm_unique_id(void* out128);
CVC_API int cvc_comm_init(int world, int rank, const void* id128, void** comm);
CVC_API int cvc_allreduce_grads(void* comm, float* grads, long long count, cvc_stream_t stream);
CVC_API int cvc_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* CVC_HIP_H */
