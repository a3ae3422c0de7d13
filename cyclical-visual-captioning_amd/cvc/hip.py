"""ctypes binding of libcvc_hip.so (include/cvc_hip.h) -- the only door between the host-side
module mirrors and the gfx950 kernels.

There is NO CPU fallback: if the library is missing or a tensor is not a contiguous fp32 CUDA
(ROCm) tensor, calls raise.  PyTorch is used for device memory and streams only; the pointers
handed over are raw `data_ptr()`s and the current HIP stream handle.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CVC_LIB: a variant library for A/B measurements -- tools/runs/build_variant.sh; the product loads the in-tree build)
LIB_PATH = os.environ.get("CVC_LIB") or os.path.join(_HERE, "lib", "libcvc_hip.so")

ATTN_ADDITIVE, ATTN_DOT = 0, 1


class AttnSet(C.Structure):
    _fields_ = [("proj", C.c_void_p), ("ctx", C.c_void_p), ("mask", C.c_void_p), ("frame_mask", C.c_void_p),
                ("scores", C.c_void_p), ("frame_masked", C.c_void_p), ("attn", C.c_void_p), ("ctx_out", C.c_void_p),
                ("n", C.c_int), ("stream", C.c_int)]


class GemmSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("idx", C.c_void_p), ("w", C.c_void_p),
                ("k", C.c_int), ("ldx", C.c_int), ("ldw", C.c_int), ("relu", C.c_int)]


class NNSeg(C.Structure):
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("ldw", C.c_int), ("ncols", C.c_int), ("ld_dst", C.c_int)]


_P, _I, _F, _LL = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class OptimSeg(C.Structure):
    """cvc_optim_seg (include/cvc_hip.h): one parameter of the fused clip + Adam step"""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p), ("n", C.c_longlong),
                ("lr", C.c_float), ("weight_decay", C.c_float)]


class OptimChunk(C.Structure):
    _fields_ = [("seg", C.c_int), ("pad", C.c_int), ("start", C.c_longlong)]


class GskGroup(C.Structure):
    """cvc_gsk_group: one GEMM of a grouped stream-K launch (include/cvc_hip.h, "Grouped stream-K form")."""
    _fields_ = [("wp", C.c_void_p), ("w_blk_stride", C.c_longlong), ("xq", C.c_void_p), ("nblk", C.c_int), ("nchunk", C.c_int),
                ("skip_at", C.c_int), ("skip_n", C.c_int), ("slab", C.c_void_p), ("maxseg", C.c_int)]


class GskSegs(C.Structure):
    """cvc_gsk_segs: where a consumer finds one group's partial tiles."""
    _fields_ = [("slab", C.c_void_p), ("unit0", C.c_int), ("nchunk", C.c_int), ("U", C.c_int), ("maxseg", C.c_int)]


class DecodeDesc(C.Structure):
    """cvc_decode_desc of include/cvc_hip.h, field for field (tests/test_cabi.py compares the size with the C compiler's)."""
    _fields_ = (
        [(n, C.c_int) for n in ("B", "beam", "T", "N", "F", "R", "A", "E", "V", "unk_idx", "attn_kind")] +
        [("inv_temp", C.c_float)] +
        [(n, C.c_int) for n in ("stream_r", "stream_f", "path", "qsplit", "ks_gate", "ks_q", "ks_o", "ks_fc")] +
        [(n, C.c_void_p) for n in ("b_ih_att", "b_hh_att", "b_ih_lang", "b_hh_lang", "b_h", "w_a", "b_a", "b_o", "embed")] +
        [("w_fc", C.c_void_p), ("ld_w_fc", C.c_int)] +
        [(n, C.c_void_p) for n in ("w_att", "w_lang", "w_h", "w_o", "w_fc_frag")] +
        [(n, C.c_void_p) for n in ("fc", "conv", "pconv", "pool", "ppool", "mask")] +
        [(n, C.c_void_p) for n in ("words", "att_steps", "logprob", "score", "done", "parent")] +
        [(n, C.c_void_p) for n in ("gate_fc", "scores_r", "scores_f", "attn_f", "q", "q_parts", "top2_part")] +
        [("xa", C.c_void_p * 2), ("xl", C.c_void_p * 2), ("ca", C.c_void_p * 2), ("cl", C.c_void_p * 2), ("xa0_init", C.c_void_p)] +
        [(n, C.c_void_p) for n in ("xaf", "xlf", "xhf", "xff")] +
        [(n, C.c_longlong) for n in ("xaf_stride", "xlf_stride", "xhf_stride", "xff_stride")] +
        [(n, C.c_void_p) for n in ("parts_gate", "parts_o", "parts_fc", "logits")] +
        [(n, C.c_void_p) for n in ("h_att", "c_att", "h_lang", "c_lang", "c_att_prev", "c_lang_prev", "zero_state")] +
        [("beam_ws", C.c_void_p)] +
        [("gsk_nwg", C.c_int)] + [(n, C.c_void_p) for n in ("slab_att", "slab_lang", "slab_q", "slab_o", "emb_gate", "sel_counter")] +
        [("att_w_cached", C.c_int), ("lang_ksx", C.c_int), ("ksx_slab", C.c_void_p), ("ksx_flags", C.c_void_p)])

class GradSrc(C.Structure):
    """cvc_grad_src: a gradient given as the sum of K-slice planes"""
    _fields_ = [("p", C.c_void_p), ("ld", C.c_longlong), ("plane_stride", C.c_longlong), ("nplanes", C.c_int)]


class PwBwdArgs(C.Structure):
    """cvc_pw_bwd_args of include/cvc_hip_blocks.h (one argument set of cvc_lstm_pointwise_bwd4_pair), field for field"""
    _fields_ = ([("d_h", GradSrc * 3), ("d_hd", C.c_void_p), ("rng_state", C.c_void_p), ("site", C.c_uint), ("p", C.c_float)] +
                [(n, C.c_void_p) for n in ("d_c", "gates", "c_prev", "c_new")] + [("M", C.c_int)] +
                [(n, C.c_void_p) for n in ("d_gates", "d_c_prev", "d_gates_q", "dg_sum")] + [("q_row0", C.c_int)])


class LstmStep(C.Structure):
    """cvc_lstm_step of include/cvc_hip.h ("Training loops driven from C"), field for field"""
    _fields_ = ([("wp", C.c_void_p), ("xq", C.c_void_p), ("K", C.c_int), ("M", C.c_int), ("R", C.c_int)] +
                [(n, C.c_void_p) for n in ("b_ih", "b_hh", "gate_pre", "row_bias", "row_index", "c_prev", "c_out", "gates_out", "h_out",
                                           "h_out2", "h_drop_out", "rng_state")] +
                [("site", C.c_uint), ("p", C.c_float), ("h_dst1_q", C.c_void_p), ("h_dst2_q", C.c_void_p), ("w_cached", C.c_int)])


class TrainLoop(C.Structure):
    """cvc_train_loop of include/cvc_hip.h, field for field (tests/test_cabi.py compares the size with the C compiler's)."""
    _fields_ = (
        [(n, C.c_int) for n in ("kind", "B", "T", "R", "A", "N", "F", "attn_kind")] + [("inv_temp", C.c_float)] +
        [(n, C.c_void_p) for n in ("wp_att", "wp_lang", "b_ih_att", "b_hh_att", "b_ih_lang", "b_hh_lang", "w_ih_att", "w_hh_att",
                                   "w_ih_lang", "w_hh_lang")] +
        [("ld_ih_att", C.c_int), ("ld_ih_lang", C.c_int)] +
        [(n, C.c_void_p) for n in ("w_h", "b_h", "wp_h")] + [("q_split", C.c_int)] +
        [(n, C.c_void_p) for n in ("w_a", "b_a", "gpre_att", "row_bias", "row_index", "gpre_lang", "pool", "ppool", "conv",
                                   "pconv", "mask", "frame_mask", "rng_state")] +
        [("site0", C.c_uint), ("p", C.c_float)] +
        [(n, C.c_void_p) for n in ("out", "h_att", "h_att_prev", "h_lang_prev", "c_att", "c_lang", "g_att", "g_lang", "ctx", "q", "attn_r",
                                   "attn_f", "fm", "scores_ws")] +
        [("xa", C.c_void_p * 2), ("xl", C.c_void_p * 2)] +
        [(n, C.c_void_p) for n in ("d_out", "d_fm", "dg_att", "dg_lang", "dgsum_att", "dgsum_lang", "dq", "dwa_part", "ds_r", "ds_f", "d_pool", "d_ppool", "d_conv",
                                   "d_pconv", "bwd_ws", "d_ctx_all")])


# name -> argtypes, exactly the declarations of include/cvc_hip.h (tests/test_cabi.py checks both)
SIGNATURES = {
    "cvc_packed_lstm_step_fwd": [C.POINTER(LstmStep), _P],
    "cvc_attn_wsum_quad_rm": [C.POINTER(AttnSet), _I, _I, _I, _P, _P, _P],
    "cvc_attn_bwd_pair": [_I, C.POINTER(GradSrc), _P, _P, _F, C.POINTER(AttnSet), _I, C.POINTER(GradSrc), _I, _I, _I, _I, _P, _P, _P,
                          C.POINTER(_P), C.POINTER(_P), _P],
    "cvc_ctxfeat_bwd_steps": [_P, _P, _I, _I, _I, _I, _P, _P],
    "cvc_tile_gemm_big": [_I, _I],
    "cvc_tile_gemm_plan": [_I, _I, _I, _P, _P, _P],
    "cvc_dproj_bwd_steps": [_P, _LL, _LL, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "cvc_lstm_pointwise_bwd4": [C.POINTER(GradSrc), _P, _P, C.c_uint, _F, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "cvc_vocab_head_nll_fwd": [_P, _I, _LL, _I, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _P],
    "cvc_scale_by_scalar": [_P, _P, _LL, _P, _P],
    "cvc_relu_dropout_fwd": [_P, _P, _LL, _I, _P, C.c_uint, _F, _P, _P],
    "cvc_relu_dropout_bwd": [_P, _P, _LL, _P, C.c_uint, _F, _P, _P],
    "cvc_bn_workspace": [_LL, _I],
    "cvc_bn_relu_train_fwd": [_P, _P, _P, _F, _F, _P, _P, _LL, _I, _P, _P, _P, _P, _P],
    "cvc_bn_relu_train_bwd": [_P, _P, _P, _P, _P, _P, _LL, _I, _P, _P, _P, _P, _P],
    "cvc_class_softmax_bwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P],
    "cvc_layernorm_cat_bwd": [C.POINTER(_P), C.POINTER(_LL), C.POINTER(_I), _I, _LL, _F, _P, _LL, C.POINTER(_P), C.POINTER(_LL), _P],
    "cvc_bbox_overlaps_fwd": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P],
    "cvc_label_glue_fwd": [_P, _P, _LL, _LL, _LL, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "cvc_attn_nll_fwd": [_P, _LL, _LL, _P, _LL, _LL, _P, _I, _I, _I, _P, _P, _P],
    "cvc_attn_nll_bwd": [_P, _LL, _LL, _P, _LL, _LL, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cvc_train_loop_bwd_ws": [_I, _I, _I],
    "cvc_train_loop_fwd": [C.POINTER(TrainLoop), _P],
    "cvc_train_loop_bwd": [C.POINTER(TrainLoop), _P],
    "cvc_train_loops_bwd_joint": [C.POINTER(TrainLoop), C.POINTER(TrainLoop), _P],
    "cvc_stable_order": [_P, _I, _P, _P],
    "cvc_attn_weighted_rows": [_P, _P, _I, _I, _I, _I, _F, _P, _P],
    "cvc_col_sum": [_P, _LL, _I, _I, _P, _P, _P, _P],
    "cvc_col_sum_ws": [_I, _I],
    "cvc_train_loop_profile": [_I],
    "cvc_train_loop_profile_read": [C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), _I],
    "cvc_attn_fwd": [_I, _P, _P, _P, _F, C.POINTER(AttnSet), _I, _I, _I, _I, _I, _P, _P],
    "cvc_attn_scores": [_I, _P, _P, _P, _F, C.POINTER(AttnSet), _I, _I, _I, _I, _P],
    "cvc_attn_scores_qparts": [_I, _P, _I, _P, _P, _P, _F, C.POINTER(AttnSet), _I, _I, _I, _I, _P],
    "cvc_linear_splitk_fwd": [C.POINTER(GemmSeg), _I, _P, _I, _I, _I, _P, _P],
    "cvc_linear_top2_fwd": [C.POINTER(GemmSeg), _I, _P, _I, _I, _P, _P, _P],
    "cvc_top2_final": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P],
    "cvc_attn_wsum_quad": [C.POINTER(AttnSet), _I, _I, _I, _I, _P, _P],
    "cvc_packed_lstm_fwd": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "cvc_packed_linear_fwd": [_P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _P],
    "cvc_packed_lstm_wg_blocks": [_I],
    "cvc_packed_lstm_embgate_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "cvc_packed_lstm_embgate_ex_fwd": [_P, _LL, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P],
    "cvc_packed_linear_select_fwd": [_P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _I, _P, _P],
    "cvc_gsk_plan": [C.POINTER(_I), C.POINTER(_I), _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)],
    "cvc_gsk_gemm": [C.POINTER(GskGroup), _I, _I, _P],
    "cvc_packed_lstm_late_fwd": [_P, _LL, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, C.POINTER(GskSegs), _P],
    "cvc_attn_scores_qslab": [_I, C.POINTER(GskSegs), _P, _P, _P, _F, C.POINTER(AttnSet), _I, _I, _I, _I, _P],
    "cvc_top2_slab": [C.POINTER(GskSegs), _P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P],
    "cvc_packed_lstm_train_fwd": [_P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "cvc_pack_lstm_weights": [_P, _I, _P, _I, _I, _P, _P],
    "cvc_pack_lstm_segs": [C.POINTER(_P), C.POINTER(_LL), C.POINTER(_I), _I, _I, _P, _P],
    "cvc_packed_lstm_train_pre_fwd": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "cvc_pack_quad_segs": [C.POINTER(_P), C.POINTER(_LL), C.POINTER(_I), _I, _I, _P, _P],
    "cvc_beam_backtrack": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "cvc_gru_seq_fwd": [_P, _P, _LL, _LL, _P, _P, _I, _I, _I, _I, _P, _P, _LL, _LL, _P],
    "cvc_gru_seq_train_fwd": [_P, _P, _LL, _LL, _P, _P, _I, _I, _I, _I, _P, _P, _LL, _LL, _P, _LL, _LL, _P],
    "cvc_lstm_pointwise_bwd4_pair": [C.POINTER(PwBwdArgs), C.POINTER(PwBwdArgs), _I, _P],
    "cvc_gru_persistent_halves": [_I],
    "cvc_gru_persistent_waves8": [_I],
    "cvc_gru_persistent_sync_words": [],
    "cvc_gru_seq_persistent_train_fwd": [_P, _P, _LL, _LL, _P, _P, _I, _I, _I, _I, _P, _P, _LL, _LL, _P, _LL, _LL, _P, _P],
    "cvc_gru_seq_bwd_ksplit": [_I],
    "cvc_gru_bwd_persistent_sync_words": [],
    "cvc_gru_seq_bwd_persistent": [_P, _LL, _LL, _P, _LL, _LL, _P, _LL, _LL, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "cvc_linear_nn_planes_fwd": [_P, _I, _I, C.POINTER(NNSeg), _I, _I, _P, _P],
    "cvc_linear_nn_planes2_fwd": [_P, _P, _I, _I, _I, C.POINTER(NNSeg), _I, _I, _P, _I, _P],
    "cvc_gru_seq_bwd": [_P, _LL, _LL, _P, _LL, _LL, _P, _LL, _LL, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "cvc_gru_seq_persistent_fwd": [_P, _P, _LL, _LL, _P, _P, _I, _I, _I, _I, _P, _P, _LL, _LL, _P, _P],
    "cvc_packed_lstm_ks_slices": [_I, _I],
    "cvc_packed_lstm_ks_fwd": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _LL, _P],
    "cvc_packed_lstm_ksf_fwd": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "cvc_packed_lstm_ksx_local": [_I],
    "cvc_packed_lstm_ksx_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, C.c_uint, _P],
    "cvc_attn_wsum": [C.POINTER(AttnSet), _I, _I, _I, _I, _P, _P],
    "cvc_attn_bwd": [_I, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cvc_linear_fwd": [C.POINTER(GemmSeg), _I, _P, _P, _I, _I, _P, _I, _P],
    "cvc_gemm_force_generic": [_I],
    "cvc_gemm_packed_split": [_I],
    "cvc_lstm_cell_fwd": [C.POINTER(GemmSeg), _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "cvc_lstm_pointwise_bwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "cvc_lstm_pointwise_bwd3": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "cvc_linear_nn_fwd": [_P, _I, _I, C.POINTER(NNSeg), _I, _I, _P, _P],
    "cvc_pack_quad": [_P, _LL, _I, _I, _P, _P],
    "cvc_embed_relu_fwd": [_P, _P, _P, _I, _I, _P, _P],
    "cvc_embed_relu_bwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "cvc_embed_relu_rng_fwd": [_P, _P, _P, C.c_uint, _F, _I, _I, _P, _P],
    "cvc_embed_relu_rng_bwd": [_P, _P, _P, _P, C.c_uint, _F, _P, _I, _I, _P, _P, _P],
    "cvc_dropout_rng": [_P, _LL, _P, C.c_uint, _F, _P, _P],
    "cvc_packed_lstm_train_drop_fwd": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, C.c_uint, _F, _P],
    "cvc_lstm_pointwise_bwd3_drop": [_P, _P, _P, _P, C.c_uint, _F, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "cvc_log_softmax_fwd": [_P, _I, _I, _P, _P],
    "cvc_log_softmax_bwd": [_P, _P, _I, _I, _P, _P],
    "cvc_nll_bwd": [_P, _P, _P, _I, _I, _P, _P],
    "cvc_top2_unk": [_P, _I, _I, _I, _P, _I, _P, _P],
    "cvc_nll_fwd": [_P, _P, _P, _I, _I, _P, _P],
    "cvc_vocab_nll_fwd": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "cvc_vocab_nll_bwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P],
    "cvc_grounder_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "cvc_grounder_bwd": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "cvc_beam_select": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cvc_beam_select_parts": [_P, _I, _LL, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cvc_gather_rows": [_P, _P, _I, _I, _I, _P, _P],
    "cvc_tile_rows_alloc": [_I],
    "cvc_tile_gemm_loaders": [_I],
    "cvc_tile_gemm": [_P, _P, _LL, _I, _I, _I, _I, _P, _I, _LL, _P],
    "cvc_tile_lstm_finish": [_P, _I, _LL, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _LL, _P, _LL, _P],
    "cvc_tile_lstm_finish_embgate": [_P, _I, _LL, _P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P, _LL, _P, _LL, _P],
    "cvc_tile_linear_finish": [_P, _I, _LL, _I, _P, _P, _I, _I, _P, _I, _P],
    "cvc_tile_pack_rows": [_P, _I, _P, _I, _I, _I, _P, _LL, _P],
    "cvc_tile_pack_rows_any": [_P, _LL, _I, _I, _P, _LL, _P],
    "cvc_tile_pack_cols": [_P, _LL, _I, _I, _P, _LL, _P],
    "cvc_tile_reorder_pack": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _LL, _P, _LL, _I, _I, _P],
    "cvc_attn_wsum_frag": [C.POINTER(AttnSet), _I, _I, _I, _I, _P, _LL, _P],
    "cvc_class_softmax_fwd": [_P, _LL, _P, _P, _I, _I, _I, _P, _P, _P],
    "cvc_layernorm_cat_fwd": [C.POINTER(_P), C.POINTER(_LL), C.POINTER(_I), _I, _LL, _F, _P, _LL, _P],
    "cvc_frame_embed_fwd": [_P, _P, _I, _P, _P, _I, _P, _P, _LL, _P, _P],
    "cvc_decode_plan_create": [C.POINTER(DecodeDesc), C.POINTER(C.c_void_p)],
    "cvc_decode_plan_destroy": [_P],
    "cvc_decode_plan_set_features": [_P, _P, _P, _P, _P, _P, _P],
    "cvc_decode_num_launches": [_P],
    "cvc_decode_greedy": [_P, _P],
    "cvc_decode_beam": [_P, _P],
    "cvc_optim_chunk_elems": [],
    "cvc_adam_clip_step": [_P, _I, _P, _I, _F, _F, _F, _F, _F, _I, _P, _P, _P, _P],
    "cvc_comm_unique_id": [_P],
    "cvc_comm_init": [_I, _I, _P, C.POINTER(C.c_void_p)],
    "cvc_allreduce_grads": [_P, _P, _LL, _P],
    "cvc_comm_destroy": [_P],
}
_VOID_RETURN = {"cvc_decode_plan_destroy"}
# Building blocks (include/cvc_hip_blocks.h) and experimental forms (include/cvc_hip_experimental.h, only in a library built with
# CVC_EXPERIMENTAL=1) are NOT exported: they are bound through cvc_block("name").  Everything else in SIGNATURES is the exported
# drop-in ABI of include/cvc_hip.h.
BLOCKS = {
    "cvc_attn_scores", "cvc_attn_wsum", "cvc_attn_scores_qparts", "cvc_attn_wsum_quad", "cvc_attn_wsum_frag", "cvc_attn_wsum_quad_rm",
    "cvc_attn_bwd_pair", "cvc_ctxfeat_bwd_steps", "cvc_dproj_bwd_steps", "cvc_tile_gemm_big", "cvc_tile_gemm_plan", "cvc_linear_splitk_fwd", "cvc_linear_top2_fwd", "cvc_top2_final", "cvc_packed_lstm_fwd", "cvc_packed_linear_fwd",
    "cvc_packed_lstm_embgate_fwd", "cvc_packed_lstm_embgate_ex_fwd", "cvc_packed_lstm_late_fwd", "cvc_packed_lstm_train_fwd",
    "cvc_packed_lstm_train_pre_fwd", "cvc_packed_lstm_train_drop_fwd", "cvc_lstm_pointwise_bwd", "cvc_lstm_pointwise_bwd3",
    "cvc_lstm_pointwise_bwd3_drop", "cvc_pack_lstm_weights", "cvc_linear_nn_planes_fwd", "cvc_linear_nn_planes2_fwd", "cvc_gru_seq_train_fwd", "cvc_lstm_pointwise_bwd4_pair", "cvc_beam_select_parts", "cvc_tile_lstm_finish",
    "cvc_tile_lstm_finish_embgate", "cvc_tile_reorder_pack", "cvc_decode_num_launches", "cvc_gemm_force_generic",
    "cvc_tile_gemm_loaders", "cvc_gru_persistent_waves8", "cvc_relu_dropout_fwd", "cvc_relu_dropout_bwd", "cvc_bn_workspace",
    "cvc_bn_relu_train_fwd", "cvc_bn_relu_train_bwd", "cvc_class_softmax_bwd", "cvc_layernorm_cat_bwd", "cvc_stable_order", "cvc_col_sum", "cvc_col_sum_ws", "cvc_attn_weighted_rows"}
EXPERIMENTAL = {
    "cvc_gsk_plan", "cvc_gsk_gemm", "cvc_attn_scores_qslab", "cvc_top2_slab", "cvc_packed_lstm_ks_slices", "cvc_packed_lstm_ks_fwd",
    "cvc_packed_lstm_ksf_fwd", "cvc_packed_lstm_ksx_local", "cvc_packed_lstm_ksx_fwd", "cvc_packed_lstm_wg_blocks",
    "cvc_packed_linear_select_fwd", "cvc_gru_persistent_halves"}

_lib = None


def _missing(name):
    def fn(*_a, **_k):
        raise RuntimeError(f"{name} is an experimental form (include/cvc_hip_experimental.h): this libcvc_hip.so was built without "
                           "CVC_EXPERIMENTAL=1 (`CVC_EXPERIMENTAL=1 python cyclical-visual-captioning_amd/build_hip.py --force`)")
    return fn


def lib() -> C.CDLL:
    """Load libcvc_hip.so (built by build_hip.py / __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python cyclical-visual-captioning_amd/build_hip.py` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the caption-decode hot path.")
        l = C.CDLL(LIB_PATH)
        l.cvc_block.restype = C.c_void_p
        l.cvc_block.argtypes = [C.c_char_p]
        for name, argtypes in SIGNATURES.items():
            restype = None if name in _VOID_RETURN else (C.c_longlong if name in ("cvc_train_loop_bwd_ws", "cvc_bn_workspace", "cvc_col_sum_ws") else C.c_int)
            if name in BLOCKS or name in EXPERIMENTAL:
                addr = l.cvc_block(name.encode())
                if not addr:
                    if name in BLOCKS:
                        raise RuntimeError(f"{LIB_PATH}: building block {name} is not in the library's table (stale build?)")
                    setattr(l, name, _missing(name))
                    continue
                setattr(l, name, C.CFUNCTYPE(restype, *argtypes)(addr))
                continue
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = restype
        l.cvc_version.restype = C.c_char_p
        l.cvc_version.argtypes = []
        _lib = l
    return _lib


def experimental_built() -> bool:
    """was the loaded library built with CVC_EXPERIMENTAL=1?"""
    return b"+experimental" in lib().cvc_version()


def version() -> str:
    return lib().cvc_version().decode()


# --------------------------------------------------------------------------- weights generation
# Packed copies of parameters (the decode engine's weight packs, the encoder's GRU / dense-layer operands) are keyed on the
# parameters' (data_ptr, _version).  A fused optimizer kernel and a HIP-graph replay of the training step update parameters
# WITHOUT bumping _version, so every such cache also carries this process-wide generation; whoever updates weights behind
# autograd's back bumps it (Trainer after every optimizer step, the captioner's invalidate_decode_cache()).
_weights_generation = 0


def weights_generation() -> int:
    return _weights_generation


_warned = set()


def warn_once(key: str, msg: str):
    """One warning per process and reason: a path that leaves the HIP kernels for a library module says so (and why) once."""
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn("cvc: " + msg, RuntimeWarning, stacklevel=3)


def bump_weights_generation() -> int:
    global _weights_generation
    _weights_generation += 1
    return _weights_generation


# --------------------------------------------------------------------------- deferred error words (graph-captured training steps)
# The persistent GRU kernels (csrc/gru_persistent.hip, gru_bwd_persistent.hip) separate their time steps by a barrier in device
# memory with a BOUNDED spin: when a peer workgroup never arrives (the grid is not co-resident) the kernel raises an error word in
# its sync buffer instead of hanging the GPU, and its outputs are invalid.  Eagerly the caller reads that word on the host and
# repeats the layer in the per-step form -- a host read per layer, which nothing that is captured into a HIP graph can contain.
# Deferred mode keeps the verdict ON THE DEVICE: every launch ORs its error word into ONE status word (a stream operation), the
# optimizer pass reads it and leaves the parameters, moments and step counts untouched when it is set (cvc_adam_clip_step's `skip`),
# and the trainer reads it back together with the losses once per display interval, re-running the few steps it names on the
# per-step forms (cvc.trainer.Trainer.train).  A time-out therefore costs one re-run step, never a corrupted update.
_step_status: dict = {}
_deferred_status = None


def step_status(device) -> torch.Tensor:
    """the device's status word: int32[1], non-zero = some launch of the current step reported invalid outputs"""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _step_status.get(key)
    if t is None:
        t = _step_status[key] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
    return t


def defer_errors(status) -> None:
    """status = a step_status() tensor: error words are OR-ed into it from now on (no host reads); None: back to host reads"""
    global _deferred_status
    _deferred_status = status


def errors_deferred() -> bool:
    return _deferred_status is not None


def error_word_ok(word: torch.Tensor) -> bool:
    """`word`: a one-element int32 view of a launch's error word.  Deferred mode: OR it into the status word (stream operation,
    capturable) and report True -- the verdict is taken later; otherwise read it now (host sync)."""
    if _deferred_status is not None:
        _deferred_status.bitwise_or_(word)
        return True
    return int(word) == 0


# --------------------------------------------------------------------------- per-entry-point HIP-event timing (bench.py)
_raw_fns = {}
# entry points that launch nothing (sizes, switches, handles): an event pair around them would read as ~5 us of GPU time each
_HOST_ONLY = {"cvc_tile_gemm_big", "cvc_tile_rows_alloc", "cvc_tile_gemm_plan", "cvc_col_sum_ws", "cvc_train_loop_bwd_ws", "cvc_bn_workspace", "cvc_optim_chunk_elems", "cvc_version",
              "cvc_block", "cvc_gemm_packed_split", "cvc_gemm_force_generic", "cvc_tile_gemm_loaders", "cvc_gru_persistent_waves8",
              "cvc_decode_plan_create", "cvc_decode_plan_destroy", "cvc_decode_plan_set_features", "cvc_decode_num_launches",
              "cvc_train_loop_profile", "cvc_train_loop_profile_read", "cvc_comm_unique_id", "cvc_comm_init", "cvc_comm_destroy"}


TIMED_SHAPES: dict = {}      # enable_timers(): entry point -> the sizes of each of its launches (what bench.py prices a role's work with)
_SHAPE_ARGS = {
    "cvc_tile_gemm": lambda a: (int(a[3]), int(a[4]), int(a[5]), int(a[6])),            # (wb, xb, x_mblk_stride, K, M, N, ksplit, ...)
    "cvc_tile_pack_rows_any": lambda a: (int(a[2]), int(a[3])),                         # (x, ldx, M, K, ...)
    "cvc_tile_pack_cols": lambda a: (int(a[2]), int(a[3])),                             # (x, ldx, S, C, ...)
    "cvc_pack_lstm_segs": lambda a: (sum(int(a[2][i]) for i in range(int(a[3]))), int(a[4])),     # (ws, lds, widths, nseg, R, ...) -> (K, R)
    # (kind, q, w_a, inv_temp, proj, ctx, attn, d_ctx, d_fm, nclip, nq, n, A, R, d_scores, d_q, d_w_part, d_proj, d_ctxfeat)
    "cvc_attn_bwd": lambda a: (int(a[9]), int(a[10]), int(a[11]), int(a[12]), int(a[13]), bool(a[7]), bool(a[17]), bool(a[18])),
}


def enable_timers() -> dict:
    """Bracket every C-ABI launch with a HIP-event pair on the launch stream (torch's current stream, which is the stream
    handed to the entry point).  Returns the dict that fills up: entry-point name -> [(start_event, end_event)]; the (K, M, N) of
    every cvc_tile_gemm launch go to TIMED_SHAPES (what bench.py prices the dense products' roofline with).
    Measurement aid for bench.py (--mode train); the product path never enables it."""
    l = lib()
    timers: dict = {}
    TIMED_SHAPES.clear()
    if _raw_fns:
        disable_timers()
    for name in SIGNATURES:
        if name in _HOST_ONLY:
            continue
        raw = getattr(l, name)
        _raw_fns[name] = raw

        def timed(*args, _raw=raw, _name=name):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = _raw(*args)
            e1.record()
            timers.setdefault(_name, []).append((e0, e1))
            pick = _SHAPE_ARGS.get(_name)
            if pick is not None:
                TIMED_SHAPES.setdefault(_name, []).append(pick(args))
            return rc
        setattr(l, name, timed)
    return timers


def disable_timers():
    l = lib()
    for name, raw in _raw_fns.items():
        setattr(l, name, raw)
    _raw_fns.clear()


def _check(rc: int, name: str):
    if rc != 0:
        kind = {-1: "bad argument (size/alignment precondition)", -2: "dimension not covered by the kernel templates",
                -3: "librccl.so could not be opened"}.get(
            rc, f"hipError_t {rc}")
        raise RuntimeError(f"{name} failed: {kind}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Optional[torch.Tensor], dtype=torch.float32, name="tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"cvc.hip: {name} must live on the GPU (no CPU fallback for the hot path)")
    if t.dtype != dtype:
        raise RuntimeError(f"cvc.hip: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"cvc.hip: {name} must be contiguous")
    return t.data_ptr()


def _mask(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """bool / uint8 mask -> contiguous uint8 view (one byte per element)."""
    if t is None:
        return None
    if t.dtype == torch.bool:
        t = t.contiguous().view(torch.uint8)
    elif t.dtype != torch.uint8:
        t = (t != 0).contiguous().view(torch.uint8)
    return t.contiguous()


def gsk_plan(ntile: Sequence[int], nchunk: Sequence[int], nwg: int) -> dict:
    """Host arithmetic of a grouped stream-K launch (cvc_gsk_plan): units per workgroup, each group's first unit and the
    segments per tile its slab must hold."""
    n = len(ntile)
    arr = lambda v: (C.c_int * n)(*[int(x) for x in v])
    U, unit0, maxseg = C.c_int(), (C.c_int * n)(), (C.c_int * n)()
    _check(lib().cvc_gsk_plan(arr(ntile), arr(nchunk), n, int(nwg), C.byref(U), unit0, maxseg), "cvc_gsk_plan")
    return dict(U=U.value, unit0=list(unit0), maxseg=list(maxseg), nwg=int(nwg))


# --------------------------------------------------------------------------- attention
def attn_fwd(kind: int, q: torch.Tensor, w_a: Optional[torch.Tensor], b_a: Optional[torch.Tensor], inv_temp: float,
             sets: Sequence[dict], nclip: int, nq: int, want_ctx: bool = True, want_sum: bool = False):
    """sets: dicts with proj [nclip,n,A], ctx [nclip,n,R], mask, frame_mask.  Returns
    ([(scores, frame_masked, attn, ctx_out)], ctx_sum)."""
    rows, A = q.shape
    assert rows == nclip * nq
    R = sets[0]["ctx"].shape[2]
    arr = (AttnSet * len(sets))()
    outs, keep = [], []
    for i, s in enumerate(sets):
        n = s["proj"].shape[1]
        assert s["proj"].shape == (nclip, n, A) and s["ctx"].shape == (nclip, n, R), (s["proj"].shape, s["ctx"].shape)
        m, fmk = _mask(s.get("mask")), _mask(s.get("frame_mask"))
        if m is not None:
            assert m.shape == (nclip, n)
        if fmk is not None:
            assert fmk.shape == (rows, n)
        scores = torch.empty(rows, n, device=q.device, dtype=torch.float32)
        attn = torch.empty_like(scores)
        fm = torch.empty_like(scores) if fmk is not None else None
        ctx_out = torch.empty(rows, R, device=q.device, dtype=torch.float32) if want_ctx else None
        arr[i] = AttnSet(_dev(s["proj"], name="proj"), _dev(s["ctx"], name="ctx"), _dev(m, torch.uint8), _dev(fmk, torch.uint8),
                         _dev(scores), _dev(fm), _dev(attn), _dev(ctx_out), n, 4 if s.get("sentinel") else 0)
        keep += [m, fmk]
        outs.append((scores, fm, attn, ctx_out))
    ctx_sum = torch.empty(rows, R, device=q.device, dtype=torch.float32) if want_sum else None
    _check(lib().cvc_attn_fwd(kind, _dev(q, name="q"), _dev(w_a), _dev(b_a), float(inv_temp), arr, len(sets), nclip, nq, A, R,
                              _dev(ctx_sum), _stream()), "cvc_attn_fwd")
    return outs, ctx_sum


def attn_bwd(kind: int, q, w_a, inv_temp: float, proj, ctx, attn, d_ctx, d_fm, nclip: int, nq: int,
             want_d_proj: bool, want_d_ctxfeat: bool, want_d_w: bool):
    rows, A = q.shape
    n, R = proj.shape[1], ctx.shape[2]
    d_scores = torch.empty(rows, n, device=q.device, dtype=torch.float32)
    d_q = torch.empty_like(q)
    d_w_part = torch.empty_like(q) if (want_d_w and kind == ATTN_ADDITIVE) else None
    d_proj = torch.zeros_like(proj) if want_d_proj else None
    d_ctxfeat = torch.zeros_like(ctx) if (want_d_ctxfeat and d_ctx is not None) else None
    _check(lib().cvc_attn_bwd(kind, _dev(q), _dev(w_a), float(inv_temp), _dev(proj), _dev(ctx), _dev(attn), _dev(d_ctx),
                              _dev(d_fm), nclip, nq, n, A, R, _dev(d_scores), _dev(d_q), _dev(d_w_part), _dev(d_proj),
                              _dev(d_ctxfeat), _stream()), "cvc_attn_bwd")
    return d_scores, d_q, d_w_part, d_proj, d_ctxfeat


# --------------------------------------------------------------------------- skinny GEMM / LSTM cell
def _segs(segs: Sequence[dict], M: int):
    """segs: dicts {x: [M,k] (or table with idx), w: [Nout, k] (a column slice is fine), idx, relu}."""
    arr = (GemmSeg * len(segs))()
    for i, s in enumerate(segs):
        x, w = s["x"], s["w"]
        k = w.shape[1]
        if x.stride(-1) != 1 or w.stride(-1) != 1:
            raise RuntimeError("cvc.hip: GEMM segments must have unit inner stride")
        idx = s.get("idx")
        if idx is None:
            assert x.shape[0] == M and x.shape[1] == k, (x.shape, M, k)
        else:
            assert idx.shape == (M,) and x.shape[1] == k
            _dev(idx, torch.int64, "idx")
        for t_ in (x, w):
            if not t_.is_cuda or t_.dtype != torch.float32:
                raise RuntimeError("cvc.hip: GEMM operands must be fp32 GPU tensors (no CPU fallback)")
        arr[i] = GemmSeg(x.data_ptr(), None if idx is None else idx.data_ptr(), w.data_ptr(), k, x.stride(0), w.stride(0),
                         1 if s.get("relu") else 0)
    return arr


def linear_fwd(segs: Sequence[dict], bias: Optional[torch.Tensor], M: int, Nout: int, bias2=None, out=None):
    dev = segs[0]["w"].device
    y = out if out is not None else torch.empty(M, Nout, device=dev, dtype=torch.float32)
    arr = _segs(segs, M)
    _check(lib().cvc_linear_fwd(arr, len(segs), _dev(bias), _dev(bias2), M, Nout, _dev(y), y.stride(0), _stream()),
           "cvc_linear_fwd")
    return y


_split_mode_cache = None


def gemm_packed_split(mode: int) -> int:
    """Packed decode GEMMs: 0 = plain fp32 MFMA, 1 = bf16x3-split products on the bf16 MFMA with 4 waves per
    workgroup, 2 = the same with 8 waves (two per SIMD).  -> previous setting.  A negative mode only queries (cached on the
    host: the backward pass asks once per LSTM step)."""
    global _split_mode_cache
    if mode < 0 and _split_mode_cache is not None:
        return _split_mode_cache
    prev = int(lib().cvc_gemm_packed_split(int(mode)))
    _split_mode_cache = prev if mode < 0 else (2 if mode > 2 else int(mode))
    return prev


def gemm_force_generic(on: bool) -> bool:
    return bool(lib().cvc_gemm_force_generic(1 if on else 0))


def lstm_cell_fwd(segs: Sequence[dict], b_ih, b_hh, c_prev, want_gates: bool = False, h_out=None, c_out=None,
                  gate_bias=None):
    M, R = c_prev.shape
    h = h_out if h_out is not None else torch.empty_like(c_prev)
    c = c_out if c_out is not None else torch.empty_like(c_prev)
    gates = torch.empty(M, 4 * R, device=c_prev.device, dtype=torch.float32) if want_gates else None
    arr = _segs(segs, M)
    _check(lib().cvc_lstm_cell_fwd(arr, len(segs), _dev(b_ih), _dev(b_hh), _dev(gate_bias), _dev(c_prev), M, R, _dev(h),
                                   _dev(c), _dev(gates), _stream()), "cvc_lstm_cell_fwd")
    return h, c, gates


# ---- training form of the packed gate GEMM: the weights are re-packed once per optimizer step, the cell's inputs per call
_train_generation = 0


def lstm_train_new_step() -> None:
    """Called by the model at the start of every training forward: the packed LSTM weights are rebuilt at their first use of
    the step whatever their autograd version says (a captured training step must contain the re-pack launch)."""
    global _train_generation
    _train_generation += 1
    _step_operands.clear()


# tile-GEMM operand packs of WEIGHTS (or column slices of them), shared by every product of one training step that reads the same
# weight the same way: the vocabulary head is multiplied in both loops and read again by both backward-data products, the embedding
# columns of the attention cell by both loops' hoisted products and both d_emb products.  Cleared at the start of every forward.
_step_operands = {}


def weight_operand(w: torch.Tensor, kmajor: bool = False) -> "TileOperand":
    key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), bool(kmajor), _train_generation, w._version)
    op = _step_operands.get(key)
    if op is None:
        op = _step_operands[key] = TileOperand(w.detach(), kmajor)
    return op


def lstm_train_ok(M: int, R: int, widths: Sequence[int]) -> bool:
    return 1 <= M <= 64 and R % 8 == 0 and all(w % 4 == 0 and w >= 4 for w in widths) and sum(widths) % 32 == 0


def lstm_train_pack(w_ih: torch.Tensor, w_hh: torch.Tensor, cols: Optional[Sequence] = None) -> torch.Tensor:
    """Packed copy [R/8][K/4][32][4] of an LSTM cell's weights for cvc_packed_lstm_train_fwd: all of weight_ih, or only its
    column ranges `cols` = ((col0, width), ...) (the recurrent inputs of a hoisted cell), followed by weight_hh.  The packs live
    ON the weight tensor object (they die with it: no table keyed by addresses that a later tensor could reuse), one per `cols`,
    and are rebuilt when the step generation, either tensor's version counter or either address changed."""
    R = w_hh.shape[1]
    key = None if cols is None else tuple((int(c0), int(n)) for c0, n in cols)
    stamp = (_train_generation, w_ih._version, w_hh._version, w_ih.data_ptr(), w_hh.data_ptr(), tuple(w_ih.shape))
    packs = getattr(w_ih, "_cvc_train_packs", None)
    if packs is None:
        packs = w_ih._cvc_train_packs = {}
    ent = packs.get(key)
    if ent is not None and ent[1] == stamp:
        return ent[0]
    K_hh = w_hh.shape[1]
    ranges = [(0, w_ih.shape[1])] if key is None else list(key)
    K = sum(n for _, n in ranges) + K_hh
    shape = (R // 8, K // 4, 32, 4)
    wp = ent[0] if (ent is not None and tuple(ent[0].shape) == shape and ent[0].device == w_ih.device) else \
        torch.empty(*shape, device=w_ih.device, dtype=torch.float32)
    n = len(ranges) + 1
    ptrs = (_P * n)(*[w_ih.data_ptr() + 4 * c0 for c0, _ in ranges], w_hh.data_ptr())
    lds = (_LL * n)(*[w_ih.stride(0)] * len(ranges), w_hh.stride(0))
    ws = (_I * n)(*[nn_ for _, nn_ in ranges], K_hh)
    _check(lib().cvc_pack_lstm_segs(ptrs, lds, ws, n, R, _dev(wp), _stream()), "cvc_pack_lstm_segs")
    packs[key] = (wp, stamp)
    return wp


def linear_train_pack(w: torch.Tensor) -> torch.Tensor:
    """Packed copy [Nout/32][K/4][32][4] of a linear layer's weight [Nout, K] for cvc_packed_linear_fwd (the training loop's
    h2attn): lives on the weight tensor like the LSTM packs, rebuilt when the step generation / version / address changed."""
    n, k = w.shape
    assert n % 32 == 0 and k % 32 == 0
    stamp = (_train_generation, w._version, w.data_ptr(), (n, k))
    ent = getattr(w, "_cvc_linear_pack", None)
    if ent is not None and ent[1] == stamp:
        return ent[0]
    wp = ent[0] if (ent is not None and tuple(ent[0].shape) == (n // 32, k // 4, 32, 4) and ent[0].device == w.device) else \
        torch.empty(n // 32, k // 4, 32, 4, device=w.device, dtype=torch.float32)
    with torch.no_grad():
        wp.copy_(w.detach().view(n // 32, 32, k // 4, 4).permute(0, 2, 1, 3))       # one strided copy
    w._cvc_linear_pack = (wp, stamp)
    return wp


def pack_quad_segs(xs: Sequence[torch.Tensor]) -> torch.Tensor:
    """Row-major segments [M <= 64, k_s] -> one quad-layout operand [sum k_s / 4][64][4]."""
    n, M = len(xs), xs[0].shape[0]
    ptrs = (_P * n)(*[x.data_ptr() for x in xs])
    lds = (_LL * n)(*[x.stride(0) for x in xs])
    ws = (_I * n)(*[x.shape[1] for x in xs])
    xq = torch.empty(sum(x.shape[1] for x in xs) // 4, 64, 4, device=xs[0].device, dtype=torch.float32)
    _check(lib().cvc_pack_quad_segs(ptrs, lds, ws, n, M, _dev(xq), _stream()), "cvc_pack_quad_segs")
    return xq


def lstm_cell_train_fwd(xs: Sequence[torch.Tensor], h_prev, c_prev, wp, b_ih, b_hh, want_gates: bool = True, copies: int = 1,
                        gate_pre: Optional[torch.Tensor] = None, drop=None):
    """nn.LSTMCell forward on the packed gate GEMM (cvc_packed_lstm_train_fwd) from the pack `lstm_train_pack` built:
    -> h (a tuple of `copies` identical tensors when copies > 1), c, activated gates (or None).  gate_pre [M, 4R]: the
    contribution of input segments multiplied beforehand for all steps (hoisted cell); xs are then the remaining segments.
    drop = (rng_state, site, p): one more output tensor, nn.Dropout(h') with the in-kernel mask (copies <= 2 then):
    -> (h or tuple of h, h_dropped), c, gates."""
    M, R = c_prev.shape
    xq = pack_quad_segs([*xs, h_prev])
    c = torch.empty_like(c_prev)
    gates = torch.empty(M, 4 * R, device=c_prev.device, dtype=torch.float32) if want_gates else None
    if drop is not None:
        state, site, p = drop
        hs = [torch.empty_like(c_prev) for _ in range(max(1, min(2, copies)))]
        hd = torch.empty_like(c_prev)
        _check(lib().cvc_packed_lstm_train_drop_fwd(_dev(wp), _dev(xq), xq.shape[0] * 4, _dev(b_ih), _dev(b_hh), _dev(gate_pre),
                                                    _dev(c_prev), M, R, _dev(hs[0]), _dev(c), _dev(gates),
                                                    _dev(hs[1]) if len(hs) > 1 else None, _dev(hd), _rng_ptr(state), int(site),
                                                    float(p), _stream()), "cvc_packed_lstm_train_drop_fwd")
        return ((hs[0] if copies <= 1 else tuple(hs)), hd), c, gates
    hs = [torch.empty_like(c_prev) for _ in range(max(1, min(3, copies)))]
    tail = (_dev(c_prev), M, R, _dev(hs[0]), _dev(c), _dev(gates), _dev(hs[1]) if len(hs) > 1 else None,
            _dev(hs[2]) if len(hs) > 2 else None, _stream())
    if gate_pre is None:
        _check(lib().cvc_packed_lstm_train_fwd(_dev(wp), _dev(xq), xq.shape[0] * 4, _dev(b_ih), _dev(b_hh), *tail), "cvc_packed_lstm_train_fwd")
    else:
        _check(lib().cvc_packed_lstm_train_pre_fwd(_dev(wp), _dev(xq), xq.shape[0] * 4, _dev(b_ih), _dev(b_hh), _dev(gate_pre), *tail),
               "cvc_packed_lstm_train_pre_fwd")
    return (hs[0] if copies <= 1 else tuple(hs)), c, gates


def lstm_pointwise_bwd(d_h, d_c, gates, c_prev, c_new, want_quad=False, d_h2=None, d_h3=None, drop3=None):
    """-> d_gates [M,4R], d_c_prev [M,R] (+ d_gates in the quad layout [R][64][4] for linear_nn when asked).  d_h2, d_h3: the
    gradients of further copies of h' (summed with d_h inside the kernel).  drop3 = (rng_state, site, p): d_h3 is the gradient
    of the copy that left the cell through the fused dropout and takes that mask."""
    M, R = c_prev.shape
    d_gates = torch.empty_like(gates)
    d_c_prev = torch.empty_like(c_prev)
    d_gates_q = torch.empty(R, 64, 4, device=gates.device, dtype=torch.float32) if want_quad else None
    if drop3 is not None and d_h3 is not None:
        state, site, p = drop3
        _check(lib().cvc_lstm_pointwise_bwd3_drop(_dev(d_h), _dev(d_h2), _dev(d_h3), _rng_ptr(state), int(site), float(p), _dev(d_c),
                                                  _dev(gates), _dev(c_prev), _dev(c_new), M, R, _dev(d_gates), _dev(d_c_prev),
                                                  _dev(d_gates_q), _stream()), "cvc_lstm_pointwise_bwd3_drop")
        return (d_gates, d_c_prev, d_gates_q) if want_quad else (d_gates, d_c_prev)
    _check(lib().cvc_lstm_pointwise_bwd3(_dev(d_h), _dev(d_h2), _dev(d_h3), _dev(d_c), _dev(gates), _dev(c_prev), _dev(c_new), M, R,
                                         _dev(d_gates), _dev(d_c_prev), _dev(d_gates_q), _stream()), "cvc_lstm_pointwise_bwd3")
    return (d_gates, d_c_prev, d_gates_q) if want_quad else (d_gates, d_c_prev)


def linear_nn_ok(M, K, ranges):
    """Shapes cvc_linear_nn_fwd accepts (everything else keeps the library GEMM)."""
    return M <= 64 and K % 8 == 0 and all(
        n >= 4 and n % 4 == 0 and w.stride(0) % 4 == 0 and w.stride(1) == 1 and c0 % 4 == 0 and w.data_ptr() % 16 == 0
        for (w, c0, n) in ranges)


def pack_quad(x: torch.Tensor) -> torch.Tensor:
    """[M <= 64, K] row-major -> [K/4][64][4] (linear_nn's dY layout)."""
    M, K = x.shape
    xq = torch.empty(K // 4, 64, 4, device=x.device, dtype=torch.float32)
    _check(lib().cvc_pack_quad(_dev(x), x.stride(0), M, K, _dev(xq), _stream()), "cvc_pack_quad")
    return xq


def linear_nn(dy_q, M, K, ranges, ksplit=None):
    """dX_s[M, n_s] = dY[M, K] @ W_s[:, c0_s : c0_s + n_s] for every (W_s, c0_s, n_s) in `ranges`, one launch.
    dy_q: dY in the quad layout [K/4][64][4] (lstm_pointwise_bwd(want_quad=True))."""
    arr = (NNSeg * len(ranges))()
    outs, keep = [], []
    slabs = 0
    for i, (w, c0, n) in enumerate(ranges):
        assert w.shape[0] == K and w.is_cuda and w.dtype == torch.float32
        out = torch.empty(M, n, device=w.device, dtype=torch.float32)
        arr[i] = NNSeg(w.data_ptr() + 4 * c0, out.data_ptr(), w.stride(0), n, n)
        outs.append(out)
        slabs += (n + 127) // 128
    # K slices: one resident round of workgroups and >= 16 rows (two double groups) per wave in every slice -- a short K on few
    # column slabs (h2attn's backward) is a latency-bound launch that wants the whole chip, not long K loops
    if ksplit is None:
        # the split-product kernel holds one workgroup per CU (256 on the chip), the fp32-MFMA kernel two
        resident = 256 if gemm_packed_split(-1) != 0 else 512
        ksplit = max(1, min(K // 8 // 16, resident // slabs))
    ws = torch.empty(ksplit * M * slabs * 128, device=dy_q.device, dtype=torch.float32) if ksplit > 1 else None
    _check(lib().cvc_linear_nn_fwd(_dev(dy_q), K, M, arr, len(ranges), ksplit, _dev(ws), _stream()), "cvc_linear_nn_fwd")
    return outs


# --------------------------------------------------------------------------- embedding / vocabulary head
def embed_relu_fwd(table, idx, drop=None):
    M, E = idx.shape[0], table.shape[1]
    out = torch.empty(M, E, device=table.device, dtype=torch.float32)
    _check(lib().cvc_embed_relu_fwd(_dev(table), _dev(idx, torch.int64), _dev(drop), M, E, _dev(out), _stream()),
           "cvc_embed_relu_fwd")
    return out


def _rng_ptr(state: torch.Tensor):
    """device pointer of a dropout generator state (cvc/dropout.py: >= 3 int32 words {seed_lo, seed_hi, step})"""
    if not (state.is_cuda and state.dtype == torch.int32 and state.is_contiguous() and state.numel() >= 3):
        raise RuntimeError("dropout generator state: contiguous int32 device tensor of >= 3 words")
    return state.data_ptr()


def embed_relu_rng_fwd(table, idx, state, site: int, p: float):
    """relu(table[idx]) * in-kernel keep-mask of (site, element index) (cvc_embed_relu_rng_fwd)"""
    M, E = idx.shape[0], table.shape[1]
    out = torch.empty(M, E, device=table.device, dtype=torch.float32)
    _check(lib().cvc_embed_relu_rng_fwd(_dev(table), _dev(idx, torch.int64), _rng_ptr(state), int(site), float(p), M, E, _dev(out),
                                        _stream()), "cvc_embed_relu_rng_fwd")
    return out


_order_cache = {}


def _embed_order(idx: torch.Tensor) -> torch.Tensor:
    """stable argsort of the word indices (rows grouped by word), cached for the step: loops A and C embed the same words"""
    key = (idx.data_ptr(), idx.numel(), idx._version, _train_generation)
    o = _order_cache.get(key)
    if o is None:
        if len(_order_cache) > 16:
            _order_cache.clear()
        o = _order_cache[key] = (stable_order(idx), idx)      # (idx kept alive: its address is the key)
    return o[0]


def stable_order(idx: torch.Tensor) -> torch.Tensor:
    """torch.argsort(idx, stable=True) for a 1-d int64 tensor (one launch, cvc_stable_order, up to 7168 keys)"""
    idx = idx.contiguous()
    if idx.dim() != 1 or idx.dtype != torch.int64 or not 1 <= idx.numel() <= 7168:
        return torch.argsort(idx, stable=True)
    order = torch.empty_like(idx)
    _check(lib().cvc_stable_order(_dev(idx, torch.int64), idx.numel(), _dev(order, torch.int64), _stream()), "cvc_stable_order")
    return order


def col_sum(x: torch.Tensor, out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [S, n] (unit inner stride) -> sum over rows [n], written to `out` (and `out2`) when given: bias gradients"""
    assert x.dim() == 2 and x.dtype == torch.float32
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        x = x.contiguous()
    S, n = x.shape
    if out is None:
        out = torch.empty(n, device=x.device, dtype=torch.float32)
    assert out.is_contiguous() and out.numel() == n and (out2 is None or (out2.is_contiguous() and out2.numel() == n))
    if not x.is_cuda:
        raise RuntimeError("cvc.hip: col_sum input must live on the GPU (no CPU fallback for the hot path)")
    nws = int(lib().cvc_col_sum_ws(S, n))
    ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
    _check(lib().cvc_col_sum(x.data_ptr(), x.stride(0), S, n, _dev(out), _dev(out2), _dev(ws), _stream()), "cvc_col_sum")
    return out


def embed_relu_rng_bwd(table, idx, state, site: int, p: float, d_out, d_table=None):
    """d_table given: ACCUMULATED into (several lookups of one table add up in one buffer)"""
    if d_table is None:
        d_table = torch.zeros_like(table)
    order = _embed_order(idx)
    ws = torch.empty(idx.shape[0], table.shape[1], device=table.device, dtype=torch.float32)
    _check(lib().cvc_embed_relu_rng_bwd(_dev(table), _dev(idx, torch.int64), _dev(order, torch.int64), _rng_ptr(state), int(site),
                                        float(p), _dev(d_out), idx.shape[0], table.shape[1], _dev(d_table), _dev(ws), _stream()),
           "cvc_embed_relu_rng_bwd")
    return d_table


def dropout_rng(x, state, site: int, p: float):
    """x * in-kernel keep-mask of (site, flat element index) (cvc_dropout_rng); its own backward on the gradient"""
    y = torch.empty_like(x)
    _check(lib().cvc_dropout_rng(_dev(x), x.numel(), _rng_ptr(state), int(site), float(p), _dev(y), _stream()), "cvc_dropout_rng")
    return y


def embed_relu_bwd(table, idx, drop, d_out, d_table=None):
    if d_table is None:
        d_table = torch.zeros_like(table)
    order = _embed_order(idx)                                    # rows grouped by word, original order inside a group
    ws = torch.empty(idx.shape[0], table.shape[1], device=table.device, dtype=torch.float32)
    _check(lib().cvc_embed_relu_bwd(_dev(table), _dev(idx, torch.int64), _dev(order, torch.int64), _dev(drop), _dev(d_out),
                                    idx.shape[0], table.shape[1], _dev(d_table), _dev(ws), _stream()), "cvc_embed_relu_bwd")
    return d_table


def log_softmax_fwd(logits, out=None):
    M, V = logits.shape
    out = out if out is not None else torch.empty_like(logits)
    _check(lib().cvc_log_softmax_fwd(_dev(logits), M, V, _dev(out), _stream()), "cvc_log_softmax_fwd")
    return out


def log_softmax_bwd(logp, d_logp):
    M, V = logp.shape
    d = torch.empty_like(logp)
    _check(lib().cvc_log_softmax_bwd(_dev(logp), _dev(d_logp), M, V, _dev(d), _stream()), "cvc_log_softmax_bwd")
    return d


def nll_bwd(target, w, g, V: int):
    M = target.shape[0]
    d = torch.empty(M, V, device=w.device, dtype=torch.float32)
    _check(lib().cvc_nll_bwd(_dev(target, torch.int64), _dev(w), _dev(g), M, V, _dev(d), _stream()), "cvc_nll_bwd")
    return d


def top2_unk(logits, unk_idx: int, word_out: torch.Tensor, word_stride: int = 1, logprob: Optional[torch.Tensor] = None):
    """word_out: int64 tensor whose element m*word_stride (from its data_ptr) receives row m's word."""
    M, V = logits.shape
    assert word_out.dtype == torch.int64 and word_out.is_cuda
    _check(lib().cvc_top2_unk(_dev(logits), M, V, int(unk_idx), word_out.data_ptr(), int(word_stride), _dev(logprob),
                              _stream()), "cvc_top2_unk")


def nll_fwd(logp, target, w):
    M, V = logp.shape
    loss = torch.zeros(1, device=logp.device, dtype=torch.float32)
    _check(lib().cvc_nll_fwd(_dev(logp), _dev(target, torch.int64), _dev(w), M, V, _dev(loss), _stream()), "cvc_nll_fwd")
    return loss


def vocab_nll_fwd(logits, target, w, want_argmax=True):
    """-> loss_sum [1], lse [M], argmax [M] int64 (or None)"""
    M, V = logits.shape
    dev = logits.device
    lse = torch.empty(M, device=dev, dtype=torch.float32)
    row_loss = torch.empty(M, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    amax = torch.empty(M, device=dev, dtype=torch.int64) if want_argmax else None
    _check(lib().cvc_vocab_nll_fwd(_dev(logits), _dev(target, torch.int64), _dev(w), M, V, _dev(lse), _dev(amax, torch.int64),
                                   _dev(row_loss), _dev(loss), _stream()), "cvc_vocab_nll_fwd")
    return loss, lse, amax


def vocab_head_nll_fwd(parts, bias, target, w):
    """parts [ks, M, V]: the vocabulary head's K-slice slabs.  -> loss_sum [1], argmax [M], pre [M, V] (aliasing slab 0)"""
    ks, M, V = parts.shape
    dev = parts.device
    row_loss = torch.empty(M, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    amax = torch.empty(M, device=dev, dtype=torch.int64)
    pre = parts[0]
    _check(lib().cvc_vocab_head_nll_fwd(_dev(parts), ks, M * V, V, _dev(bias), _dev(target, torch.int64), _dev(w), M, V, pre.data_ptr(), V,
                                        _dev(amax, torch.int64), _dev(row_loss), _dev(loss), _stream()), "cvc_vocab_head_nll_fwd")
    return loss, amax, pre


def scale_by_scalar(x, g):
    """g[0] * x with g a device scalar tensor (no host read)"""
    y = torch.empty_like(x)
    _check(lib().cvc_scale_by_scalar(_dev(x), _dev(g), x.numel(), _dev(y), _stream()), "cvc_scale_by_scalar")
    return y


def vocab_nll_bwd(logits, lse, target, w, g):
    M, V = logits.shape
    d = torch.empty_like(logits)
    _check(lib().cvc_vocab_nll_bwd(_dev(logits), _dev(lse), _dev(target, torch.int64), _dev(w), _dev(g), M, V, _dev(d),
                                   _stream()), "cvc_vocab_nll_bwd")
    return d


def grounder_fwd(xt, feats, bias, mask):
    B, T, G = xt.shape
    N = feats.shape[1]
    out = torch.empty(B, T, N, device=xt.device, dtype=torch.float32)
    m = _mask(mask)
    _check(lib().cvc_grounder_fwd(_dev(xt), _dev(feats), _dev(bias), _dev(m, torch.uint8), B, T, N, G, _dev(out), _stream()),
           "cvc_grounder_fwd")
    return out


def grounder_bwd(d, xt, feats, want_xt: bool, want_feats: bool):
    B, T, G = xt.shape
    N = feats.shape[1]
    d_xt = torch.empty_like(xt) if want_xt else None
    d_feats = torch.empty_like(feats) if want_feats else None
    _check(lib().cvc_grounder_bwd(_dev(d), _dev(xt), _dev(feats), B, T, N, G, _dev(d_xt), _dev(d_feats), _stream()), "cvc_grounder_bwd")
    return d_xt, d_feats


def beam_select(logits, score_in, done_in, B: int, beam: int, unk_idx: int, first_step: bool):
    V = logits.shape[1]
    dev = logits.device
    parent = torch.empty(B * beam, dtype=torch.int64, device=dev)
    word = torch.empty(B * beam, dtype=torch.int64, device=dev)
    score = torch.empty(B * beam, dtype=torch.float32, device=dev)
    done = torch.empty(B * beam, dtype=torch.uint8, device=dev)
    ws = torch.empty(17 * B * beam, dtype=torch.float32, device=dev)
    _check(lib().cvc_beam_select(_dev(logits), _dev(score_in), _dev(done_in, torch.uint8), B, beam, V, int(unk_idx),
                                 1 if first_step else 0, parent.data_ptr(), word.data_ptr(), _dev(score),
                                 _dev(done, torch.uint8), _dev(ws), _stream()), "cvc_beam_select")
    return parent, word, score, done


def beam_select_parts(parts, bias, score_in, done_in, B: int, beam: int, unk_idx: int, first_step: bool):
    """beam_select over logits that still are the K-slice slabs [nparts, rows, V] of the vocabulary GEMM (+ bias [V]): summed while
    the rows are scanned, in the finishing pass's order (cvc_beam_select_parts)"""
    nparts, rows, V = parts.shape
    dev = parts.device
    parent = torch.empty(B * beam, dtype=torch.int64, device=dev)
    word = torch.empty(B * beam, dtype=torch.int64, device=dev)
    score = torch.empty(B * beam, dtype=torch.float32, device=dev)
    done = torch.empty(B * beam, dtype=torch.uint8, device=dev)
    ws = torch.empty(17 * B * beam, dtype=torch.float32, device=dev)
    _check(lib().cvc_beam_select_parts(_dev(parts), nparts, rows * V, _dev(bias), _dev(score_in), _dev(done_in, torch.uint8), B, beam, V,
                                       int(unk_idx), 1 if first_step else 0, parent.data_ptr(), word.data_ptr(), _dev(score),
                                       _dev(done, torch.uint8), _dev(ws), _stream()), "cvc_beam_select_parts")
    return parent, word, score, done


def gather_rows(src, parent, beam: int):
    rows, width = src.shape
    dst = torch.empty_like(src)
    _check(lib().cvc_gather_rows(_dev(src), _dev(parent, torch.int64), rows, beam, width, _dev(dst), _stream()),
           "cvc_gather_rows")
    return dst


# --------------------------------------------------------------------------- label glue / supervised attention criteria
def bbox_overlaps(rois: torch.Tensor, gt: torch.Tensor, frm_mask: torch.Tensor, pnt_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """IoU [B, N, K] of proposals [B, N, >= 4] vs ground-truth boxes [B, K, >= 4] (cvc_bbox_overlaps_fwd); frm_mask [B, N, K] and
    pnt_mask [B, N] (nullable) zero the overlap where set."""
    B, N, K = frm_mask.shape
    rois, gt = rois.contiguous(), gt.contiguous()
    fm = _mask(frm_mask)
    pm = None
    if pnt_mask is not None:       # [B, N], any row stride (the trainer's pnt_mask[:, 1:]): a byte view, no copy
        pm = pnt_mask.view(torch.uint8) if pnt_mask.dtype == torch.bool else (pnt_mask if pnt_mask.dtype == torch.uint8 else (pnt_mask != 0).view(torch.uint8))
        if pm.stride(1) != 1:
            pm = pm.contiguous()
    ov = torch.empty(B, N, K, device=rois.device, dtype=torch.float32)
    _check(lib().cvc_bbox_overlaps_fwd(_dev(rois), rois.shape[2], _dev(gt), gt.shape[2], _dev(fm, torch.uint8),
                                       None if pm is None else pm.data_ptr(), 0 if pm is None else pm.stride(0), B, N, K, _dev(ov),
                                       _stream()), "cvc_bbox_overlaps_fwd")
    return ov


def label_glue(ov: torch.Tensor, box_mask_kt: torch.Tensor, frm_mask: torch.Tensor, pnt_mask: torch.Tensor, want_steps: bool = True):
    """ov [B, N, K]; box_mask_kt: bool view [B, K, T] (any strides) of mask_boxes[:, 0, :, 1:T+1]; frm_mask [B, N, K]; pnt_mask
    [B, N + 1].  -> roi_labels [B, T, N] bool, frm_mask_output [B, T, N + 1] bool, step_fmask [T, B, N] bool (or None)."""
    B, N, K = ov.shape
    T = box_mask_kt.shape[2]
    bm = box_mask_kt if box_mask_kt.dtype == torch.uint8 else box_mask_kt.view(torch.uint8) if box_mask_kt.dtype == torch.bool else (box_mask_kt != 0).view(torch.uint8)
    fm, pm = _mask(frm_mask), _mask(pnt_mask)
    labels = torch.empty(B, T, N, device=ov.device, dtype=torch.uint8)
    fmo = torch.empty(B, T, N + 1, device=ov.device, dtype=torch.uint8)
    steps = torch.empty(T, B, N, device=ov.device, dtype=torch.uint8) if want_steps else None
    _check(lib().cvc_label_glue_fwd(_dev(ov), bm.data_ptr(), bm.stride(0), bm.stride(1), bm.stride(2), _dev(fm, torch.uint8),
                                    _dev(pm, torch.uint8), B, N, K, T, _dev(labels, torch.uint8), _dev(fmo, torch.uint8),
                                    _dev(steps, torch.uint8), _stream()), "cvc_label_glue_fwd")
    return labels.view(torch.bool), fmo.view(torch.bool), (None if steps is None else steps.view(torch.bool))


def attn_nll_fwd(x0: torch.Tensor, x1: Optional[torch.Tensor], target: torch.Tensor):
    """-> loss [2] (or [1]), workspace (kept for the backward)"""
    B, T, N = x0.shape
    for x in (x0, x1):
        if x is not None and not (x.is_cuda and x.dtype == torch.float32 and x.stride(2) == 1):
            raise RuntimeError("cvc.hip.attn_nll: fp32 GPU tensors with unit inner stride (no CPU fallback)")
    tg = _mask(target)
    ws = torch.empty(5 * B * T + 1, device=x0.device, dtype=torch.float32)
    loss = torch.empty(2 if x1 is not None else 1, device=x0.device, dtype=torch.float32)
    _check(lib().cvc_attn_nll_fwd(x0.data_ptr(), x0.stride(0), x0.stride(1), None if x1 is None else x1.data_ptr(),
                                  0 if x1 is None else x1.stride(0), 0 if x1 is None else x1.stride(1), _dev(tg, torch.uint8), B, T, N,
                                  _dev(ws), _dev(loss), _stream()), "cvc_attn_nll_fwd")
    return loss, ws, tg


def attn_nll_bwd(x0, x1, tg, ws, g0, g1, want0: bool, want1: bool):
    B, T, N = x0.shape
    d0 = torch.empty(B, T, N, device=x0.device, dtype=torch.float32) if want0 else None
    d1 = torch.empty(B, T, N, device=x0.device, dtype=torch.float32) if (want1 and x1 is not None) else None
    _check(lib().cvc_attn_nll_bwd(x0.data_ptr(), x0.stride(0), x0.stride(1), None if x1 is None else x1.data_ptr(),
                                  0 if x1 is None else x1.stride(0), 0 if x1 is None else x1.stride(1), _dev(tg, torch.uint8), B, T, N,
                                  _dev(ws), _dev(g0), _dev(g1), _dev(d0), _dev(d1), _stream()), "cvc_attn_nll_bwd")
    return d0, d1


# --------------------------------------------------------------------------- tile path (rows > 64), csrc/gemm_tile.hip
def tile_rows_alloc(M: int) -> int:
    return int(lib().cvc_tile_rows_alloc(int(M)))


def _frag_ptr(xb: torch.Tensor, k0: int = 0):
    """(address of k step k0/16 of row block 0, row-block stride in bf16 elements) of a fragment tensor
    [blocks][k steps][3][2][32][8] int16."""
    if not xb.is_cuda or xb.dtype != torch.int16 or not xb.is_contiguous() or xb.dim() != 6:
        raise RuntimeError("cvc.hip: activation / weight fragments must be a contiguous int16 GPU tensor [blk][kstep][3][2][32][8]")
    assert k0 % 16 == 0
    return xb.data_ptr() + (k0 // 16) * 3 * 1024, xb.shape[1] * 1536


def tile_gemm(wb: torch.Tensor, xb: torch.Tensor, k0: int, K: int, M: int, N: int, ksplit: int, parts: Optional[torch.Tensor] = None):
    """parts [ksplit, M, N] = per-K-slice partial products of x[:, k0:k0+K] @ W^T (wb packed for exactly these K columns)."""
    wp, _ = _frag_ptr(wb)
    assert wb.shape[1] == K // 16 and wb.shape[0] * 32 >= N and xb.shape[0] * 32 >= tile_rows_alloc(M)
    xp, xs = _frag_ptr(xb, k0)
    if parts is None:
        parts = torch.empty(ksplit, M, N, device=xb.device, dtype=torch.float32)
    _check(lib().cvc_tile_gemm(wp, xp, xs, K, M, N, ksplit, _dev(parts), N, M * N, _stream()), "cvc_tile_gemm")
    return parts


def tile_pack_rows(x: torch.Tensor, xb: torch.Tensor, k0: int = 0, idx: Optional[torch.Tensor] = None, relu: bool = False):
    M = x.shape[0] if idx is None else idx.shape[0]
    K = x.shape[1]
    xp, xs = _frag_ptr(xb, k0)
    _check(lib().cvc_tile_pack_rows(_dev(x), x.stride(0), _dev(idx, torch.int64), 1 if relu else 0, M, K, xp, xs, _stream()),
           "cvc_tile_pack_rows")
    return xb


class _FragPool:
    """Fragment buffers of TileOperand, reused across calls.  A buffer is keyed by everything that decides which of its bytes the
    packers never write (rows, K -> the zero padding of the last row blocks / k step), so the padding is cleared ONCE per buffer
    instead of by a fill launch per operand (27 per training step); buffers return to the pool when their operand dies.  All use is
    stream-ordered on the launch stream.  Buffers taken while a HIP graph is being captured stay with that capture."""

    def __init__(self):
        self.free = {}

    def take(self, rows: int, K: int, device):
        rows_alloc = max(tile_rows_alloc(rows), (rows + 127) // 128 * 128)
        key = (rows, K, str(device), torch.cuda.is_current_stream_capturing())
        lst = self.free.get(key)
        if lst:
            return lst.pop(), key
        frags = torch.empty(rows_alloc // 32, (K + 15) // 16, 3, 2, 32, 8, dtype=torch.int16, device=device)
        # padding must read as zero: only what the packer does not write completely is cleared -- the row blocks past the last
        # full one, and the last k step when K is not a multiple of 16 (the row packer writes whole quads, not whole k steps)
        if rows_alloc > rows:
            frags[rows // 32:].zero_()
        if K % 16 != 0:
            frags[:, -1].zero_()
        return frags, key

    def give(self, frags, key):
        lst = self.free.setdefault(key, [])
        if len(lst) < 8:
            lst.append(frags)


_FRAGS = _FragPool()


class TileOperand:
    """One operand of tile_mm, packed once into bf16 split-term fragments (reusable as either side of several products)."""

    def __init__(self, t: torch.Tensor, kmajor: bool = False):
        """t is [rows, K], or [K, rows] with kmajor (read as its transpose).  Unit inner stride; any sizes."""
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError("cvc.hip: tile_mm operands must be fp32 GPU tensors (no CPU fallback)")
        if t.dim() != 2:
            raise RuntimeError("cvc.hip: tile_mm operands are matrices")
        if t.stride(1) != 1:
            t = t.contiguous()
        self.rows, self.K = (t.shape[1], t.shape[0]) if kmajor else (t.shape[0], t.shape[1])
        self.frags, self._key = _FRAGS.take(self.rows, self.K, t.device)
        self.ptr, self.stride = _frag_ptr(self.frags)
        L = lib()
        if kmajor:
            _check(L.cvc_tile_pack_cols(t.data_ptr(), t.stride(0), self.K, self.rows, self.ptr, self.stride, _stream()), "cvc_tile_pack_cols")
        else:
            _check(L.cvc_tile_pack_rows_any(t.data_ptr(), t.stride(0), self.rows, self.K, self.ptr, self.stride, _stream()),
                   "cvc_tile_pack_rows_any")

    def __del__(self):
        try:
            _FRAGS.give(self.frags, self._key)
        except Exception:
            pass


@functools.lru_cache(maxsize=None)
def tile_gemm_plan(M: int, N: int, K: int):
    """(ksplit, rows per workgroup, workgroups) of the tile GEMM for an [M, K] x [N, K]^T product (cvc_tile_gemm_plan)"""
    ks, rows, wgs = C.c_int(0), C.c_int(0), C.c_int(0)
    _check(lib().cvc_tile_gemm_plan(int(M), int(N), int(K), C.addressof(ks), C.addressof(rows), C.addressof(wgs)), "cvc_tile_gemm_plan")
    return ks.value, rows.value, wgs.value


def tile_mm(a, b, a_kmajor: bool = False, b_kmajor: bool = False, out: Optional[torch.Tensor] = None, parts_only: bool = False,
            bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[M, N] = sum_k A[m, k] B[n, k] on the tile GEMM (split products on the bf16 MFMA, fp32 accumulate, fp32-grade error):
    the dense products of the backward pass -- no library GEMM.  `a` is [M, K], or [K, M] with a_kmajor (read as its transpose);
    `b` is [N, K], or [K, N] with b_kmajor; either may be a TileOperand packed earlier (shared operands are packed once).
    out: [M, N] fp32 with unit inner stride and any row stride (a column block of a wider matrix: the weight-gradient products
    write their segment of weight_ih in place)."""
    A = a if isinstance(a, TileOperand) else TileOperand(a, a_kmajor)
    B = b if isinstance(b, TileOperand) else TileOperand(b, b_kmajor)
    assert A.K == B.K, (A.rows, A.K, B.rows, B.K)
    M, N, L, st = A.rows, B.rows, lib(), _stream()
    Kp = (A.K + 15) // 16 * 16
    ks = tile_gemm_plan(M, N, Kp)[0]               # chosen where the grid is chosen (csrc/gemm_tile.hip::cvc_tile_gemm_plan)
    if out is None:
        out = torch.empty(0 if parts_only else M, N, device=A.frags.device, dtype=torch.float32)
    elif not (out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (M, N) and out.stride(1) == 1
              and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0):
        raise RuntimeError("cvc.hip.tile_mm: out must be an fp32 GPU [M, N] view with unit inner stride, 16-byte aligned rows")
    ldy = out.stride(0)
    if parts_only:          # the K-slice slabs themselves, for a consumer that sums them (cvc_vocab_head_nll_fwd)
        parts = torch.empty(ks, M, N, device=A.frags.device, dtype=torch.float32)
        _check(L.cvc_tile_gemm(B.ptr, A.ptr, A.stride, Kp, M, N, ks, parts.data_ptr(), N, M * N, st), "cvc_tile_gemm")
        return parts
    if ks == 1 and bias is None:
        _check(L.cvc_tile_gemm(B.ptr, A.ptr, A.stride, Kp, M, N, 1, out.data_ptr(), ldy, M * ldy, st), "cvc_tile_gemm")
        return out
    # bias [N]: added by the finishing pass (nn.Linear's epilogue), not by a framework add over the result
    parts = torch.empty(ks, M, N, device=out.device, dtype=torch.float32)
    _check(L.cvc_tile_gemm(B.ptr, A.ptr, A.stride, Kp, M, N, ks, parts.data_ptr(), N, M * N, st), "cvc_tile_gemm")
    _check(L.cvc_tile_linear_finish(parts.data_ptr(), ks, M * N, N, _dev(bias), None, M, N, out.data_ptr(), ldy, st), "cvc_tile_linear_finish")
    return out
