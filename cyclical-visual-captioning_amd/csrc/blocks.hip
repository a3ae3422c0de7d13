// cvc_block(name): the address of a building block (include/cvc_hip_blocks.h) -- and, in a build made with CVC_EXPERIMENTAL=1, of
// an experimental form (include/cvc_hip_experimental.h) -- by name.  The library exports only the drop-in ABI of include/cvc_hip.h
// (hidden visibility for everything else); unit tests and the host mirror's eager launch lists (cvc/decode.py) bind the blocks
// through this table.
#include <string.h>
#include "cvc_common.h"

namespace {
struct Entry { const char* name; void* fn; };
#define CVC_B(f) {#f, (void*)&f}
const Entry table[] = {
    CVC_B(cvc_attn_scores),
    CVC_B(cvc_attn_wsum),
    CVC_B(cvc_attn_scores_qparts),
    CVC_B(cvc_attn_wsum_quad),
    CVC_B(cvc_attn_wsum_frag),
    CVC_B(cvc_attn_wsum_quad_rm),
    CVC_B(cvc_attn_bwd_pair),
    CVC_B(cvc_ctxfeat_bwd_steps),
    CVC_B(cvc_tile_gemm_big),
    CVC_B(cvc_tile_gemm_plan),
    CVC_B(cvc_dproj_bwd_steps),
    CVC_B(cvc_linear_splitk_fwd),
    CVC_B(cvc_linear_top2_fwd),
    CVC_B(cvc_top2_final),
    CVC_B(cvc_packed_lstm_fwd),
    CVC_B(cvc_packed_linear_fwd),
    CVC_B(cvc_packed_lstm_embgate_fwd),
    CVC_B(cvc_packed_lstm_embgate_ex_fwd),
    CVC_B(cvc_packed_lstm_late_fwd),
    CVC_B(cvc_packed_lstm_train_fwd),
    CVC_B(cvc_packed_lstm_train_pre_fwd),
    CVC_B(cvc_packed_lstm_train_drop_fwd),
    CVC_B(cvc_lstm_pointwise_bwd),
    CVC_B(cvc_lstm_pointwise_bwd3),
    CVC_B(cvc_lstm_pointwise_bwd3_drop),
    CVC_B(cvc_pack_lstm_weights),
    CVC_B(cvc_linear_nn_planes_fwd),
    CVC_B(cvc_linear_nn_planes2_fwd),
    CVC_B(cvc_gru_seq_train_fwd),
    CVC_B(cvc_lstm_pointwise_bwd4_pair),
    CVC_B(cvc_beam_select_parts),
    CVC_B(cvc_tile_lstm_finish),
    CVC_B(cvc_tile_lstm_finish_embgate),
    CVC_B(cvc_tile_reorder_pack),
    CVC_B(cvc_decode_num_launches),
    CVC_B(cvc_gemm_force_generic),
    CVC_B(cvc_tile_gemm_loaders),
    CVC_B(cvc_gru_persistent_waves8),
    CVC_B(cvc_relu_dropout_fwd),
    CVC_B(cvc_relu_dropout_bwd),
    CVC_B(cvc_bn_workspace),
    CVC_B(cvc_bn_relu_train_fwd),
    CVC_B(cvc_bn_relu_train_bwd),
    CVC_B(cvc_class_softmax_bwd),
    CVC_B(cvc_layernorm_cat_bwd),
    CVC_B(cvc_attn_weighted_rows),
    CVC_B(cvc_stable_order),
    CVC_B(cvc_col_sum),
    CVC_B(cvc_col_sum_ws),
#ifdef CVC_EXPERIMENTAL
    CVC_B(cvc_gsk_plan),
    CVC_B(cvc_gsk_gemm),
    CVC_B(cvc_attn_scores_qslab),
    CVC_B(cvc_top2_slab),
    CVC_B(cvc_packed_lstm_ks_slices),
    CVC_B(cvc_packed_lstm_ks_fwd),
    CVC_B(cvc_packed_lstm_ksf_fwd),
    CVC_B(cvc_packed_lstm_ksx_local),
    CVC_B(cvc_packed_lstm_ksx_fwd),
    CVC_B(cvc_packed_lstm_wg_blocks),
    CVC_B(cvc_packed_linear_select_fwd),
    CVC_B(cvc_gru_persistent_halves),
#endif
};
}  // namespace

extern "C" void* cvc_block(const char* name) {
    if (name == nullptr) return nullptr;
    for (const Entry& e : table)
        if (strcmp(e.name, name) == 0) return e.fn;
    return nullptr;
}
