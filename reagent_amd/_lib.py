"""ctypes binding of libreagent_hip.so (the C ABI in include/reagent_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C reagent_amd/csrc``.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RG_LIB: another build of the same library (same-box A/B of compile-time kernel variants); default = the in-tree build
LIB_PATH = os.environ.get("RG_LIB") or os.path.join(_HERE, "lib", "libreagent_hip.so")

ABI_VERSION = 11  # rg_abi_version() of include/reagent_hip.h this module's structs and signatures mirror
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
DT_F32, DT_BF16 = 0, 1
ACT = {"linear": 0, "relu": 1, "leaky_relu": 2, "tanh": 3, "sigmoid": 4, "softplus": 5}
LOSS = {"mse": 0, "huber": 1}
MAX_GATHER_COLS = 16

c_void_p, c_int, c_i64, c_f, c_d, c_sz = (
    ctypes.c_void_p,
    ctypes.c_int,
    ctypes.c_int64,
    ctypes.c_float,
    ctypes.c_double,
    ctypes.c_size_t,
)


class GatherCol(ctypes.Structure):
    _fields_ = [
        ("src", c_void_p),
        ("dst", c_void_p),
        ("indices", c_void_p),
        ("row_elems", ctypes.c_int32),
        ("elem_bytes", ctypes.c_int32),
        ("norm", c_void_p),
        ("norm_quantiles", c_void_p),
        ("out_dtype", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class NormCol(ctypes.Structure):
    _fields_ = [
        ("op", ctypes.c_int32),
        ("in_col", ctypes.c_int32),
        ("p0", c_f),
        ("p1", c_f),
        ("p2", c_f),
        ("p3", c_f),
    ]


MLP_MAX_LAYERS = 6


class MlpDesc(ctypes.Structure):
    _fields_ = [
        ("n_layers", ctypes.c_int32),
        ("dims", ctypes.c_int32 * (MLP_MAX_LAYERS + 1)),
        ("acts", ctypes.c_int32 * MLP_MAX_LAYERS),
        ("wfrag_fwd", c_void_p * MLP_MAX_LAYERS),
        ("wfrag_bwd", c_void_p * MLP_MAX_LAYERS),
        ("bias", c_void_p * MLP_MAX_LAYERS),
        ("act_frag", c_void_p * MLP_MAX_LAYERS),
        ("dz_frag", c_void_p * MLP_MAX_LAYERS),
        ("act_sign", c_void_p * MLP_MAX_LAYERS),
        ("db", c_void_p * MLP_MAX_LAYERS),
        ("w", c_void_p * MLP_MAX_LAYERS),
        ("dw", c_void_p * MLP_MAX_LAYERS),
        ("x3", ctypes.c_int32),
        ("dx_only", ctypes.c_int32),
        ("x2", c_void_p),
        ("ldx2", ctypes.c_int64),
        ("x_split", ctypes.c_int32),
        ("dx_col0", ctypes.c_int32),
        ("x2_dtype", ctypes.c_int32),
        ("wgrad_flags", ctypes.c_int32),  # ABI 10: bit 0 = the weight-gradient launch shares the chip (even splits)
        ("rowmap", c_void_p),
        ("tile_key", c_void_p),
        ("row_begin", c_void_p),
        ("n_groups", ctypes.c_int32),
        ("out_scatter", ctypes.c_int32),
        ("group_stride_fwd", ctypes.c_int64),
        ("group_stride_bwd", ctypes.c_int64),
        ("defer_db", ctypes.c_int32),
        ("sum_n", ctypes.c_int32),
        ("db_partials", c_void_p),
        ("sum_in", c_void_p),
        ("sum_out", c_void_p),
        ("sum_scale", ctypes.c_double),
    ]


class MlpUpdateDesc(ctypes.Structure):
    _fields_ = [
        ("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("target", c_void_p),
        ("n_layers", ctypes.c_int32),
        ("dims", ctypes.c_int32 * (MLP_MAX_LAYERS + 1)),
        ("w_off", ctypes.c_int64 * MLP_MAX_LAYERS),
        ("b_off", ctypes.c_int64 * MLP_MAX_LAYERS),
        ("wfrag_fwd", c_void_p * MLP_MAX_LAYERS),
        ("wfrag_bwd", c_void_p * MLP_MAX_LAYERS),
        ("target_wfrag_fwd", c_void_p * MLP_MAX_LAYERS),
        ("x3", ctypes.c_int32),
        ("group_rows", ctypes.c_int32 * MLP_MAX_LAYERS),
        ("sched_pre_ticked", ctypes.c_int32),
        ("post_tick_mod", ctypes.c_int32),
        ("post_tick", c_void_p),
    ]


TABLE_COLUMNS = ("state_features", "state_features_presence", "next_state_features", "next_state_features_presence",
                 "action", "next_action", "reward", "action_probability", "time_diff", "step", "mdp_id",
                 "sequence_number", "possible_actions_mask", "possible_next_actions_mask")


class DqnTable(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in TABLE_COLUMNS] + [
        ("n_rows", ctypes.c_int64), ("n_features", ctypes.c_int32), ("n_actions", ctypes.c_int32)]


BATCH_OUT_FIELDS = ("state", "next_state", "action", "next_action", "reward", "time_diff", "step", "not_terminal",
                    "possible_actions_mask", "possible_next_actions_mask", "action_probability", "mdp_id",
                    "sequence_number")


class DqnBatchOut(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in BATCH_OUT_FIELDS] + [
        ("state_dtype", ctypes.c_int32), ("reserved", ctypes.c_int32)]


REPLAY_VIEW_COLUMNS = ("observation", "action", "reward", "terminal", "log_prob", "possible_actions_mask", "mdp_id",
                       "sequence_number", "decays")


class ReplayView(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in REPLAY_VIEW_COLUMNS] + [
        ("capacity", ctypes.c_int64), ("n_features", ctypes.c_int32), ("n_actions", ctypes.c_int32),
        ("update_horizon", ctypes.c_int32), ("reserved", ctypes.c_int32)]


POLICY_VIEW_COLUMNS = ("observation", "action", "reward", "terminal", "log_prob", "decays", "ranges")


class PolicyReplayView(ctypes.Structure):  # rg_policy_replay_view (ABI 11)
    _fields_ = [(n, c_void_p) for n in POLICY_VIEW_COLUMNS] + [
        ("capacity", ctypes.c_int64), ("n_features", ctypes.c_int32), ("action_dim", ctypes.c_int32),
        ("update_horizon", ctypes.c_int32), ("reserved", ctypes.c_int32)]


POLICY_OUT_FIELDS = ("state", "next_state", "action", "next_action", "reward", "not_terminal", "action_probability")


class PolicyBatchOut(ctypes.Structure):  # rg_policy_batch_out (ABI 11)
    _fields_ = [(n, c_void_p) for n in POLICY_OUT_FIELDS] + [("state_dtype", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# every symbol include/reagent_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rg_strerror": (ctypes.c_char_p, [c_int]),
    "rg_abi_version": (c_int, []),
    "rg_fc_forward": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64,
                               c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rg_fc_dgrad": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p,
                             c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "rg_fc_wgrad_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "rg_fc_wgrad": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_sz,
                             c_int, c_int, c_int, c_int, c_void_p]),
    "rg_act_backward": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "rg_td3_target_action": (c_int, [c_void_p, c_i64, c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "rg_transpose_cast": (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_i64, c_void_p,
                                   c_i64, c_int, c_void_p]),
    "rg_mlp_fused_supported": (c_int, [ctypes.POINTER(MlpDesc)]),
    "rg_frag_elems": (c_sz, [c_int, c_int]),
    "rg_sign_bytes": (c_sz, [c_int, c_int]),
    "rg_wfrag_elems": (c_sz, [c_int, c_int]),
    "rg_stage_weights_frag": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rg_mlp_forward_fused": (c_int, [ctypes.POINTER(MlpDesc), c_void_p, c_int, c_i64, c_int, c_void_p, c_i64,
                                      c_int, c_void_p]),
    "rg_mlp_backward_fused_workspace_bytes": (c_sz, [ctypes.POINTER(MlpDesc), c_int]),
    "rg_mlp_backward_fused": (c_int, [ctypes.POINTER(MlpDesc), c_void_p, c_i64, c_int, c_void_p, c_i64,
                                       c_void_p, c_sz, c_void_p]),
    "rg_fc_wgrad_frag_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "rg_fc_wgrad_frag": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_sz, c_void_p]),
    "rg_mlp_stage_weights_fused": (c_int, [ctypes.POINTER(MlpDesc), c_int, c_void_p]),
    "rg_mlp_wgrad_fused_workspace_bytes": (c_sz, [ctypes.POINTER(MlpDesc), c_int]),
    "rg_mlp_wgrad_fused": (c_int, [ctypes.POINTER(MlpDesc), c_int, c_void_p, c_sz, c_void_p]),
    "rg_replay_nstep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rg_replay_gather": (c_int, [ctypes.POINTER(GatherCol), c_int, c_i64, c_int, c_int, c_void_p]),
    "rg_mlp_update_fused": (c_int, [ctypes.POINTER(MlpUpdateDesc)] + [ctypes.c_double] * 9 + [c_void_p]),
    "rg_sumtree_depth": (c_int, [c_i64]),
    "rg_sumtree_nodes": (c_sz, [c_i64]),
    "rg_sumtree_set": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rg_sumtree_sample": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "rg_sumtree_get": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "rg_make_dqn_input": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "rg_layer_norm_forward": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_d, c_int, c_int, c_int, c_void_p, c_int, c_i64,
                                       c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "rg_layer_norm_backward_workspace_bytes": (c_sz, [c_int, c_int]),
    "rg_layer_norm_backward": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_void_p, c_int, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_sz, c_void_p]),
    "rg_batch_norm_workspace_bytes": (c_sz, [c_int, c_int]),
    "rg_batch_norm_forward": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_d, c_d, c_int, c_int,
                                       c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_sz, c_void_p]),
    "rg_batch_norm_backward": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_d,
                                        c_int, c_int, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_sz, c_void_p]),
    "rg_dropout": (c_int, [c_void_p, c_i64, c_int, c_int, c_d, c_int, ctypes.c_uint64, ctypes.c_uint64, c_void_p, c_void_p,
                            c_i64, c_void_p]),
    "rg_ragged_offsets": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "rg_ragged_copy": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "rg_make_policy_input": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rg_normalize_dense": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_void_p,
                                    c_void_p, c_i64, c_int, c_void_p]),
    "rg_table_dqn_batch": (c_int, [ctypes.POINTER(DqnTable), c_void_p, c_int, c_void_p, c_int, c_void_p,
                                    ctypes.POINTER(DqnBatchOut), c_void_p]),
    "rg_replay_dqn_batch": (c_int, [ctypes.POINTER(ReplayView), c_void_p, c_int, c_void_p, c_void_p,
                                     ctypes.POINTER(DqnBatchOut), c_void_p]),
    "rg_replay_policy_batch": (c_int, [ctypes.POINTER(PolicyReplayView), c_void_p, c_int, c_void_p, c_void_p,
                                        ctypes.POINTER(PolicyBatchOut), c_void_p]),
    "rg_replay_dqn_batch_pooled": (c_int, [ctypes.POINTER(ReplayView), c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                            ctypes.POINTER(DqnBatchOut), c_void_p]),
    "rg_table_check_actions": (c_int, [ctypes.POINTER(DqnTable), c_void_p, c_int, c_void_p, c_void_p]),
    "rg_bcq_filter": (c_int, [c_void_p, c_int, c_int, c_d, c_void_p, c_void_p]),
    "rg_dqn_head_partials": (c_int, [c_int]),
    "rg_dqn_head": (c_int, [c_void_p] * 8 + [c_d, c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "rg_cpe_head": (c_int, [c_void_p] * 9 + [ctypes.c_double, c_void_p, ctypes.c_double, c_int, c_int, c_int, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rg_c51_head": (c_int, [c_void_p] * 8 + [c_d, c_void_p, c_void_p, c_d, c_d, c_int, c_int, c_int, c_int, c_void_p,
                            c_void_p, c_void_p, c_void_p]),
    "rg_crr_partials": (c_int, [c_int]),
    "rg_crr_critic_head": (c_int, [c_void_p] * 9 + [c_d, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "rg_crr_actor_head": (c_int, [c_void_p] * 4 + [c_d, c_d, c_d, c_d, c_int, c_int] + [c_void_p] * 3 + [c_void_p]),
    "rg_qr_head": (c_int, [c_void_p] * 8 + [c_d, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                            c_void_p, c_void_p, c_void_p]),
    "rg_gaussian_head_forward": (c_int, [c_void_p, c_i64, c_void_p, c_int, c_int, c_void_p, c_i64, c_void_p,
                                          c_void_p, c_void_p]),
    "rg_gaussian_log_prob": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p]),
    "rg_gaussian_head_backward": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int,
                                           c_void_p, c_i64, c_void_p]),
    "rg_gaussian_head_backward_kld": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int,
                                               c_void_p, c_i64, c_void_p, c_int, c_void_p]),
    "rg_sac_kld": (c_int, [c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_d] + [c_void_p] * 4 + [c_void_p]),
    "rg_sac_partials": (c_int, [c_int]),
    "rg_sac_critic_head": (c_int, [c_void_p] * 7 + [c_d, c_void_p, c_int] + [c_void_p] * 5 + [c_void_p]),
    "rg_sac_actor_head": (c_int, [c_void_p] * 4 + [c_d, c_int, c_void_p, c_int, c_d, c_d, c_int] + [c_void_p] * 5 + [c_void_p]),
    "rg_sac_alpha_grad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rg_adam_step_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_d, c_d, c_d, c_d, c_d, c_d,
                                  c_void_p, c_void_p]),
    "rg_add_cols": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "rg_reduce_sum": (c_int, [c_void_p, c_int, c_f, c_void_p, c_void_p]),
    "rg_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_d, c_d, c_d, c_d, c_d,
                              c_d, c_d, c_d, c_void_p]),
    "rg_soft_update": (c_int, [c_void_p, c_void_p, c_i64, c_d, c_void_p]),
    "rg_dueling_combine": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "rg_dueling_split": (c_int, [c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p]),
    "rg_group_rows_workspace_bytes": (c_sz, [c_int, c_int]),
    "rg_group_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_sz, c_void_p]),
    "rg_group_wfrag_elems": (c_sz, [c_int, c_int, c_int]),
    "rg_group_weights_stage": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rg_wide_head_mean": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rg_wide_head_mean_staged": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rg_qr_select_action": (c_int, [c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rg_qr_select_group_rows": (c_int, [c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_sz, c_void_p]),
    "rg_qr_compact_head": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p, c_d, c_void_p, c_void_p, c_int, c_int, c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "rg_group_head_wgrad_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "rg_group_head_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_sz, c_void_p]),
    "rg_adam_step_sched": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_d, c_d, c_d, c_d, c_d, c_void_p,
                                    c_void_p]),
    "rg_mlp_update_fused_sched": (c_int, [ctypes.POINTER(MlpUpdateDesc)] + [ctypes.c_double] * 6 + [c_void_p, c_void_p]),
    "rg_adam_step_f64_sched": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_d, c_d, c_d, c_void_p,
                                        c_void_p, c_void_p]),
    "rg_sched_tick": (c_int, [c_void_p, c_void_p]),
    "rg_sched_tick_many": (c_int, [ctypes.POINTER(c_void_p), c_int, c_void_p]),
}

_lib = None
_lock = threading.Lock()


EUNSUPPORTED = -3  # RG_EUNSUPPORTED


class ReagentHipError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    """dlopen the library and attach prototypes.  Raises if it is missing or lacks a symbol."""
    if not os.path.exists(path):
        raise ReagentHipError(
            f"{path} not found: the HIP extension is not built and there is no CPU fallback. "
            "Run `make -C reagent_amd/csrc` (or __graft_entry__.build())."
        )
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ReagentHipError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.rg_abi_version() != ABI_VERSION:  # struct layouts / argument meanings moved: a stale build must not run
        raise ReagentHipError(f"{path} has ABI {lib.rg_abi_version()}, this package binds ABI {ABI_VERSION}: rebuild "
                              "(`make -C reagent_amd/csrc`)")
    return lib


def lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = load()
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().rg_strerror(rc).decode()
        raise ReagentHipError(f"{what or 'reagent_hip call'} failed: {msg} (code {rc})")


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream (kernels are enqueued there)."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_cuda(t: torch.Tensor, name: str = "tensor"):
    if not t.is_cuda:
        raise ReagentHipError(
            f"{name} lives on {t.device}: reagent_amd ops run only on an AMD GPU (no CPU fallback)"
        )
