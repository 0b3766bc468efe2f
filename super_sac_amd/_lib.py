"""ctypes binding of libssac_hip.so (the C ABI declared in include/ssac_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing the
import fails with instructions to build it; if a kernel launch fails the call raises
``RuntimeError(ssac_last_error())``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The LAB build (`./build.sh --lab`: phase stamps, per-workgroup timelines, SKIP_* experiment builds) is a SEPARATE file,
# libssac_hip_lab.so, that only the measurement scripts under tools/ ask for (SSAC_LAB_BUILD=1 in their environment): a
# forgotten rebuild can no longer ship the scaffolding as the product library.
LIB_PATH = os.path.join(_HERE, "libssac_hip.so")
if os.environ.get("SSAC_LAB_BUILD") == "1":   # (SSAC_LAB_TAG=name: an experiment variant, `./build.sh --lab --tag name -D...`)
    _tag = os.environ.get("SSAC_LAB_TAG")
    LIB_PATH = os.path.join(_HERE, f"libssac_hip_lab_{_tag}.so" if _tag else "libssac_hip_lab.so")


class MlpDesc(C.Structure):
    """struct ssac_mlp"""
    _fields_ = [("params", C.c_void_p), ("net_stride", C.c_int64), ("n_nets", C.c_int32),
                ("in_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32)]


class AdamCtl(C.Structure):
    """struct ssac_adam_ctl (device resident; this mirror is used to initialise / read it)"""
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step_size", C.c_float), ("bc2_sqrt", C.c_float),
                ("clip_coef", C.c_float), ("step", C.c_int32), ("_pad", C.c_int32 * 3),
                ("lr_d", C.c_double), ("beta1_d", C.c_double), ("beta2_d", C.c_double)]


class PopArtState(C.Structure):
    """struct ssac_popart"""
    _fields_ = [("mu", C.c_float), ("nu", C.c_float), ("w", C.c_float), ("b", C.c_float),
                ("t", C.c_int32), ("min_steps", C.c_int32), ("stable", C.c_int32), ("_pad", C.c_int32),
                ("beta", C.c_double)]


class Feed(C.Structure):
    """struct ssac_feed"""
    _fields_ = [("host_ring", C.c_void_p), ("dst", C.c_void_p), ("log_ring", C.c_void_p),
                ("tick", C.c_int64), ("n_slots", C.c_int32), ("slot_words", C.c_int32),
                ("log_slot_word", C.c_int32), ("log_width", C.c_int32), ("late_word", C.c_void_p)]


class Rng(C.Structure):
    """struct ssac_rng"""
    _fields_ = [("seed", C.c_uint64), ("counter", C.c_void_p), ("offset", C.c_int64)]


class TdSpec(C.Structure):
    """struct ssac_td_spec"""
    _fields_ = [("q_t", C.c_void_p), ("logp", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p),
                ("log_alpha", C.c_void_p), ("td_out", C.c_void_p), ("gamma", C.c_float),
                ("n_sel", C.c_int32), ("use_entropy", C.c_int32), ("n_parts", C.c_int32)]


class PushField(C.Structure):
    """struct ssac_push_field"""
    _fields_ = [("dst", C.c_void_p), ("row_bytes", C.c_int64), ("src_offset", C.c_int64)]


class LogFold(C.Structure):
    """struct ssac_logfold"""
    _fields_ = [("done_counter", C.c_void_p), ("logs", C.c_void_p), ("td_logs", C.c_void_p), ("feed", C.c_void_p),
                ("deferred_stats", C.c_void_p), ("late_word", C.c_void_p)]


class DeferredLogs(C.Structure):
    """struct ssac_deferred_logs"""
    _fields_ = [("partials", C.c_void_p), ("n_nets", C.c_int32), ("n_ss", C.c_int32), ("sumsq", C.c_void_p),
                ("td_stats", C.c_void_p), ("td_off", C.c_int32), ("n_rows", C.c_int32), ("denom", C.c_float),
                ("_pad", C.c_int32), ("feed", C.c_void_p)]


class ActorLogFold(C.Structure):
    """struct ssac_actor_logfold"""
    _fields_ = [("done_counter", C.c_void_p), ("partials", C.c_void_p), ("n_tiles", C.c_int32), ("n_rows", C.c_int32),
                ("inv_members", C.c_float), ("width", C.c_int32), ("logs_loss", C.c_void_p), ("logs_gn", C.c_void_p),
                ("block", C.c_void_p), ("ring", C.c_void_p), ("ring_slot", C.c_int64)]


class Gather(C.Structure):
    """struct ssac_gather"""
    _fields_ = [("s", C.c_void_p), ("s1", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p),
                ("s_elems", C.c_int64), ("a_elems", C.c_int64), ("idx", C.c_void_p), ("feed", C.c_void_p),
                ("xsa", C.c_void_p), ("ld_x", C.c_int64), ("x1sa", C.c_void_p), ("ld_x1", C.c_int64),
                ("rew_out", C.c_void_p), ("done_out", C.c_void_p), ("logs", C.c_void_p), ("n_logs", C.c_int32),
                ("rng_word", C.c_int32), ("ctl", C.c_void_p), ("ids_word", C.c_int32), ("_pad", C.c_int32)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_MP = C.POINTER(MlpDesc)

# name -> argtypes; every function returns int status unless listed in _RESTYPES
SIGNATURES = {
    "ssac_abi_version": [],
    "ssac_last_error": [],
    "ssac_mlp_layout": [_I, _I, _I, C.POINTER(C.c_int64)],
    "ssac_record_begin": [],
    "ssac_record_end": [],
    "ssac_launch_list_size": [_P],
    "ssac_polyak_multi": [_P, _P, _P, _I, _F, _P],
    "ssac_replay": [_P, _P],
    "ssac_replay_value": [_P, _P, C.c_longlong],
    "ssac_replay_value2": [_P, _P, C.c_longlong, C.c_longlong],
    "ssac_launch_list_free": [_P],
    "ssac_xchg_create": [C.c_int, C.c_int, C.c_int, C.c_int],
    "ssac_xchg_handle_bytes": [],
    "ssac_xchg_handle": [_P, _P],
    "ssac_xchg_connect": [_P, _P],
    "ssac_xchg_reduce": [_P, _P, _I, _I, _P],
    "ssac_xchg_reduce_owned": [_P, _P, _I, _P, _I, _I, _P],
    "ssac_bc_det_logprob_bwd": [_P, _L, _P, _L, _P, _I, _I, _F, _P, _L, _P, _P, _P],
    "ssac_action_invariance_det_bwd": [_P, _L, _P, _L, _I, _I, _F, _P, _L, _P, _P, _P],
    "ssac_per_assign": [_P, _P, _L, _P, _I, _P, _I, C.c_double, _P, _I, _L, _P, _P, _P],
    "ssac_per_sample": [_P, _P, _L, _L, _P, _I, C.c_double, _P, _P, _P],
    "ssac_xchg_error": [_P],
    "ssac_xchg_destroy": [_P],
    "ssac_step_create": [_P, _I, _I, _I, _I, _I, _I, _I, _I],
    "ssac_step_add_list": [_P, _P],
    "ssac_step_run": [_P, _P, _P, C.c_int32, _L, _P],
    "ssac_step_count": [_P],
    "ssac_step_seek": [_P, _L],
    "ssac_step_destroy": [_P],
    "ssac_replay_push": [_P, _I, _P, _I, _L, _L, _P],
    "ssac_gather_rows": [_P, _I, _L, _P, _I, _P, _L, _L, _P],
    "ssac_mlp_layer_fwd": [_MP, _I, _P, _I, _P, _L, _L, _I, _P, _L, _L, _I, _P],
    "ssac_mlp_layer_dgrad": [_MP, _I, _P, _I, _P, _L, _L, _P, _L, _L, _I, _P, _L, _L, _P],
    "ssac_wgrad_tiles": [_MP, _I],
    "ssac_mlp_layer_wgrad": [_MP, _I, _P, _I, _P, _L, _L, _P, _L, _L, _I, _P, _P, _P, _P, _P, _L, _P, _F, _P],
    "ssac_group_norms": [_P, _I, _I, _P, _P, _P],
    "ssac_mlp_wgrad_fc12": [_MP, _P, _I, _P, _L, _L, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P, _F, _P],
    "ssac_mlp_wgrad_all": [_MP, _P, _I, _P, _L, _L, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _F, _P],
    "ssac_mlp_wgrad_all_actor": [_MP, _P, _L, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "ssac_mlp_wgrad_all_scaled": [_MP, _P, _I, _P, _L, _L, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _F,
                                  _P],
    "ssac_critic_loss_bwd_lazy": [_P, _I, _I, _I, _P, _L, _P, _P, _P, _I, _F, _P, _P, _P],
    "ssac_mlp_wgrad_all_lossfold": [_MP, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _I, _P, _P, _P, _P,
                                    _P, _P, _P, _L, _P, _F, _P, _P],
    "ssac_target_fwd_critic_bwdu": [_MP, _P, _I, _P, _L, _I, _P, _MP, _P, _P, _P, _L, _P, _P, _P],
    "ssac_gather_transition": [_P, _P, _I, _L, _P, _L, _P, _P, _P, _I, _P, _L, _P, _L, _P, _P, _P],
    "ssac_gather_transition_begin": [_P, _P, _I, _L, _P, _L, _P, _P, _I, _P, _L, _P, _L, _P, _P, _P, _P, _I, _P, _P],
    "ssac_adam_step": [_P, _P, _P, _P, _L, _P, _P],
    "ssac_adam_advance": [_P, _P],
    "ssac_begin_update": [_P, _I, _P, _P, _P],
    "ssac_publish_logs": [_P, _P, _P],
    "ssac_clip_coef": [_P, _P, _I, _F, _P, _P],
    "ssac_polyak": [_P, _P, _L, _F, _P],
    "ssac_tanh_normal_fwd": [_P, _L, _P, _I, _I, _F, _F, _P, _L, _L, _P, _P],
    "ssac_det_action_fwd": [_P, _L, _P, _F, _P, _F, _F, _I, _I, _P, _L, _L, _P],
    "ssac_td_target": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _I, _P, _P, _P],
    "ssac_critic_loss_bwd": [_P, _I, _I, _I, _P, _L, _P, _P, _P, _I, _F, _P, _P, _P],
    "ssac_dr3_blocks": [],
    "ssac_dr3_add": [_P, _P, _I, _I, _I, _F, _P, _P],
    "ssac_adv_filter_discrete": [_P, _I, _I, _I, _P, _I, _P, _L, _P, _P, _P, _P, _P, _P],
    "ssac_bc_discrete_bwd": [_P, _P, _L, _P, _I, _I, _F, _P, _P, _P, _P],
    "ssac_adv_filter": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P],
    "ssac_bc_logprob_bwd": [_P, _L, _P, _L, _P, _I, _I, _F, _F, _F, _P, _L, _P, _P, _P],
    "ssac_bce_sigmoid_bwd": [_P, _I, _I, _F, _P, _P, _P],
    "ssac_markov_smoothness_bwd": [_P, _L, _P, _L, _I, _I, _F, _F, _P, _L, _P, _L, _I, _P, _P],
    "ssac_frobenius_diff_bwd": [_P, _L, _P, _L, _I, _I, _F, _P, _L, _I, _P, _P, _P],
    "ssac_action_invariance_bwd": [_P, _L, _P, _L, _P, _I, _I, _F, _F, _F, _P, _L, _P, _P, _P],
    "ssac_action_invariance_discrete_bwd": [_P, _P, _P, _I, _I, _F, _P, _P, _P, _P],
    "ssac_exploration_noise": [_P, _L, _L, _P, _F, _F, _I, _I, _P],
    "ssac_det_logprob": [_P, _I, _I, _P, _P],
    "ssac_markov_logs": [_P, _F, _P, _P, _F, _F, _F, _P, _P],
    "ssac_actor_loss_bwd": [_P, _I, _I, _P, _P, _I, _P, _I, _F, _P, _P, _P, _P],
    "ssac_actor_loss_bwd_adv": [_P, _I, _I, _P, _P, _I, _P, _I, _F, _P, _P, _P, _P],
    "ssac_tanh_normal_bwd": [_P, _I, _L, _L, _L, _P, _L, _P, _I, _I, _F, _F, _P, _I, _F, _P, _L, _P],
    "ssac_det_action_bwd": [_P, _I, _L, _L, _L, _P, _L, _I, _I, _P, _L, _P],
    "ssac_discrete_actor_loss_bwd": [_P, _P, _I, _I, _I, _P, _P, _I, _F, _P, _P, _P],
    "ssac_alpha_update": [_P, _P, _P, _P, _P, _I, _I, _F, _P, _P],
    "ssac_sunrise_weights": [_P, _I, _I, _F, _P, _P, _P],
    "ssac_softmax_weights": [_P, _I, _I, _F, _P, _P, _P],
    "ssac_ensemble_min_select": [_P, _I, _I, _I, _P, _L, _P, _P],
    "ssac_drq_shift": [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P],
    "ssac_zero": [_P, _L, _P],
    "ssac_im2col": [_P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P],
    "ssac_col2im": [_P, _P, _L, _L, _L, _L, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P],
    "ssac_col2im_cl": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ssac_linear_fwd": [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P],
    "ssac_conv_implicit_supported": [_I, _I, _I],
    "ssac_conv_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ssac_conv_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ssac_conv_wgrad_slices": [_I, _I, _I, _I],
    "ssac_conv_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ssac_conv_first_supported": [_I, _I, _I, _I, _I, _I, _L],
    "ssac_conv_first_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P],
    "ssac_conv_first_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P],
    "ssac_conv_wgrad_img_slices": [_I, _I, _I, _I, _I, _I, _I],
    "ssac_conv_wgrad_img": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ssac_conv_first_wgrad_band_slices": [_I, _I, _I, _I, _I, _I, _I],
    "ssac_conv_first_wgrad_band": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P],
    "ssac_linear_fwd_splitk": [_P, _L, _P, _L, _P, _I, _I, _I, _I, _P],
    "ssac_linear_fwd_stream_supported": [_I, _I, _I, _I, _L, _L],
    "ssac_linear_fwd_stream": [_P, _L, _P, _L, _P, _I, _I, _I, _I, _P],
    "ssac_reduce_slices_bias": [_P, _I, _I, _I, _P, _P, _L, _P],
    "ssac_linear_dgrad": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P],
    "ssac_linear_dgrad_masked": [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _P],
    "ssac_linear_wgrad_splitk": [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _P],
    "ssac_reduce_slices": [_P, _I, _L, _P, _P],
    "ssac_reduce_slices_pair": [_P, _L, _P, _P, _L, _P, _I, _P],
    "ssac_relu_mask": [_P, _P, _L, _P],
    "ssac_relu_mask_to": [_P, _P, _L, _P, _P],
    "ssac_permute_cp": [_P, _P, _I, _I, _I, _I, _P],
    "ssac_sumsq_blocks": [],
    "ssac_sumsq": [_P, _L, _P, _P],
    "ssac_ln_tanh_fwd": [_P, _L, _P, _P, _I, _I, _P, _L, _P, _P, _P],
    "ssac_ln_tanh_bwd": [_P, _L, _P, _L, _P, _P, _P, _I, _I, _P, _L, _P, _P, _P, _P],
    "ssac_fused_supported": [_MP],
    "ssac_fused_row_tiles": [_MP, _I, _I],
    "ssac_step_polyak": [_P, _F],
    "ssac_step_polyak_done": [_P],
    "ssac_feed_ring_alloc": [C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int)],
    "ssac_feed_ring_free": [_P, _I],
    "ssac_feed_ring_mode": [_I],
    "ssac_feed_write": [_P, _P, C.c_size_t],
    "ssac_mlp3_fwd_fused": [_MP, _P, _I, _P, _L, _L, _I, _P, _P, _P, _P],
    "ssac_actor_sample_fused": [_MP, _P, _L, _I, _P, _F, _F, _P, _L, _L, _P, _P, _P, _P, _P, _P],
    "ssac_actor_sample_critic_fwd": [_MP, _P, _L, _I, _P, _F, _F, _P, _L, _L, _P, _P, _MP, _P, _L, _P, _P, _P, _P, _P],
    "ssac_chain_update": [_MP, _P, _L, _I, _P, _F, _F, _P, _L, _L, _P, _P, _MP, _P, _I, _P, _MP, _P, _L, _P, _P, _P, _P, _P,
                          _P, _P, _P, _P, _I, _P, _P],
    "ssac_chain_target_splits": [_MP, _MP, _MP, _I, _I],
    "ssac_deferred_logs_flush": [_P, _I, _P],
    "ssac_philox_normal": [_P, _I, _I, _P, _P],
    "ssac_actor_sample_concat_fused": [_MP, _P, _L, _I, _P, _F, _F, _P, _L, _P, _P, _P, _P, _P, _P],
    "ssac_critic_fwd_dx_fused": [_MP, _P, _L, _I, _I, _I, _P, _P, _P],
    "ssac_actor_bwd_fused": [_MP, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _F, _F, _F, _P, _I, _P, _P, _P, _P, _P],
    "ssac_actor_logs": [_P, _I, _I, _F, _P, _I, _P, _P, _P, _I, _P, C.c_longlong, _P],
    "ssac_actor_route_local": [_P, _P, _I, _I, _I, _P, _P, _P, _P],
    "ssac_actor_route_claim": [_P, _P, _I, _I, _P, _P],
    "ssac_actor_route_mask": [_P, _I, _I, _I, _P, _P],
    "ssac_actor_chain_fused": [_MP, _P, _L, _I, _P, _P, _F, _F, _P, _L, _P, _P, _P, _P, _MP, _P, _P, _P, _I, _F, _P, _I,
                               _P, _P, _P, _P, _P, _L, _P, _I, _P, _P],
    "ssac_actor_chain_handoff_words": [_I, _I, _I],
    "ssac_critic_fwd_bwd_fused": [_MP, _P, _L, _I, _P, _P, _P, _L, _P, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ssac_critic_bwd_fused": [_MP, _I, _P, _P, _P, _L, _P, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ssac_head_wgrad_tiles": [_MP],
    "ssac_head_wgrad": [_MP, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _L, _P, _F, _P],
    "ssac_critic_logs": [_P, _I, _I, _I, _F, _P, _I, _P, _P, _P, _P, _P, _P],
    "ssac_act_create": [_I, _I],
    "ssac_act_obs": [_P],
    "ssac_act_counter": [_P],
    "ssac_act_publish": [_P, _P, _I, _P],
    "ssac_act_add_list": [_P, _P],
    "ssac_act_run": [_P, _I, _P, _I, _P, _I, _P],
    "ssac_act_calls": [_P],
    "ssac_act_destroy": [_P],
    "ssac_ucb_select": [_P, _I, _I, _I, _I, _F, _P, _L, _I, _I, _P, _P],
    "ssac_act_candidates": [_P, _I, _I, _I, _P, _L, _I, _F, _F, _P, C.c_longlong, _P, _L, _P],
    "ssac_act_mean_tanh": [_P, _I, _L, _I, _I, _P, _P],
    "ssac_act_take_clamp": [_P, _L, _I, _I, _I, _F, _F, _P, _P],
    "ssac_act_discrete": [_P, _I, _L, _I, _I, _I, _P, _P, _P],
    "ssac_bf16_layout": [_I, _I, _I, C.POINTER(C.c_int64)],
    "ssac_bf16_supported": [_MP],
    "ssac_bf16_sync": [_MP, _P, _P],
    "ssac_bf16_polyak": [_MP, _MP, _F, _P, _P],
    "ssac_bf16_mlp3_fwd": [_MP, _P, _P, _I, _P, _L, _I, _P, _P],
    "ssac_bf16_chain_update": [_MP, _P, _P, _L, _I, _P, _F, _F, _P, _L, _L, _P, _P, _MP, _P, _P, _I, _P, _MP, _P, _P, _L,
                               _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ssac_bf16_wgrad_tiles": [_MP],
    "ssac_bf16_wgrad_lossfold": [_MP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _I, _P, _P, _P, _P, _P, _L,
                                 _P, _P, _F, _P, _P],
}
# include/ssac_hip_test.h, group 1 (form selection: always exported, used by tests/ and tools/ only)
TEST_SIGNATURES = {
    "ssac_slot_by_value": [_I],
    "ssac_gemm_lean": [_I],
    "ssac_wgrad_variant": [_I],
    "ssac_fused_tile_rows": [_I],
    "ssac_xcd_order": [_I],
    "ssac_chain_form": [_I],
    "ssac_bf16_fwd_form": [_I],
}
# include/ssac_hip_test.h, group 2 (lab hooks): defined by the LAB build only (`./build.sh --lab`, SSAC_LAB_BUILD=1); bound when present
LAB_SIGNATURES = {
    "ssac_fused_debug_stamps": [_P],
    "ssac_gemm_debug_stamps": [_P],
    "ssac_bf16_debug_stamps": [_P],
    "ssac_debug_timeline": [_P],
    "ssac_xchg_test_mode": [_P, _I],
}
_RESTYPES = {"ssac_act_create": C.c_void_p, "ssac_act_obs": C.c_void_p, "ssac_act_counter": C.c_void_p, "ssac_act_calls": C.c_longlong,
             "ssac_act_destroy": None, "ssac_xchg_create": C.c_void_p, "ssac_xchg_destroy": None, "ssac_step_create": C.c_void_p, "ssac_step_count": C.c_int64, "ssac_actor_chain_handoff_words": C.c_int64, "ssac_step_destroy": None,
             "ssac_last_error": C.c_char_p, "ssac_mlp_layout": C.c_int64, "ssac_bf16_layout": C.c_int64, "ssac_record_end": C.c_void_p,
             "ssac_launch_list_free": None}


# SSAC_ABI_VERSION of include/ssac_hip.h this binding table was written against (bumped with every signature change:
# a stale .so called with shifted pointer arguments would corrupt device memory)
ABI_VERSION = 7


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is the only implementation of this package. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in {**SIGNATURES, **TEST_SIGNATURES}.items():
        fn = getattr(lib, name)  # AttributeError here == ABI mismatch, fail loudly
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    for name, argtypes in LAB_SIGNATURES.items():
        fn = getattr(lib, name, None)   # (absent from the product library by design)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = C.c_int
    if lib.ssac_abi_version() != ABI_VERSION:
        raise ImportError(f"libssac_hip.so has ABI version {lib.ssac_abi_version()}, this package binds version "
                          f"{ABI_VERSION}: rebuild the extension (build.sh)")
    return lib


lib = _load()


def check(status):
    if status != 0:
        raise RuntimeError("libssac_hip: " + lib.ssac_last_error().decode())
