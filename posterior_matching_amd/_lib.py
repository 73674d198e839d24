"""ctypes binding of libpmhip.so (include/pmhip.h).

There is no CPU fallback: if the HIP library is missing or a kernel reports an error the
product path raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PM_LIB_PATH") or os.path.join(_HERE, "lib", "libpmhip.so")   # PM_LIB_PATH: A/B builds

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_GELU = 0, 1, 2, 3
AUX_AFTER_RES = 16  # PM_AUX_AFTER_RES (include/pmhip.h)
LEAKY_SLOPE = 0.01  # jax.nn.leaky_relu default negative_slope


class PmHipError(RuntimeError):
    pass


class GatherDesc(C.Structure):
    """struct pm_gather_desc (include/pmhip.h)."""

    _fields_ = [
        ("B", C.c_int),
        ("IH", C.c_int), ("IW", C.c_int), ("C", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int), ("N", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int),
        ("a", C.c_int), ("cs", C.c_int), ("off", C.c_int), ("d", C.c_int),
        ("wts", C.c_int), ("wcs", C.c_int), ("wns", C.c_int),
        ("groups", C.c_int),
        ("in_gs", C.c_longlong), ("w_gs", C.c_longlong), ("out_gs", C.c_longlong), ("bias_gs", C.c_longlong),
        ("in_act", C.c_int), ("out_act", C.c_int), ("aux_act", C.c_int),
        ("slope", C.c_float),
        ("off_x", C.c_int), ("kws", C.c_int),
    ]


class LossCfg(C.Structure):
    _fields_ = [
        ("beta_kind", C.c_int),
        ("low", C.c_float), ("high", C.c_float),
        ("period_or_steps", C.c_int), ("delay_or_begin", C.c_int),
        ("matching_coef", C.c_float),
        ("grad_scale", C.c_float),
    ]


class AdamCfg(C.Structure):
    _fields_ = [
        ("b1", C.c_float), ("b2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
        ("lr_init", C.c_float), ("lr_decay_rate", C.c_float), ("lr_transition_steps", C.c_float),
        ("grad_scale", C.c_float),
        ("lr_kind", C.c_int), ("lr_end", C.c_float), ("zero_grad", C.c_int),
    ]


class MaskComponent(C.Structure):
    """struct pm_mask_component (include/pmhip.h)."""

    _fields_ = [
        ("kind", C.c_int), ("p", C.c_float),
        ("y1", C.c_int), ("x1", C.c_int), ("y2", C.c_int), ("x2", C.c_int),
        ("size", C.c_int), ("min_prop", C.c_float), ("max_prop", C.c_float), ("cum_weight", C.c_float),
    ]


class VdvaeBlockIO(C.Structure):
    """struct pm_vdvae_block_io (include/pmhip.h)."""

    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p), ("res", C.c_void_p), ("xpre", C.c_void_p), ("xg_out", C.c_void_p),
        ("h", C.c_void_p * 3), ("g", C.c_void_p * 3), ("out", C.c_void_p),
        ("w", C.c_void_p * 4), ("plane", C.c_longlong * 4), ("bias", C.c_void_p * 4),
        ("Cin", C.c_int), ("Cout", C.c_int), ("Ca", C.c_int), ("dense_k3", C.c_int),
    ]


class SplitJob(C.Structure):
    """struct pm_split_job (include/pmhip.h)."""

    _fields_ = [
        ("src_off", C.c_longlong), ("dst_off", C.c_longlong), ("plane", C.c_longlong),
        ("taps", C.c_int), ("C", C.c_int), ("N", C.c_int), ("npad", C.c_int),
        ("wts", C.c_int), ("wcs", C.c_int), ("wns", C.c_int),
        ("first_block", C.c_int), ("num_blocks", C.c_int),
        ("kw", C.c_int), ("kws", C.c_int), ("dense_k", C.c_int),
    ]


class WgradPart(C.Structure):
    """struct pm_wgrad_part (include/pmhip.h): arenas of a weight-gradient launch in partial-sum mode."""

    _fields_ = [
        ("w", C.c_void_p), ("b", C.c_void_p), ("bg", C.c_void_p),
        ("w_stride", C.c_longlong), ("b_stride", C.c_longlong), ("bg_stride", C.c_longlong),
        ("nslots", C.c_int),
    ]


class ColsumJob(C.Structure):
    """struct pm_colsum_job (include/pmhip.h)"""
    _fields_ = [("x", C.c_void_p), ("part", C.c_void_p), ("part_stride", C.c_longlong), ("M", C.c_longlong),
                ("N", C.c_int), ("nslots", C.c_int)]


class ReduceJob(C.Structure):
    """struct pm_reduce_job (include/pmhip.h)."""

    _fields_ = [
        ("src", C.c_void_p), ("stride", C.c_longlong), ("g_off", C.c_longlong),
        ("count", C.c_int), ("nslots", C.c_int),
    ]


_P = C.c_void_p
_I = C.c_int
_LL = C.c_longlong
_F = C.c_float

# name -> argtypes ; every function returns int (0 = ok) unless listed in _OTHER_RESTYPE
SIGNATURES = {
    "pm_gather_gemm": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P],
    "pm_gather_wgrad": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P],
    "pm_gather_gemm_bf16": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P],
    "pm_gather_gemm_bf16_dual": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P, _P, _I],
    "pm_gemm_splitk_floats": [C.POINTER(GatherDesc), _I, _I, C.POINTER(C.c_longlong)],
    "pm_gather_gemm_sk": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P, _P, C.c_longlong],
    "pm_gather_gemm_bf16_sk": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P, _P, _I, _P, C.c_longlong],
    "pm_gather_gemm_bf16_insum": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P, _P],
    "pm_image_conv_insum_applies": [C.POINTER(GatherDesc)],
    "pm_split_weights": [_P, _P, _P, _P, _I, _I],
    "pm_gather_wgrad_bf16": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P],
    "pm_gather_wgrad_table": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _I, _I],
    "pm_wgrad_part_slots": [C.POINTER(GatherDesc), _P, _P, _I, _I, _I, C.POINTER(_I)],
    "pm_gather_wgrad_part": [_P, C.POINTER(GatherDesc), _P, _P, _P, _I, _I, C.POINTER(WgradPart)],
    "pm_thin_wgrad_part_slots": [C.POINTER(GatherDesc), _P, _P, C.POINTER(_I)],
    "pm_thin_wgrad_part": [_P, C.POINTER(GatherDesc), _P, _P, C.POINTER(WgradPart)],
    "pm_colsum_part_slots": [_LL, _I, C.POINTER(_I)],
    "pm_colsum_part": [_P, _P, _LL, _I, _P, _LL, _I],
    "pm_colsum_part_multi": [_P, _P, _I],
    "pm_reduce_partials": [_P, _P, _I, _P],
    "pm_adam_step_jobs": [_P, _P, _I, _P, _P, _P, _P, _LL, _P, C.POINTER(AdamCfg)],
    "pm_thin_conv": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P],
    "pm_thin_wgrad": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P],
    "pm_thin_to1_bf16": [_P, C.POINTER(GatherDesc), _P, _P, _P, _P, _P, _P],
    "pm_tap_shift_add": [_P, C.POINTER(GatherDesc), _P, _I, _P, _P],
    "pm_mask_concat": [_P, _P, _P, _P, _LL, _I, _I],
    "pm_tril_sample_kl_fwd": [_P, _P, _P, _P, _P, _I, _I],
    "pm_tril_sample_kl_bwd": [_P, _P, _P, _P, _P, _P, _I, _I],
    "pm_tril_logprob_fwd": [_P, _P, _P, _P, _I, _I],
    "pm_tril_logprob_bwd": [_P, _P, _P, _P, _P, _P, _I, _I],
    "pm_diag_gaussian_sample_kl_fwd": [_P, _P, _P, _P, _P, _I, _I],
    "pm_diag_gaussian_sample_kl_bwd": [_P, _P, _P, _P, _P, _P, _I, _I],
    "pm_diag_gaussian_logprob_fwd": [_P, _P, _P, _P, _I, _I],
    "pm_diag_gaussian_logprob_bwd": [_P, _P, _P, _P, _P, _P, _I, _I],
    "pm_bernoulli_ll_fwd": [_P, _P, _P, _P, _I, _I],
    "pm_bernoulli_ll_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _F],
    "pm_bernoulli_ll_fwd_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F],
    "pm_tril_sample_kl_bwd2": [_P, _P, _P, _P, _P, _P, _P, _I, _I],
    "pm_normal_ll_fwd": [_P, _P, _P, _P, _P, _I, _I, _F],
    "pm_normal_ll_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F],
    "pm_normal_ll_bwd_det": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "pm_vq_select": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F],
    "pm_vq_dw_exact": [_P, _P, _P, _P, _I, _I, _I, _P, C.c_longlong],
    "pm_embed_bwd_sorted": [_P, _P, _P, _P, C.c_longlong, _I, _I, _P, C.c_longlong],
    "pm_vq_dw_exact_floats": [_I, _I, _I, C.POINTER(C.c_longlong)],
    "pm_vq_ema_update": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F],
    "pm_vq_lookup": [_P, _P, _P, _P, _I, _I, _I],
    "pm_vqvae_loss": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P],
    "pm_argmm_build_input": [_P, _P, _P, _P, _I, _I, _I],
    "pm_argmm_input_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _F],
    "pm_gmm_logprob_fwd": [_P, _P, _P, _P, _I, _I, _I],
    "pm_gmm_logprob_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "pm_embed_fwd": [_P, _P, _P, _P, _LL, _I, _I],
    "pm_embed_bwd": [_P, _P, _P, _P, _LL, _I, _I],
    "pm_embed_bwd_exact": [_P, _P, _P, _P, _LL, _I, _I, _P],
    "pm_concat_elu_fwd": [_P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_concat_elu_bwd": [_P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P],
    "pm_gate_bwd_rows_sum": [_P, _P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_concat_elu_fwd_philox": [_P, _P, _P, _P, _LL, _I, _I, _F, C.c_ulonglong, _P, _I],
    "pm_concat_elu_bwd_philox": [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P, _F, C.c_ulonglong, _P, _I],
    "pm_gate_fwd": [_P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_gate_fwd_ce": [_P, _P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_gate_bwd": [_P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_rows_sum": [_P, _P, _P, _LL, _I, _I],
    "pm_rows_sum_multi": [_P, _P, _I, _P, C.c_longlong, _I, _I],
    "pm_groups_sum": [_P, _P, _P, _LL, _I, _LL, _I],
    "pm_elu_fwd": [_P, _P, _P, _LL],
    "pm_elu_bwd": [_P, _P, _P, _P, _LL, _I],
    "pm_categorical_ll_fwd": [_P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_categorical_ll_bwd": [_P, _P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_neg_mean_loss": [_P, _P, _I, _F, _P, _P],
    "pm_categorical_sample": [_P, _P, _P, _P, _LL, _I, _I, _I],
    "pm_impute_blend": [_P, _P, _P, _P, _LL, _I, _LL, _I, _I, _F, _F],
    "pm_imputation_psnr": [_P, _P, _P, _P, _LL, _I, _LL, _F],
    "pm_gumbel_fill": [_P, _P, _LL, C.c_ulonglong, _P, _I],
    "pm_mlp_pair_bf16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _F],
    "pm_info_gain_inputs": [_P, _P, _P, _P, _P, _I, _I, _I],
    "pm_gaussian_entropy": [_P, _P, _P, _LL, _I, _I, _I],
    "pm_info_gain_finish": [_P, _P, _P, _P, _I, _I],
    "pm_sample_project_fwd": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I],
    "pm_sample_project_bwd": [_P, _P, _P, _I, _P, _P, _P, _F, _P, _P, _LL, _I, _I],
    "pm_mlp_chain_bf16": [_P, _P, _I, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _F],
    "pm_repeat_rows": [_P, _P, _P, _LL, _I, _LL],
    "pm_sigmoid": [_P, _P, _P, _LL],
    "pm_bernoulli_ll_rep_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I],
    "pm_normal_ll_rep_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F],
    "pm_std_normal_logprob": [_P, _P, _P, _LL, _I],
    "pm_logmeanexp3": [_P, _P, _P, _P, _P, _I, _I, _I],
    "pm_gmm_sample_step": [_P, _P, _P, _P, _P, _LL, _I, _I, _I],
    "pm_diag_logprob_acc": [_P, _P, _I, _P, _P, _LL, _I, _I, _F],
    "pm_segment_wsum": [_P, _P, _P, _P, _I, _I, _F, _I],
    "pm_image_mask_mixture": [_P, _P, _I, _I, _I, C.POINTER(MaskComponent), _I, C.c_ulonglong, _P, _I, _P, _P,
                              C.c_ulonglong],
    "pm_bernoulli_mask": [_P, _P, _LL, _F, C.c_ulonglong, _P, _I],
    "pm_uniform_mask": [_P, _P, _I, _I, _I, _I, C.c_ulonglong, _P, _I],
    "pm_dropout_mask": [_P, _P, _LL, _F, C.c_ulonglong, _P, _I],
    "pm_gelu_fwd": [_P, _P, _P, _P, _LL, _I, _I],
    "pm_gelu_bwd": [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I],
    "pm_avgpool_fwd": [_P, _P, _P, _I, _I, _I, _I, _I],
    "pm_avgpool_bwd": [_P, _P, _P, _I, _I, _I, _I, _I],
    "pm_resize_nearest_add": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "pm_resize_nearest_add_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "pm_broadcast_rows": [_P, _P, _P, _LL, _LL],
    "pm_add_cols": [_P, _P, _P, _I, _I, _P, _LL, _I],
    "pm_copy_cols": [_P, _P, _P, _I, _I, _LL, _I],
    "pm_scale_shift": [_P, _P, _F, _F, _P, _LL],
    "pm_diag_sample_kl_fwd": [_P, _P, _P, _I, _P, _P, _P, _LL, _I, _I],
    "pm_diag_sample_kl_bwd": [_P, _P, _P, _I, _P, _P, _F, _P, _P, _LL, _I],
    "pm_diag_tril_kl_fwd": [_P, _P, _P, _P, _LL, _I, _I],
    "pm_diag_tril_kl_bwd": [_P, _P, _P, _F, _P, _LL, _I, _I],
    "pm_affine_fwd": [_P, _P, _P, _P, _P, _LL, _I],
    "pm_affine_bwd": [_P, _P, _P, _P, _P, _P, _P, _LL, _I],
    "pm_dmol_ll_fwd": [_P, _P, _P, _P, _LL, _I, _I, _F, _F],
    "pm_dmol_ll_bwd": [_P, _P, _P, _F, _P, _LL, _I, _I, _F, _F],
    "pm_dmol_mean": [_P, _P, _P, _LL, _I, _F, _F],
    "pm_vdvae_loss": [_P, _P, _P, _P, _I, _F, _P],
    "pm_sumsq": [_P, _P, _LL, _P],
    "pm_sumsq_det": [_P, _P, _LL, _P, _P],
    "pm_affine_bwd_part_slots": [_LL, C.POINTER(_I)],
    "pm_affine_bwd_part": [_P, _P, _P, _P, _P, _P, _P, _LL, _I, _LL, _I],
    "pm_adam_step_clip_ema": [_P, _P, _P, _P, _P, _P, _LL, _LL, _P, _P, C.POINTER(AdamCfg), _F, _F, _I],
    "pm_pmvae_loss": [_P, _P, _P, _P, _I, C.POINTER(LossCfg), _P, _P, _P, _P, _P],
    "pm_adam_step": [_P, _P, _P, _P, _P, _LL, _LL, _P, C.POINTER(AdamCfg)],
    "pm_counter_increment": [_P, _P],
    "pm_counter_snapshot_increment": [_P, _P, _P],
    "pm_stamp": [_P, _P],
    "pm_normal_fill": [_P, _P, _LL, C.c_ulonglong, _P, _I],
    "pm_fill_zero": [_P, _P, _LL],
    "pm_axpy1": [_P, _P, _P, _LL],
    "pm_colsum": [_P, _P, _P, _LL, _I],
    "pm_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _LL, _I, _F],
    "pm_layernorm_bwd": [_P, _P, _P, _P, _P, _LL, _I],
    "pm_relu_mask_fwd": [_P, _P, _P, _P, _LL],
    "pm_relu_mask_bwd": [_P, _P, _P, _P, _P, _LL],
    "pm_random_indices": [_P, _P, _I, _I, C.c_ulonglong, _P, _I],
    "pm_gather_u8_rows": [_P, _P, _P, _P, _I, _LL, _F],
    "pm_vdvae_blocks_fwd": [_P, C.POINTER(VdvaeBlockIO), _I, _I, _I, _I, _I, _I],
    "pm_vdvae_blocks_bwd": [_P, C.POINTER(VdvaeBlockIO), _I, _I, _I, _I, _I, _I],
    "pm_vade_prior_fwd": [_P, _P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_vade_prior_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I],
    "pm_vade_cluster_probs": [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I],
    "pm_lookahead_inputs": [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _I],
    "pm_lookahead_ll_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "pm_lookahead_ll_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "pm_lookahead_info_gains": [_P, _P, _P, _P, _P, _I, _I],
    "pm_acquisition_policy": [_P, _P, _P, _P, _I],
    "pm_reconstruction_rmse": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "pm_graph_begin": [_P],
    "pm_graph_end": [_P, C.POINTER(_P)],
    "pm_graph_launch": [_P, _P],
    "pm_graph_destroy": [_P],
    "pm_event_create": [C.POINTER(_P)],
    "pm_event_record": [_P, _P],
    "pm_event_synchronize": [_P],
    "pm_event_elapsed_ms": [_P, _P, C.POINTER(_F)],
    "pm_event_destroy": [_P],
    "pm_query_gemm_plan": [C.POINTER(GatherDesc), _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)],
    "pm_query_wgrad_plan": [C.POINTER(GatherDesc), _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I),
                            C.POINTER(_I)],
    "pm_version": [],
}
_OTHER_RESTYPE = {"pm_strerror": ([_I], C.c_char_p), "pm_last_error": ([], C.c_char_p),
                  "pm_kernel_names_enable": ([_I], None), "pm_last_kernel_name": ([], C.c_char_p),
                  "pm_last_kernel_variant": ([], C.c_char_p),
                  "pm_clear_kernel_name": ([], None)}

_lib = None


def load() -> C.CDLL:
    """Loads libpmhip.so once; raises PmHipError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PmHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C posterior_matching_amd/csrc` (there is no CPU fallback)")
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import it first so that
    # libpmhip.so binds to the SAME HIP runtime that owns torch's streams and allocations; two
    # runtimes in one process cannot see each other's devices ("no ROCm-capable device").
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (argtypes, restype) in _OTHER_RESTYPE.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        raise PmHipError(f"{what} failed: {lib.pm_strerror(rc).decode()} [{lib.pm_last_error().decode()}]")
