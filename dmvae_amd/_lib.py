"""ctypes binding of libdmvae_hip.so (the C ABI declared in include/dmvae_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C dmvae_amd/csrc``).
Loading fails loudly: there is no Python/CPU fallback for any entry point.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_longlong, c_size_t, c_uint, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMVAE_LIB") or os.path.join(_HERE, "libdmvae_hip.so")      # DMVAE_LIB: another build of the same ABI (kernel A/B runs on one box)


class ConvDesc(Structure):
    """struct dmvae_conv_desc (include/dmvae_hip.h)."""
    _fields_ = [("n", c_int32), ("h", c_int32), ("w", c_int32), ("cin", c_int32), ("cout", c_int32),
                ("ks", c_int32), ("upsample", c_int32), ("act", c_int32), ("out_f32", c_int32), ("stride", c_int32), ("transposed", c_int32),
                ("w_layout", c_int32)]


class PackEntry(Structure):
    """struct dmvae_pack_entry (include/dmvae_hip.h)."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("dst2", c_void_p),
                ("cout", c_int32), ("cin", c_int32), ("T", c_int32), ("rows_pad", c_int32), ("cols_pad", c_int32), ("mode", c_int32), ("subpixel", c_int32),
                ("reserved", c_int32), ("start", c_ulonglong), ("count", c_ulonglong)]


class WtEntry(Structure):
    """record of dmvae_linear_weight_t_kmajor_batched's table (include/dmvae_hip.h)."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("N", c_int32), ("K", c_int32), ("start", c_uint), ("tiles_x", c_uint)]


ABI_VERSION = 8     # 8: dmvae_reparam_kl_*, dmvae_linear_bf16_sk*, the f32 transformer steps of csrc/parity_dit.hip; 7: dmvae_linear_wgrad_grouped_plan / _xcd, dmvae_conv_k4c1_*; 6: dmvae_dit_stack_* / dmvae_dit_boundary_bwd / batched rows Linears / batched weight transposes; 5: dmvae_groupnorm_*_short; include/dmvae_hip.h: dmvae_abi_version (3: struct dmvae_pack_entry, dmvae_pack_weights_batched, dmvae_linear_bf16*; 4: dmvae_norm_conv_out_bwd*)

# name -> (restype, argtypes); every symbol include/dmvae_hip.h declares
SIGNATURES = {
    "dmvae_last_error": (c_char_p, []),
    "dmvae_abi_version": (c_int, []),
    "dmvae_rms_modulate_fwd_f32": (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_int, c_int, c_float, c_void_p]),
    "dmvae_rms_modulate_bwd_f32": (c_int, [c_void_p] * 8 + [c_size_t, c_int, c_int, c_int, c_void_p]),
    "dmvae_layernorm_bwd_full_f32": (c_int, [c_void_p] * 5 + [c_size_t, c_int, c_float, c_void_p]),
    "dmvae_bcast_rows_f32": (c_int, [c_int] + [c_void_p] * 4 + [c_size_t, c_int, c_int, c_int, c_void_p]),
    "dmvae_colsum_groups_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_swiglu_fwd_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dmvae_swiglu_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dmvae_qknorm_rope_fwd_f32": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_float, c_void_p]),
    "dmvae_qknorm_rope_bwd_f32": (c_int, [c_void_p] * 12 + [c_int] * 5 + [c_void_p]),
    "dmvae_reparam_kl_workspace": (c_size_t, [c_size_t, c_int]),
    "dmvae_reparam_kl_fwd": (c_int, [c_void_p] * 5 + [c_size_t, c_size_t, c_int, c_int, c_void_p]),
    "dmvae_reparam_kl_bwd": (c_int, [c_void_p] * 4 + [c_float, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "dmvae_conv2d_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_void_p]),
    "dmvae_conv2d_nhwc_fwd_gnstats_workspace": (c_size_t, [POINTER(ConvDesc), c_int]),
    "dmvae_conv2d_nhwc_fwd_gnstats": (c_int, [c_void_p] * 7 + [c_size_t, c_int, c_float, POINTER(ConvDesc), c_void_p]),
    "dmvae_conv2d_nhwc_wgrad_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "dmvae_gemm_nt_batched": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_longlong] * 3 + [c_int, c_int, c_void_p]),
    "dmvae_gemm_tn_batched_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dmvae_gemm_tn_batched": (c_int, [c_void_p] * 4 + [c_size_t] + [c_int] * 4 + [c_longlong] * 3 + [c_float, c_int, c_void_p]),
    "dmvae_set_dynamic": (c_int, [c_int]),
    "dmvae_linear_rows_supported": (c_int, [c_int, c_int, c_int]),
    "dmvae_linear_rows_bf16": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "dmvae_linear_rows_wgrad": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "dmvae_linear_bf16": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "dmvae_linear_bf16_splitk_supported": (c_int, [c_int] * 4),
    "dmvae_linear_bf16_splitk": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    "dmvae_splitk_sum_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "dmvae_linear_bf16_swiglu_pre": (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p]),
    "dmvae_linear_bf16_sk_supported": (c_int, [c_int] * 5),
    "dmvae_linear_bf16_sk_counter_bytes": (c_size_t, []),
    "dmvae_linear_bf16_sk_workspace": (c_size_t, [c_int] * 5),
    "dmvae_linear_bf16_sk": (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 11 + [c_void_p]),
    "dmvae_linear_bf16_batched_supported": (c_int, [c_int] * 4),
    "dmvae_linear_bf16_batched": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_longlong] * 3 + [c_int, c_void_p]),
    "dmvae_linear_bf16_plan": (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "dmvae_linear_weight_t_kmajor": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dmvae_pack_entry_bytes": (c_size_t, []),
    "dmvae_pack_weights_batched": (c_int, [c_void_p, c_int, c_ulonglong, c_int, c_void_p]),
    "dmvae_softmax_rows_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "dmvae_softmax_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "dmvae_transpose_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dmvae_pack_conv_weight": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "dmvae_pack_conv_weight_v2": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "dmvae_conv_halo_applies": (c_int, [POINTER(ConvDesc)]),
    "dmvae_conv_kmajor_applies": (c_int, [POINTER(ConvDesc)]),
    "dmvae_subpixel_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dmvae_subpixel_weight_fold": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dmvae_colsum_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p]),
    "dmvae_conv_out_wgrad_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dmvae_conv_out_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_sumpool2x2_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_maxpool2x2_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_maxpool2x2_relu_bwd_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dmvae_leaky_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "dmvae_sde_euler_step": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t] + [c_float] * 6 + [c_void_p]),
    "dmvae_image_to_u8": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "dmvae_batchnorm_running_update": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_void_p]),
    "dmvae_diffaug_fwd": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    "dmvae_diffaug_bwd": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    "dmvae_im2col_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "dmvae_im2col_nhwc_taps": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "dmvae_im2col_nhwc_sub": (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "dmvae_col2im_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "dmvae_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_nhwc_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_silu_fwd": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "dmvae_silu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dmvae_layernorm_f32_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "dmvae_scale_residual_layernorm": (c_int, [c_void_p] * 6 + [c_int, c_int, c_float, c_void_p]),
    "dmvae_scale_residual_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dmvae_softmax_rows_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, c_void_p]),
    "dmvae_attention_qkv_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dmvae_attention_heads_bf16": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float, c_void_p]),
    "dmvae_attention_qknorm_rope_bf16": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_float, c_float, c_void_p]),
    "dmvae_attention_bwd_qkv_bf16": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float, c_void_p]),
    "dmvae_attention_bwd_heads_bf16": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_float, c_void_p]),
    "dmvae_attention_qkv_lse_bf16": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float, c_void_p]),
    "dmvae_attention_heads_lse_bf16": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_float, c_void_p]),
    "dmvae_attention_bwd_qkv_lse_bf16": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_float, c_void_p]),
    "dmvae_attention_bwd_heads_lse_bf16": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_float, c_void_p]),
    "dmvae_vit_bwd_workspace": (c_size_t, [c_int]),
    "dmvae_layernorm_bwd_f32": (c_int, [c_void_p] * 7 + [c_size_t, c_int, c_int, c_float, c_int, c_void_p]),
    "dmvae_layerscale_bwd": (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_int, c_int, c_void_p]),
    "dmvae_gelu_fwd": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "dmvae_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dmvae_rmsnorm_modulate_bf16": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float, c_void_p]),
    "dmvae_gated_residual_rmsnorm_modulate": (c_int, [c_void_p] * 3 + [c_int] * 2 + [c_void_p] * 3 + [c_int] * 6 + [c_float, c_void_p]),
    "dmvae_gated_residual_out": (c_int, [c_void_p] * 4 + [c_int] * 2 + [c_void_p] * 3 + [c_int] * 6 + [c_float, c_void_p]),
    "dmvae_qknorm_rope_bf16": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_float, c_void_p]),
    "dmvae_swiglu_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dmvae_dit_bwd_workspace": (c_size_t, [c_int, c_int]),
    "dmvae_gated_residual_bwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "dmvae_swiglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "dmvae_rmsnorm_modulate_bwd": (c_int, [c_void_p] * 8 + [c_size_t] + [c_int] * 6 + [c_float, c_int, c_void_p]),
    "dmvae_qknorm_rope_bwd": (c_int, [c_void_p] * 12 + [c_size_t] + [c_int] * 5 + [c_float, c_int, c_void_p]),
    "dmvae_qknorm_rope_bwd_nblk": (c_int, [c_int] * 5),
    "dmvae_qknorm_rope_bwd_partial": (c_int, [c_void_p] * 10 + [c_size_t] + [c_int] * 5 + [c_float, c_void_p]),
    "dmvae_dit_stack_bps": (c_int, [c_int]),
    "dmvae_dit_stack_part_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dmvae_dit_stack_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dmvae_dit_boundary_bwd": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dmvae_dit_stack_finalize": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p] + [c_int] * 5 + [c_void_p]),
    "dmvae_colsum2_batched": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "dmvae_linear_rows_batched_bf16": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong] + [c_int] * 11 + [c_void_p]),
    "dmvae_linear_rows_wgrad_batched": (c_int, [c_void_p, c_longlong, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "dmvae_wt_entry_bytes": (c_size_t, []),
    "dmvae_linear_weight_t_kmajor_batched": (c_int, [c_void_p, c_int, c_uint, c_void_p]),
    "dmvae_linear_wgrad_grouped_entry_bytes": (c_size_t, []),
    "dmvae_linear_wgrad_grouped_bias_entry_bytes": (c_size_t, []),
    "dmvae_linear_wgrad_grouped_supported": (c_int, [c_int] * 3),
    "dmvae_linear_wgrad_grouped_bias_parts": (c_int, [c_int]),
    "dmvae_linear_wgrad_grouped_fill": (c_int, [c_void_p] * 7 + [c_int] * 3 + [POINTER(c_uint), POINTER(c_uint)]),
    "dmvae_linear_wgrad_grouped": (c_int, [c_void_p, c_int, c_uint, c_int, c_void_p, c_int, c_uint, c_void_p]),
    "dmvae_linear_wgrad_grouped_chunk_bytes": (c_size_t, []),
    "dmvae_linear_wgrad_grouped_plan": (c_int, [c_void_p, c_int, c_void_p, c_int, POINTER(c_int), POINTER(c_uint), POINTER(c_uint)]),
    "dmvae_linear_wgrad_grouped_xcd": (c_int, [c_void_p, c_void_p, POINTER(c_uint), c_uint, c_int, c_void_p, c_int, c_uint, c_void_p]),
    "dmvae_gated_residual_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_loss_workspace": (c_size_t, []),
    "dmvae_l1_mse": (c_int, [c_void_p] * 5 + [c_size_t, c_size_t, c_float, c_float, c_void_p]),
    "dmvae_lpips_diff": (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dmvae_lpips_diff_pool": (c_int, [c_void_p] * 8 + [c_size_t, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dmvae_dmd_pre": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "dmvae_dmd_post": (c_int, [c_void_p] * 10 + [c_size_t, c_int, c_int, c_float, c_int, c_void_p]),
    "dmvae_kl_mmd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "dmvae_kl_mmd": (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "dmvae_grad_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_int, c_void_p]),
    "dmvae_adamw_ema_step": (c_int, [c_void_p] * 6 + [c_size_t] + [c_float] * 5 + [c_int, c_float, c_void_p]),
    "dmvae_adamw_ema_step_shadow": (c_int, [c_void_p] * 7 + [c_size_t] + [c_float] * 5 + [c_int, c_float, c_void_p]),
    "dmvae_groupnorm_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dmvae_groupnorm_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dmvae_groupnorm_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_groupnorm_bwd": (c_int, [c_void_p] * 10 + [c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_groupnorm_bwd_colsum": (c_int, [c_void_p] * 11 + [c_size_t] + [c_int] * 7 + [c_void_p]),
    "dmvae_groupnorm_bwd_reduce": (c_int, [c_void_p] * 9 + [c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_groupnorm_bwd_apply": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dmvae_groupnorm_short_supported": (c_int, [c_int] * 5),
    "dmvae_groupnorm_apply_short": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "dmvae_groupnorm_bwd_short_workspace": (c_size_t, [c_int] * 5),
    "dmvae_groupnorm_bwd_short": (c_int, [c_void_p] * 12 + [c_size_t] + [c_int] * 8 + [c_void_p]),
    "dmvae_conv_k4c1_supported": (c_int, [c_int] * 4),
    "dmvae_conv_k4c1_wgrad_workspace": (c_size_t, [c_int]),
    "dmvae_conv_k4c1_fwd": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "dmvae_conv_k4c1_dgrad": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "dmvae_conv_k4c1_wgrad": (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 4 + [c_void_p]),
    "dmvae_conv_in3_supported": (c_int, [c_int] * 4),
    "dmvae_conv_in3_workspace": (c_size_t, [c_int] * 3),
    "dmvae_conv_in3": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 6 + [c_size_t] + [c_int] * 5 + [c_void_p]),
    "dmvae_conv_to_image_supported": (c_int, [c_int] * 5),
    "dmvae_conv_to_image": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "dmvae_norm_conv_out_fwd_supported": (c_int, [c_int] * 6),
    "dmvae_norm_conv_out_fwd": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "dmvae_norm_conv_out_bwd_supported": (c_int, [c_int] * 6),
    "dmvae_norm_conv_out_bwd_workspace": (c_size_t, [c_int] * 5),
    "dmvae_norm_conv_out_bwd": (c_int, [c_void_p] * 10 + [c_size_t] + [c_int] * 7 + [c_void_p]),
    "dmvae_conv2d_nhwc_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(ConvDesc), c_int, c_void_p]),
    # fp32 parity mode (csrc/parity.hip)
    "dmvae_split3_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_void_p]),
    "dmvae_groupnorm_stats_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "dmvae_groupnorm_apply_f32": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "dmvae_groupnorm_f32_workspace": (c_size_t, [c_int, c_int, c_int]),
    "dmvae_groupnorm_bwd_f32": (c_int, [c_void_p] * 10 + [c_size_t] + [c_int] * 6 + [c_float, c_void_p]),
    "dmvae_eltwise_f32": (c_int, [c_int] + [c_void_p] * 4 + [c_size_t, c_int, c_int, c_float, c_void_p]),
    "dmvae_softmax_rows_fwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "dmvae_softmax_rows_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "dmvae_pool2x2_f32": (c_int, [c_int] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "dmvae_nchw_f32_to_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dmvae_lpips_diff_f32": (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dmvae_layernorm_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float, c_void_p]),
}

_lib = None


class DmvaeHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DmvaeHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dmvae_amd/csrc` (there is no CPU fallback)")
        l = ctypes.CDLL(LIB_PATH)
        # the version gate comes BEFORE the symbols are bound: an older build handed in through DMVAE_LIB must say "rebuild", not die on a missing name,
        # and a build that does export every name must still not be called with structs laid out for another version
        l.dmvae_abi_version.restype, l.dmvae_abi_version.argtypes = c_int, []
        if l.dmvae_abi_version() != ABI_VERSION:
            raise DmvaeHipError(f"{LIB_PATH}: ABI version {l.dmvae_abi_version()}, this package binds version {ABI_VERSION}; rebuild the library")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name, None)
            if fn is None:
                raise DmvaeHipError(f"{LIB_PATH} does not export {name} (declared in include/dmvae_hip.h); rebuild the library")
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().dmvae_last_error()
        raise DmvaeHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
