"""ctypes binding of the gfx950 C-ABI library (include/diffute_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Nothing under oracle/ is ever imported from here.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p, POINTER

# DIFFUTE_HIP_LIB: A/B builds of the same library (kernel experiments); the default is the in-tree build
_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_LIB_PATH = os.environ.get("DIFFUTE_HIP_LIB") or os.path.join(_LIB_DIR, "libdiffute_hip.so")
# the same sources compiled with -DDMX_F16: fp16 storage / MFMA operands (BASELINE configs[4]; `.to(dtype=torch.float16)`)
_LIB_PATH_F16 = os.environ.get("DIFFUTE_HIP_LIB_F16") or os.path.join(_LIB_DIR, "libdiffute_hip_f16.so")
_lib = None
_lib_f16 = None
_exclusive = None        # set_exclusive_device(): None = the library default (1)
_last_elem = "bf16"      # the build handed out last: check() reads ITS error message (the call that failed went through it)


class XfChainDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_int), ("C", c_int), ("x", c_void_p), ("ldx", c_int), ("res", c_void_p), ("ldres", c_int),
        ("w0", c_void_p), ("b0", c_void_p), ("h_out", c_void_p), ("ldh", c_int), ("w1", c_void_p),
        ("c1", c_void_p), ("c2", c_void_p), ("y", c_void_p), ("ldy", c_int), ("wf1", c_void_p),
        ("wf2", c_void_p), ("bf2", c_void_p), ("wpo", c_void_p), ("bpo", c_void_p), ("xres", c_void_p), ("ldxres", c_int),
        ("eps", c_float), ("dbg", c_int), ("timing", c_void_p), ("colstats", c_void_p), ("cs_rows", c_int),
        ("gn_st", c_void_p), ("gn_gamma", c_void_p), ("gn_beta", c_void_p), ("gn_groups", c_int), ("gn_rows", c_int), ("gn_eps", c_float),
    ]


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("x0", c_void_p), ("x1", c_void_p), ("ldx0", c_int), ("ldx1", c_int), ("cx0", c_int),
        ("direct", c_int), ("IH", c_int), ("IW", c_int), ("OH", c_int), ("OW", c_int),
        ("stride", c_int), ("pad", c_int), ("ups", c_int), ("ksize", c_int), ("Cin", c_int), ("Ktaps", c_int),
        ("s0", c_void_p), ("s1", c_void_p), ("lds0", c_int), ("lds1", c_int), ("cs0", c_int),
        ("w", c_void_p), ("ldw", c_int), ("M", c_int), ("N", c_int), ("K", c_int),
        ("bias", c_void_p), ("rowbias", c_void_p), ("rows_per_group", c_int), ("ldrb", c_int),
        ("res", c_void_p), ("ldres", c_int), ("out", c_void_p), ("ldo", c_int), ("out_f32", c_int), ("geglu", c_int),
        ("force_tn", c_int), ("force_splitk", c_int), ("group_m", c_int), ("timing", c_void_p), ("dbg", c_int), ("act", c_int),
        ("rowstats_out", c_void_p), ("ln_stats", c_void_p), ("ln_tiles", c_int), ("ln_c1", c_void_p), ("ln_c2", c_void_p),
        ("ln_C", c_int), ("ln_eps", c_float), ("colstats", c_void_p), ("cs_rows", c_int),
    ]


class HaloConvDesc(ctypes.Structure):
    _fields_ = [
        ("x0", c_void_p), ("x1", c_void_p), ("ldx0", c_int), ("ldx1", c_int), ("cx0", c_int), ("Cin", c_int),
        ("B", c_int), ("H", c_int), ("W", c_int), ("gn", c_int), ("silu", c_int), ("groups", c_int), ("eps", c_float),
        ("st0", c_void_p), ("st1", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
        ("s0", c_void_p), ("s1", c_void_p), ("lds0", c_int), ("lds1", c_int), ("cs0", c_int), ("Csc", c_int),
        ("w", c_void_p), ("ldw", c_int), ("N", c_int), ("bias", c_void_p), ("rowbias", c_void_p), ("ldrb", c_int),
        ("res", c_void_p), ("ldres", c_int), ("out", c_void_p), ("ldo", c_int), ("colstats", c_void_p),
        ("force_split", c_int), ("force_bn", c_int), ("force_waves", c_int), ("dbg", c_int), ("timing", c_void_p),
    ]


class SkinnySeg(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("ld", c_int), ("C", c_int), ("taps", c_int), ("st", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("gn_c0", c_int)]


class SkinnyDesc(ctypes.Structure):
    _fields_ = [("seg", SkinnySeg * 4), ("nseg", c_int), ("B", c_int), ("H", c_int), ("W", c_int),
                ("gn_groups", c_int), ("gn_Ctot", c_int), ("gn_eps", c_float), ("silu", c_int),
                ("wp", c_void_p), ("N", c_int), ("bias", c_void_p), ("rowbias", c_void_p), ("ldrb", c_int),
                ("res", c_void_p), ("ldres", c_int), ("out", c_void_p), ("ldo", c_int), ("colstats", c_void_p), ("force_S", c_int), ("timing", c_void_p), ("dbg", c_int)]


class ViTConfig(ctypes.Structure):
    _fields_ = [("image_size", c_int), ("patch_size", c_int), ("num_channels", c_int), ("hidden_size", c_int), ("num_layers", c_int),
                ("num_heads", c_int), ("intermediate_size", c_int), ("qkv_bias", c_int), ("layer_norm_eps", c_float)]


class UNetConfig(ctypes.Structure):
    _fields_ = [("in_channels", c_int), ("out_channels", c_int), ("block_out_channels", c_int * 4),
                ("layers_per_block", c_int), ("heads", c_int * 4), ("cross_attention_dim", c_int),
                ("norm_num_groups", c_int), ("down_has_attn", c_int * 4), ("up_has_attn", c_int * 4)]


class VAEConfig(ctypes.Structure):
    _fields_ = [("in_channels", c_int), ("out_channels", c_int), ("latent_channels", c_int),
                ("block_out_channels", c_int * 4), ("layers_per_block", c_int), ("norm_num_groups", c_int)]


_P = c_void_p
_PROTOS = {
    "dmx_version": (c_int, []),
    "dmx_last_error": (c_char_p, []),
    "dmx_device_error": (c_int, []),
    "dmx_test_raise_device_error": (c_int, [c_int, c_void_p]),
    "dmx_test_occupy_cus": (c_int, [c_int, c_int64, c_void_p]),
    "dmx_element_type": (c_char_p, []),
    "dmx_conv_gemm_workspace_bytes": (c_size_t, [POINTER(GemmDesc)]),
    "dmx_conv_gemm": (c_int, [POINTER(GemmDesc), _P, c_size_t, _P]),
    "dmx_conv_gemm_rowstats_tiles": (c_int, [POINTER(GemmDesc)]),
    "dmx_conv_gemm_colstats_ok": (c_int, [POINTER(GemmDesc)]),
    "dmx_set_gn_producer_stats": (c_int, [c_int]),
    "dmx_conv3x3_gn_supported": (c_int, [POINTER(HaloConvDesc)]),
    "dmx_conv3x3_gn_workspace_bytes": (c_size_t, [POINTER(HaloConvDesc)]),
    "dmx_conv3x3_gn": (c_int, [POINTER(HaloConvDesc), _P, c_size_t, _P]),
    "dmx_colstats": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "dmx_skinny_conv_supported": (c_int, [POINTER(SkinnyDesc)]),
    "dmx_skinny_conv_workspace_bytes": (c_size_t, [POINTER(SkinnyDesc)]),
    "dmx_skinny_conv": (c_int, [POINTER(SkinnyDesc), _P, c_size_t, _P]),
    "dmx_skinny_pack": (c_int, [_P, c_int, _P, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), _P]),
    "dmx_set_skinny": (c_int, [c_int]),
    "dmx_set_halo_conv": (c_int, [c_int]),
    "dmx_set_halo_ws": (c_int, [c_int]),
    "dmx_set_halo_peers": (c_int, [c_int]),
    "dmx_set_attn_balanced": (c_int, [c_int]),
    "dmx_attention_fwd_v_balanced_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dmx_attention_fwd_v_balanced": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                             c_void_p, c_size_t, c_void_p]),
    "dmx_set_defer_reduce": (c_int, [c_int]),
    "dmx_set_exclusive_device": (c_int, [c_int]),
    "dmx_get_exclusive_device": (c_int, []),
    "dmx_plan_epoch": (c_int, []),
    "dmx_xf_chain_ok": (c_int, [c_int, c_int]),
    "dmx_xf_chain": (c_int, [POINTER(XfChainDesc), c_int, _P]),
    "dmx_set_xf_chain": (c_int, [c_int]),
    "dmx_set_weight_prefetch": (c_int, [c_int]),
    "dmx_groupnorm_from_stats": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, c_int, _P, _P, _P, c_int, _P]),
    "dmx_conv_wgrad_workspace_bytes": (c_size_t, [POINTER(GemmDesc), c_int]),
    "dmx_conv_wgrad": (c_int, [POINTER(GemmDesc), _P, c_int, _P, c_int, _P, c_size_t, _P]),
    "dmx_colsum_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dmx_colsum": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_size_t, _P]),
    "dmx_groupnorm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dmx_groupnorm": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, c_int, _P, c_int, _P, c_size_t, _P]),
    "dmx_groupnorm_train": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    "dmx_groupnorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dmx_groupnorm_bwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P,
                                  _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "dmx_layernorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dmx_layernorm_bwd": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "dmx_geglu_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "dmx_geglu_bwd": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P]),
    "dmx_layernorm": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, c_float, _P]),
    "dmx_attention_fwd": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "dmx_attention_fwd_v": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "dmx_attention_wide": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "dmx_attention_fwd_train": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "dmx_attention_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dmx_attention_bwd": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, _P, _P, c_int, _P, _P, c_int, _P, c_int, _P, c_int,
                                  c_int, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "dmx_timestep_embedding": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P]),
    "dmx_linear_small": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dmx_im2col_small": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "dmx_pack_conv_weight": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dmx_pack_linear_weight": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dmx_pack_geglu_bias": (c_int, [_P, _P, c_int, _P]),
    "dmx_pack_conv_weight_t": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dmx_pack_linear_weight_t": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "dmx_zero_insert2": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "dmx_sumpool2": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dmx_cast_f32_to_bf16": (c_int, [_P, _P, c_size_t, _P]),
    "dmx_nhwc_bf16_to_nchw_f32": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "dmx_nhwc_f32_to_nchw_f32": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "dmx_nchw_f32_to_nhwc_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dmx_sched_step_ddim": (c_int, [_P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "dmx_sched_step_ddpm": (c_int, [_P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "dmx_sched_add_noise": (c_int, [_P, _P, _P, _P, _P, c_int, c_size_t, _P]),
    "dmx_sched_get_velocity": (c_int, [_P, _P, _P, _P, _P, c_int, c_size_t, _P]),
    "dmx_gaussian_sample": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "dmx_vit_create": (_P, [POINTER(ViTConfig)]),
    "dmx_vit_destroy": (None, [_P]),
    "dmx_vit_param_count": (c_int, [_P]),
    "dmx_vit_param_info": (c_int, [_P, c_int, POINTER(c_char_p), POINTER(c_int * 4)]),
    "dmx_vit_arena_bytes": (c_size_t, [_P]),
    "dmx_vit_bind_arena": (c_int, [_P, _P, c_size_t]),
    "dmx_vit_load_param": (c_int, [_P, c_char_p, _P, _P]),
    "dmx_vit_finalize": (c_int, [_P, _P]),
    "dmx_vit_workspace_bytes": (c_size_t, [_P, c_int]),
    "dmx_vit_forward": (c_int, [_P, _P, _P, c_int, _P, c_size_t, _P]),
    "dmx_vit_master_bytes": (c_size_t, [_P]),
    "dmx_vit_master_import": (c_int, [_P, _P, c_char_p, _P, _P]),
    "dmx_vit_workspace_bytes_f32": (c_size_t, [_P, c_int]),
    "dmx_vit_forward_f32": (c_int, [_P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "dmx_unet_create": (_P, [POINTER(UNetConfig)]),
    "dmx_unet_destroy": (None, [_P]),
    "dmx_unet_param_count": (c_int, [_P]),
    "dmx_unet_param_info": (c_int, [_P, c_int, POINTER(c_char_p), POINTER(c_int * 4)]),
    "dmx_unet_arena_bytes": (c_size_t, [_P]),
    "dmx_unet_bind_arena": (c_int, [_P, _P, c_size_t]),
    "dmx_unet_load_param": (c_int, [_P, c_char_p, _P, _P]),
    "dmx_unet_finalize": (c_int, [_P, _P, _P]),
    "dmx_unet_context_bytes": (c_size_t, [_P, c_int, c_int]),
    "dmx_unet_workspace_bytes": (c_size_t, [_P, c_int, c_int, c_int, c_int]),
    "dmx_unet_set_context": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_size_t, _P, c_size_t, _P]),
    "dmx_unet_forward": (c_int, [_P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_unet_forward_taps": (c_int, [_P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_size_t,
                                      _P, c_size_t, _P, _P, _P]),
    "dmx_unet_workspace_bytes_f32": (c_size_t, [_P, c_int, c_int, c_int, c_int]),
    "dmx_unet_forward_f32": (c_int, [_P, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_size_t,
                                     _P, c_size_t, _P, _P, _P]),
    "dmx_unet_forward_graph": (c_int, [_P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_unet_train_workspace_bytes": (c_size_t, [_P, c_int, c_int, c_int, c_int]),
    "dmx_unet_train_wt_bytes": (c_size_t, [_P]),
    "dmx_unet_train_prepare": (c_int, [_P, _P, c_size_t, _P]),
    "dmx_unet_grad_bytes": (c_size_t, [_P]),
    "dmx_unet_train_forward": (c_int, [_P, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_unet_train_bucket_count": (c_int, [_P]),
    "dmx_unet_train_bucket_range": (c_int, [_P, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "dmx_unet_train_tail_range": (c_int, [_P, POINTER(c_size_t), POINTER(c_size_t)]),
    "dmx_unet_train_backward": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "dmx_unet_grad_export": (c_int, [_P, _P, c_char_p, _P, _P]),
    "dmx_unet_grad_range": (c_int, [_P, c_char_p, POINTER(c_size_t), POINTER(c_size_t)]),
    "dmx_vae_train_workspace_bytes": (c_size_t, [_P, c_int, c_int, c_int]),
    "dmx_vae_train_wt_bytes": (c_size_t, [_P]),
    "dmx_vae_train_prepare": (c_int, [_P, _P, c_size_t, _P]),
    "dmx_vae_grad_bytes": (c_size_t, [_P]),
    "dmx_vae_train_forward": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_vae_train_backward": (c_int, [_P, _P, _P, _P]),
    "dmx_vae_grad_export": (c_int, [_P, _P, c_char_p, _P, _P]),
    "dmx_unet_optim_chunks": (c_int, [_P]),
    "dmx_unet_optim_table_bytes": (c_size_t, [_P]),
    "dmx_unet_optim_elements": (c_size_t, [_P]),
    "dmx_unet_optim_table": (c_int, [_P, _P, c_size_t, _P]),
    "dmx_unet_master_import": (c_int, [_P, _P, c_char_p, _P, _P]),
    "dmx_vae_master_import": (c_int, [_P, _P, c_char_p, _P, _P]),
    "dmx_vae_workspace_bytes_f32": (c_size_t, [_P, c_int, c_int, c_int, c_int]),
    "dmx_vae_encode_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_vae_decode_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_unet_adamw_step": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P, _P, c_size_t, _P, c_float, _P]),
    "dmx_unet_adamw_step_scaled": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P, _P, c_size_t, _P, c_float, c_float, _P]),
    "dmx_unet_refresh_derived": (c_int, [_P, _P]),
    "dmx_unet_temb_table_floats": (c_size_t, [_P, c_int]),
    "dmx_unet_temb_table_workspace_bytes": (c_size_t, [_P, c_int]),
    "dmx_unet_temb_table": (c_int, [_P, _P, c_int, _P, _P, c_size_t, _P]),
    "dmx_unet_use_temb_table": (c_int, [_P, _P, _P]),
    "dmx_mask_rasterize": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dmx_preprocess_crop": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "dmx_postprocess_paste": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dmx_gemm_plan_override": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "dmx_pack_ups_phase_weights": (c_int, [_P, c_int, _P, c_int, c_int, _P]),
    "dmx_conv_ups2x_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "dmx_conv_ups2x": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_mse_loss_workspace_bytes": (c_size_t, []),
    "dmx_mse_loss": (c_int, [_P, _P, c_size_t, _P, _P, c_float, _P, c_size_t, _P]),
    "dmx_profile_begin": (c_int, []),
    "dmx_profile_end": (c_int, [POINTER(ctypes.c_double), c_int]),
    "dmx_profile_dump_path": (c_int, [c_char_p]),
    "dmx_profile_symbols": (c_size_t, [c_char_p, c_size_t]),
    "dmx_vae_create": (_P, [POINTER(VAEConfig)]),
    "dmx_vae_destroy": (None, [_P]),
    "dmx_vae_param_count": (c_int, [_P]),
    "dmx_vae_param_info": (c_int, [_P, c_int, POINTER(c_char_p), POINTER(c_int * 4)]),
    "dmx_vae_arena_bytes": (c_size_t, [_P]),
    "dmx_vae_bind_arena": (c_int, [_P, _P, c_size_t]),
    "dmx_vae_load_param": (c_int, [_P, c_char_p, _P, _P]),
    "dmx_vae_finalize": (c_int, [_P, _P]),
    "dmx_vae_workspace_bytes": (c_size_t, [_P, c_int, c_int, c_int, c_int]),
    "dmx_vae_encode": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dmx_vae_decode": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
}


def lib_path():
    return _LIB_PATH


def _load(path, want_elem):
    if not os.path.exists(path):
        raise RuntimeError(
            f"diffute_amd: HIP extension not built ({path} missing). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C diffute_amd/csrc`. "
            "There is no CPU fallback.")
    l = ctypes.CDLL(path)                  # RTLD_LOCAL: the two builds export the same names and must not see each other
    for name, (res, args) in _PROTOS.items():
        fn = getattr(l, name)              # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = l.dmx_element_type().decode()
    if got != want_elem:
        raise RuntimeError(f"diffute_amd: {path} computes in {got}, expected the {want_elem} build")
    if _exclusive is not None:
        l.dmx_set_exclusive_device(1 if _exclusive else 0)
    return l


def lib(elem="bf16"):
    """Load (once) and return the C-ABI library - the bf16 build, or with elem="fp16" the fp16 build of the same sources;
    raises if it has not been built."""
    global _lib, _lib_f16, _last_elem
    if elem == "fp16":
        if _lib_f16 is None:
            _lib_f16 = _load(_LIB_PATH_F16, "fp16")
        _last_elem = "fp16"
        return _lib_f16
    if elem != "bf16":
        raise ValueError(f"diffute_amd: no build for element type {elem!r}")
    if _lib is None:
        _lib = _load(_LIB_PATH, "bf16")
    _last_elem = "bf16"
    return _lib


def elem_of(dtype):
    """torch dtype a model was moved to -> the build that computes it: float16 -> "fp16"; float32 / bfloat16 -> "bf16"
    (fp32 requests are served by the bf16 build: master parameters stay fp32, compute is bf16 MFMA, fp32 accumulation)."""
    import torch
    return "fp16" if dtype == torch.float16 else "bf16"


def torch_elem(elem):
    import torch
    return torch.float16 if elem == "fp16" else torch.bfloat16


def exported_symbols():
    return sorted(_PROTOS.keys())


def check(rc, what="", l=None):
    """raise on a non-zero return code with the message of the library the call went through: `l`, or the build that lib() handed
    out last (every wrapper fetches its library right before the call)"""
    if rc != 0:
        if l is None:
            l = _lib_f16 if (_last_elem == "fp16" and _lib_f16 is not None) else lib()
        msg = l.dmx_last_error()
        raise RuntimeError(f"diffute_amd: {what} failed (code {rc}): {msg.decode() if msg else ''}")


def poll_device_error(l=None):
    """raise if a kernel of an EARLIER launch gave up on an in-kernel wait (include/diffute_hip.h dmx_device_error): no synchronisation, so
    call it after your own synchronize() to cover the launches in flight"""
    for lb in ([l] if l is not None else [x for x in (_lib, _lib_f16) if x is not None]):
        check(lb.dmx_device_error(), "device error poll", lb)


def exclusive_device(l):
    """current dmx_set_exclusive_device setting of library `l`"""
    return int(l.dmx_get_exclusive_device())


def set_exclusive_device(on):
    """Tell the library (every loaded build) whether its launches have the GPU to themselves (include/diffute_hip.h dmx_set_exclusive_device).
    The default, True, lets dmx_conv3x3_gn split the K range of a tile over co-resident blocks; pass False before running ANYTHING else on the
    same GPU next to the library's launches - a second model on another stream or thread, your own kernels or collectives on a side stream.
    denoise(micro_batches > 1) and set_gradient_sync(world > 1) switch it themselves.  A starved split launch never passes silently: it raises
    DMX_ERR_DEVICE (RuntimeError at the next call, or at diffute_amd.synchronize()).  Returns the previous setting."""
    global _exclusive
    old = True if _exclusive is None else _exclusive
    _exclusive = bool(on)                      # (a build loaded later starts from this setting: _load)
    for lb in (_lib, _lib_f16):
        if lb is not None:
            lb.dmx_set_exclusive_device(1 if on else 0)
    return old


def synchronize(device=None):
    """torch.cuda.synchronize() + the device -> host error poll: the public sync point.  Every launch of the library checks for an error raised by
    an EARLIER launch, so an error inside a sequence of calls surfaces by itself; only the LAST launches before the host reads results
    (the end of denoise(), of a training step, of vae.decode) have nobody after them - synchronise through this function (bench.py, the
    tests and __graft_entry__.smoke() do) and a kernel that gave up on an in-kernel wait raises here instead of handing back a wrong tensor."""
    import torch
    torch.cuda.synchronize(device)
    poll_device_error()


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("diffute_amd: tensors must live on the GPU (cuda / ROCm device); there is no CPU path")
