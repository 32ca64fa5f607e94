"""ctypes binding of liblunaris_hip.so (the C ABI declared in include/lunaris_hip.h).

The library is the product: if it is missing or fails to load there is NO fallback — importing
this module raises.  ``import torch`` must happen first so that the HIP runtime torch already
loaded (``libamdhip64.so.7``) is the one our kernels register with (streams and device pointers
are then shared with PyTorch).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime first; see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblunaris_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -m lunaris_orion_amd.build` (hipcc, gfx950). "
        "There is no CPU or PyTorch fallback for the lunaris_orion_amd hot path."
    )

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

vp, f32p, i32, sz, u64, flt = C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_uint64, C.c_float

# name -> (restype, argtypes); mirrors include/lunaris_hip.h one to one
SIGNATURES = {
    "lo_last_error": (C.c_char_p, []),
    "lo_version": (i32, []),
    "lo_prof_enable": (None, [i32]),
    "lo_prof_count": (i32, []),
    "lo_prof_get": (i32, [i32, C.c_char_p, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lo_packed_weight_elems_for": (sz, [i32] * 6),
    "lo_pack_weight_for": (i32, [i32] * 6 + [f32p, vp, vp]),
    "lo_conv_forward": (i32, [i32] * 6 + [vp, vp, f32p, vp, vp, f32p, C.POINTER(C.c_int), vp]),
    "lo_quantize_act_f8": (i32, [vp, vp, sz, vp]),
    "lo_pack_weight_f8_for": (i32, [i32] * 6 + [vp, vp, f32p, vp]),
    "lo_conv_forward_f8": (i32, [i32] * 6 + [vp, vp, f32p, f32p, vp, vp, f32p, C.POINTER(C.c_int), vp]),
    "lo_conv3x3_fused_tap_forward": (i32, [i32] * 6 + [vp, vp, f32p, f32p, i32, vp, f32p, vp]),
    "lo_linear_splitk": (i32, [i32, i32, i32, vp, vp, f32p, f32p, i32, f32p, vp, vp]),
    "lo_wgrad_slab_bytes_for": (sz, [i32] * 6),
    "lo_conv_wgrad": (i32, [i32] * 6 + [vp, vp, f32p, f32p, flt, vp]),
    "lo_gn_mish_forward": (i32, [vp, f32p, i32, f32p, f32p, vp, vp, f32p, i32, i32, i32, i32, vp]),
    "lo_gn_nchunk_for": (i32, [i32, i32]),
    "lo_gn_mish_backward": (i32, [vp, vp, vp, f32p, f32p, f32p, vp, vp, f32p, f32p, f32p, f32p, f32p, i32, i32, i32, i32, flt, vp]),
    "lo_first_conv_forward": (i32, [f32p, f32p, f32p, vp, f32p, i32, vp]),
    "lo_first_conv_wgrad_op": (i32, [f32p, vp, f32p, f32p, i32, flt, vp]),
    "lo_final_conv_forward": (i32, [vp, f32p, f32p, f32p, f32p, f32p, i32, vp]),
    "lo_final_conv_backward": (i32, [vp, f32p, f32p, f32p, f32p, f32p, flt, vp, f32p, f32p, f32p, i32, flt, vp]),
    "lo_decode_sprites_u8": (i32, [vp, f32p, i32, vp]),
    "lo_selfattn2d_forward": (i32, [f32p] * 12 + [i32, i32, i32, vp]),
    "lo_selfattn2d_backward_scratch_elems": (C.c_size_t, [i32, i32, i32]),
    "lo_selfattn2d_backward": (i32, [f32p] * 18 + [i32, i32, i32, vp]),
    "lo_clip_adamw_step": (i32, [f32p, f32p, f32p, f32p, sz, flt, flt, flt, flt, flt, flt, i32, f32p, vp]),
    "lo_teacher_create": (i32, [i32, i32, i32, i32, C.POINTER(C.c_void_p)]),
    "lo_teacher_create_ex": (i32, [i32, i32, i32, i32, C.c_uint, C.POINTER(C.c_void_p)]),
    "lo_teacher_destroy": (None, [vp]),
    "lo_teacher_num_tensors": (i32, [vp]),
    "lo_teacher_tensor_name": (C.c_char_p, [vp, i32]),
    "lo_teacher_tensor_numel": (sz, [vp, i32]),
    "lo_teacher_tensor_offset": (C.c_longlong, [vp, i32]),
    "lo_teacher_flat_elems": (sz, [vp]),
    "lo_teacher_workspace_bytes": (sz, [vp]),
    "lo_teacher_pack": (i32, [vp, f32p, vp, vp]),
    "lo_teacher_forward": (i32, [vp, f32p, f32p, vp, i32, flt, u64, f32p, f32p, f32p, f32p, f32p, vp]),
    "lo_teacher_last_path": (i32, [vp]),
    "lo_dropout_mask": (i32, [u64, i32, flt, sz, vp, vp]),
    "lo_teacher_grad_range": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lo_teacher_heads_backward": (i32, [vp, f32p, vp, f32p, flt, f32p, f32p, vp]),
    "lo_teacher_heads_saved": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lo_teacher_full_backward_bytes": (C.c_size_t, [vp]),
    "lo_teacher_full_backward_ex": (i32, [vp, f32p, f32p, vp, vp, f32p, f32p, f32p, f32p, f32p, f32p, flt, u64, flt, f32p, f32p, vp]),
    "lo_teacher_forward_keep": (i32, [vp, f32p, f32p, vp, vp, flt, u64, f32p, f32p, f32p, f32p, f32p, vp]),
    "lo_teacher_full_backward": (i32, [vp, f32p, f32p, vp, vp, f32p, flt, flt, f32p, f32p, vp]),
    "lo_teacher_clip_adamw_full": (i32, [vp, f32p, f32p, f32p, f32p, flt, flt, flt, flt, flt, flt, i32, f32p, vp]),
    "lo_teacher_full_param_count": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lo_teacher_heads_backward_ex": (i32, [vp, f32p, f32p, f32p, f32p, f32p, f32p, f32p, flt, u64, f32p, f32p, vp]),
    "lo_hybrid_reward": (i32, [f32p, f32p, i32, flt, flt, flt, flt, flt, f32p, f32p, f32p, vp]),
    "lo_vae_create": (i32, [i32, i32, C.POINTER(C.c_void_p)]),
    "lo_vae_create_ex": (i32, [i32, i32, C.c_uint, C.POINTER(C.c_void_p)]),
    "lo_vae_set_gradnorm_scratch": (i32, [vp, f32p]),
    "lo_gradnorm_early_range": (i32, [f32p, sz, sz, f32p, vp]),
    "lo_vae_set_async_handover": (i32, [vp, i32]),
    "lo_vae_wait_handover": (i32, [vp, vp]),
    "lo_vae_optimizer_step": (i32, [vp, f32p, f32p, f32p, f32p, vp, flt, flt, flt, flt, flt, flt, i32, f32p, i32, vp]),
    "lo_vae_join": (i32, [vp, vp]),
    "lo_vae_set_linear_factored": (i32, [vp, i32]),
    "lo_vae_linear_factored": (i32, [vp]),
    "lo_vae_factor_block": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lo_vae_materialize_gathered_linear_grads": (i32, [vp, vp, i32, f32p, vp]),
    "lo_vae_materialize_linear_grads": (i32, [vp, vp, f32p, vp]),
    "lo_vae_gradnorm_presummed": (i32, [vp]),
    "lo_clip_adamw_step_presummed": (i32, [f32p, f32p, f32p, f32p, sz, sz, flt, flt, flt, flt, flt, flt, i32, f32p, vp]),
    "lo_vae_destroy": (None, [vp]),
    "lo_vae_num_params": (i32, [vp]),
    "lo_vae_param_offset": (sz, [vp, i32]),
    "lo_vae_param_numel": (sz, [vp, i32]),
    "lo_vae_flat_elems": (sz, [vp]),
    "lo_vae_workspace_bytes": (sz, [vp]),
    "lo_vae_pack": (i32, [vp, f32p, vp, vp]),
    "lo_vae_forward": (i32, [vp, f32p, f32p, u64, f32p, vp, f32p, f32p, f32p, f32p, vp]),
    "lo_vae_decode": (i32, [vp, f32p, f32p, vp, f32p, vp]),
    "lo_vae_encode": (i32, [vp, f32p, f32p, vp, f32p, f32p, f32p, f32p, f32p, vp]),
    "lo_vae_decode_skips": (i32, [vp, f32p, i32, f32p, f32p, f32p, f32p, vp, f32p, vp]),
    "lo_vae_decoder_backward": (i32, [vp, f32p, vp, f32p, f32p, flt, f32p, f32p, f32p, f32p, f32p, vp]),
    "lo_vae_encoder_backward": (i32, [vp, f32p, f32p, vp, f32p, f32p, f32p, f32p, f32p, flt, f32p, vp]),
    "lo_vae_loss": (i32, [vp, vp, flt, flt, flt, f32p, flt, flt, f32p, vp]),
    "lo_vae_backward": (i32, [vp, f32p, f32p, vp, f32p, f32p, i32, f32p, f32p, f32p, flt, f32p, vp]),
    "lo_vae_backward_phase": (i32, [vp, i32, f32p, f32p, vp, f32p, f32p, i32, f32p, f32p, f32p, flt, f32p, vp]),
    "lo_vae_linear_grad_range": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lo_vae_phase1_grad_range": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lo_vae_debug_tensor": (i32, [vp, i32, i32, i32, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "lo_dp_pack_f16": (i32, [f32p, vp, sz, flt, vp]),
    "lo_dp_unpack_f16": (i32, [vp, f32p, sz, flt, vp]),
    "lo_dp_unpack_f16_sumsq": (i32, [vp, f32p, sz, flt, f32p, vp]),
    "lo_dp_sum_shares": (i32, [vp, vp, i32, sz, i32, vp]),
    "lo_vae_sync_fail_word": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "lo_grad_scale_pick": (i32, [f32p, sz, f32p, sz, f32p, sz, f32p, sz, f32p, sz, f32p, vp]),
    "lo_scale_copy_dev": (i32, [f32p, f32p, sz, f32p, vp]),
    "lo_grad_unscale_dev": (i32, [f32p, sz, f32p, vp, vp]),
    "lo_vae_fp8_layers": (i32, [vp, C.POINTER(C.c_int)]),
    "lo_vae_stage4_grad_range": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args


class LunarisHipError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.lo_last_error()
        raise LunarisHipError(f"{what or 'liblunaris_hip'} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int | None:
    """Device pointer of a tensor (None passes NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr() -> int:
    """The HIP stream PyTorch is currently enqueuing on (hipStream_t as an integer)."""
    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise LunarisHipError(
            "lunaris_orion_amd needs an AMD GPU (gfx950) visible to PyTorch-ROCm; there is no CPU fallback. "
            "The CPU oracle lives in oracle/ and is test infrastructure only."
        )
