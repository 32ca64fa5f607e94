"""Helpers for the GPU parity tests: layout conversion and thin wrappers over the C ABI."""
import ctypes as C

import torch


def L():
    from lunaris_orion_amd import _lib
    return _lib


def to_nhwc_h(x_nchw: torch.Tensor) -> torch.Tensor:
    """fp32 NCHW (CPU) -> fp16 NHWC contiguous on the GPU."""
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(device="cuda", dtype=torch.float16)


def from_nhwc(x_nhwc: torch.Tensor) -> torch.Tensor:
    """fp16 NHWC (GPU) -> fp32 NCHW on the CPU."""
    return x_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def h16(x: torch.Tensor) -> torch.Tensor:
    """Round an fp32 CPU tensor through fp16 (what the kernels see)."""
    return x.to(torch.float16).to(torch.float32)


def sync():
    torch.cuda.synchronize()


def conv_forward(kind, x_nchw, w, bias, Cout, add_src=None, want_partial=False):
    lib = L()
    B, Cin, H, W = x_nchw.shape
    xin = to_nhwc_h(x_nchw)
    n = lib.lib.lo_packed_weight_elems_for(kind, B, H, W, Cin, Cout)
    assert n > 0, lib.lib.lo_last_error()
    wp = torch.empty(n, dtype=torch.float16, device="cuda")
    wd = w.contiguous().cuda()
    lib.check(lib.lib.lo_pack_weight_for(kind, B, H, W, Cin, Cout, wd.data_ptr(), wp.data_ptr(), lib.stream_ptr()), "pack")
    if kind in (0, 3, 6):
        Ho, Wo = H, W
    elif kind in (1, 5):
        Ho, Wo = H // 2, W // 2
    else:
        Ho, Wo = 2 * H, 2 * W
    out = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.float16, device="cuda")
    bd = bias.cuda() if bias is not None else None
    ad = to_nhwc_h(add_src) if add_src is not None else None
    part = torch.full((B * 4096 * 16,), float("nan"), dtype=torch.float32, device="cuda") if want_partial else None
    mt = C.c_int(0)
    lib.check(lib.lib.lo_conv_forward(kind, B, H, W, Cin, Cout, xin.data_ptr(), wp.data_ptr(), lib.ptr(bd), lib.ptr(ad),
                                      out.data_ptr(), lib.ptr(part), C.byref(mt), lib.stream_ptr()), "conv_forward")
    sync()
    if want_partial:
        return out, part[: B * mt.value * 16].view(B, mt.value, 8, 2), mt.value
    return out


def conv_wgrad(kind, x_nchw, dy_nchw, Cout, wshape, scale=1.0):
    lib = L()
    B, Cin, H, W = x_nchw.shape
    x = to_nhwc_h(x_nchw)
    dy = to_nhwc_h(dy_nchw)
    nb = lib.lib.lo_wgrad_slab_bytes_for(kind, B, H, W, Cin, Cout)
    slab = torch.empty(max(nb // 4, 1), dtype=torch.float32, device="cuda")
    grad = torch.full(wshape, float("nan"), dtype=torch.float32, device="cuda")
    lib.check(lib.lib.lo_conv_wgrad(kind, B, H, W, Cin, Cout, x.data_ptr(), dy.data_ptr(), slab.data_ptr(), grad.data_ptr(),
                                    scale, lib.stream_ptr()), "wgrad")
    sync()
    return grad.cpu()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def oracle_teacher_on_device(x, S, training, seed=None, p=0.0, device="cuda"):
    """The oracle's teacher forward (oracle/teacher_ref.py: the same Python functions, plain PyTorch fp32 ops) evaluated on DEVICE
    tensors for the sizes where the CPU evaluation alone cost the suite minutes (batch 64: 80 s; feature_dim 256 / 512: 40-50 s).
    The arithmetic then runs on ATen / MIOpen / hipBLASLt fp32 kernels -- still not this library's -- with TF32 off; the dropout
    bits come from the numpy restatement either way.  At the oracle's own sizes (batch 2) the CPU evaluation stays, and it is that
    one the reference's fixtures pin."""
    import torch
    from oracle import dropout_ref as D
    from oracle import teacher_ref as T
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    Sd = {k: v.to(device) for k, v in S.items()}
    masks = D.TeacherMasks(seed, p, x.shape[0], device=device) if (training and p > 0) else None
    with torch.no_grad():
        ref, stats = T.teacher_forward(x.to(device), Sd, training=training, masks=masks)
    torch.cuda.synchronize()
    return {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in ref.items()}, {k: v.cpu() for k, v in stats.items()}
