"""LunarisCoreVAE on MI355X: same ``nn.Module`` surface as the reference, arithmetic in liblunaris_hip.so.

Drop-in contract (reference: /root/reference/lunar_generate.py):
  * ``LunarisCoreVAE(latent_dim=256)``; ``forward(x) -> (reconstruction, mu, logvar)`` (:263-276);
    ``.encoder`` / ``.decoder`` sub-module trees with the reference's attribute paths, so the 72
    ``state_dict`` keys and shapes are identical (``encoder.down1.0.weight`` ...
    ``decoder.final_conv.bias``) and checkpoints interchange; default PyTorch initialisers, created in
    the reference's order (same weights for the same seed); ``reparameterize`` (:248-261); ``sample(n)``
    (:278-291).
  * parameters are fp32 ``nn.Parameter``s.  They are views of ONE flat fp32 buffer (so clip + AdamW is
    a single fused kernel and a data-parallel gradient exchange is one contiguous buffer); any
    optimizer that works on ``model.parameters()`` still works.

``LunarisCoreVAE.forward`` hands the flat parameter buffer to the native step executor
(``lo_vae_forward`` / ``lo_vae_backward``: one call each), which enqueues every kernel of the pass on
PyTorch's current HIP stream.  ``model.encoder(x) -> (mu, logvar, skips)`` and ``model.decoder(z, skips)``
(lunar_generate.py:127-153, 194-229, called in turn at :273-275 and with ``skips=[]`` at :290) are
callable on their own as well: each is one native call (``lo_vae_encode`` / ``lo_vae_decode_skips``)
with its own autograd node (``lo_vae_encoder_backward`` / ``lo_vae_decoder_backward``); the leaf
modules below them (``Conv2d``, ``GroupNorm`` ...) are parameter containers.  Parameter updates made
by a foreign optimizer (``torch.optim.AdamW(model.parameters())``) are noticed through the
parameters' version counters: the fp16 operand copies are re-packed before the next forward without
any call from the user.  There is no PyTorch/CPU fallback: on a machine without the GPU library the
import of ``_lib`` fails.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib

_LCG_A, _LCG_C, _M64 = 6364136223846793005, 1442695040888963407, 0xFFFFFFFFFFFFFFFF


def lcg_advance(seed: int, n: int) -> int:
    """`n` steps of the 64-bit LCG both counter-RNG streams advance by per call (noise of the reparameterisation, teacher dropout
    masks), in O(log n): how a resumed run re-enters its streams at the position the checkpoint recorded."""
    a, c, ra, rc = _LCG_A, _LCG_C, 1, 0          # (ra, rc) = the affine map composed so far; (a, c) = the map for 2**k steps
    while n > 0:
        if n & 1:
            ra, rc = (ra * a) & _M64, (rc * a + c) & _M64
        a, c = (a * a) & _M64, (c * a + c) & _M64
        n >>= 1
    return (seed * ra + rc) & _M64


_ENC = ((3, 64), (64, 128), (128, 256), (256, 512))
_DEC = ((512, 256), (256, 128), (128, 64), (64, 32))


def _conv_gn_mish(cin: int, cout: int, **conv_kw) -> List[nn.Module]:
    return [nn.Conv2d(cin, cout, **conv_kw), nn.GroupNorm(num_groups=8, num_channels=cout), nn.Mish()]


class ResBlock(nn.Module):
    """Parameter container for the reference ResBlock (lunar_generate.py:28-53)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        if in_channels != out_channels:
            raise NotImplementedError("the VAE only uses identity-shortcut ResBlocks (lunar_generate.py:98,105,112,119)")
        self.conv1 = nn.Sequential(*_conv_gn_mish(in_channels, out_channels, kernel_size=3, padding=1))
        self.conv2 = nn.Sequential(*_conv_gn_mish(out_channels, out_channels, kernel_size=3, padding=1))
        self.shortcut = nn.Identity()


class SelfAttention2d(nn.Module):
    """Spatial self-attention block of the reference (lunar_generate.py:56-78), which defines it but never instantiates
    it.  Same parameters (`query_conv`, `key_conv`, `value_conv`, `gamma`); forward is one fused HIP pass
    (`lo_selfattn2d_forward` / `lo_selfattn2d_backward`: no N x N tensor is materialised in either direction)."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.query_conv = nn.Conv2d(in_channels, in_channels // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_channels, in_channels // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu()
        if not x.is_cuda:
            raise _lib.LunarisHipError("SelfAttention2d needs CUDA (ROCm) tensors; there is no CPU path")
        return _SelfAttnFunction.apply(x, self.query_conv.weight, self.query_conv.bias, self.key_conv.weight, self.key_conv.bias,
                                       self.value_conv.weight, self.value_conv.bias, self.gamma)


class _SelfAttnFunction(torch.autograd.Function):
    """Forward / backward of SelfAttention2d as two native calls (`lo_selfattn2d_forward` / `_backward`)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma):
        B, Cc, H, W = x.shape
        N = H * W
        xd = x.detach().contiguous().float()
        P = [t.detach().contiguous().float() for t in (wq, bq, wk, bk, wv, bv, gamma)]
        q = torch.empty(B, Cc // 8, N, dtype=torch.float32, device=x.device)
        k = torch.empty_like(q)
        v = torch.empty(B, Cc, N, dtype=torch.float32, device=x.device)
        out = torch.empty_like(xd)
        _lib.check(_lib.lib.lo_selfattn2d_forward(xd.data_ptr(), *[t.data_ptr() for t in P], q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                  out.data_ptr(), B, Cc, N, _lib.stream_ptr()), "lo_selfattn2d_forward")
        ctx.save_for_backward(xd, P[0], P[2], P[4], P[6], q, k, v)
        ctx.shapes = (wq.shape, bq.shape, wk.shape, bk.shape, wv.shape, bv.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        xd, wq, wk, wv, gamma, q, k, v = ctx.saved_tensors
        B, Cc, H, W = xd.shape
        N = H * W
        dy = dy.detach().contiguous().float()
        dev = xd.device
        scratch = torch.empty(_lib.lib.lo_selfattn2d_backward_scratch_elems(B, Cc, N), dtype=torch.float32, device=dev)
        dx = torch.empty_like(xd)
        dwq, dwk, dwv = torch.empty_like(wq), torch.empty_like(wk), torch.empty_like(wv)
        dbq = torch.empty(Cc // 8, dtype=torch.float32, device=dev)
        dbk = torch.empty_like(dbq)
        dbv = torch.empty(Cc, dtype=torch.float32, device=dev)
        dg = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib.lo_selfattn2d_backward(xd.data_ptr(), wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), gamma.data_ptr(), q.data_ptr(),
                                                   k.data_ptr(), v.data_ptr(), dy.data_ptr(), scratch.data_ptr(), dx.data_ptr(),
                                                   dwq.data_ptr(), dbq.data_ptr(), dwk.data_ptr(), dbk.data_ptr(), dwv.data_ptr(),
                                                   dbv.data_ptr(), dg.data_ptr(), B, Cc, N, _lib.stream_ptr()), "lo_selfattn2d_backward")
        s = ctx.shapes
        return dx, dwq.view(s[0]), dbq.view(s[1]), dwk.view(s[2]), dbk.view(s[3]), dwv.view(s[4]), dbv.view(s[5]), dg


def _owner_of(mod: nn.Module) -> "LunarisCoreVAE":
    ref = getattr(mod, "_owner", None)
    owner = ref() if ref is not None else None
    if owner is None:
        raise _lib.LunarisHipError(f"{type(mod).__name__} runs through the LunarisCoreVAE that owns it (flat parameter buffer, native "
                                   "plan); construct it as part of a LunarisCoreVAE")
    return owner


class _OwnedByVAE:
    """Pickling / copying support for the sub-modules' weak back-reference to their LunarisCoreVAE (re-bound by the owner)."""

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop("_owner", None)
        return d


class Encoder(_OwnedByVAE, nn.Module):
    """4 x (Conv k3 s2 -> GN -> Mish -> ResBlock), fc_mu, fc_logvar (lunar_generate.py:84-125).  ``forward(x) -> (mu, logvar,
    skips)`` (:127-153) is one native call (``lo_vae_encode``); with gradients enabled it is an autograd node whose backward is
    ``lo_vae_encoder_backward``."""

    def __init__(self, latent_dim: int = 256):
        super().__init__()
        for i, (cin, cout) in enumerate(_ENC, start=1):
            stage = _conv_gn_mish(cin, cout, kernel_size=3, stride=2, padding=1) + [ResBlock(cout, cout)]
            setattr(self, f"down{i}", nn.Sequential(*stage))
        self.flatten = nn.Flatten()
        self.fc_mu = nn.Linear(512 * 8 * 8, latent_dim)
        self.fc_logvar = nn.Linear(512 * 8 * 8, latent_dim)

    def forward(self, x: torch.Tensor):
        vae = _owner_of(self)
        vae._ensure_flat()
        x = x.detach().contiguous().float()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            mu, logvar, s0, s1, s2 = _EncoderFunction.apply(vae, x, *self.parameters())
        else:
            mu, logvar, s0, s1, s2, _ = vae._native_encode(x)
        return mu, logvar, [s0, s1, s2]


class Decoder(_OwnedByVAE, nn.Module):
    """fc, 4 x (ConvTranspose k4 s2 -> GN -> Mish), final_conv (lunar_generate.py:155-192).  ``forward(z, skips)`` (:194-229, with
    the reference's ``len(skips) >= k`` guards; ``skips=[]`` is the sampling call of :290) is one native call
    (``lo_vae_decode_skips``); with gradients enabled it is an autograd node whose backward is ``lo_vae_decoder_backward``."""

    def __init__(self, latent_dim: int = 256):
        super().__init__()
        self.fc = nn.Linear(latent_dim, 512 * 8 * 8)
        for i, (cin, cout) in enumerate(_DEC, start=1):
            setattr(self, f"up{i}", nn.Sequential(nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1),
                                                  nn.GroupNorm(8, cout), nn.Mish()))
        self.final_conv = nn.Conv2d(32, 3, kernel_size=3, padding=1)

    def forward(self, z: torch.Tensor, skips: Sequence[torch.Tensor]):
        vae = _owner_of(self)
        vae._ensure_flat()
        skips = list(skips)[:3]
        needs_grad = torch.is_grad_enabled() and (z.requires_grad or any(t.requires_grad for t in skips)
                                                  or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            return _DecoderFunction.apply(vae, len(skips), z, *skips, *self.parameters())
        return vae._native_decode(z, skips)[0]


def _normalise_upstream(grads):
    """Bring the upstream gradients of an autograd node into the range the fp16 backward is built for, whatever loss scale the
    caller's loop has multiplied them by (train_hybrid.py:899-904: ``scaler.scale(vae_loss).backward()`` with GradScaler's 65 536):
    ``r = 2**k`` with ``max|g| * r`` in [2, 4) is picked ON THE DEVICE (``lo_grad_scale_pick``; no host synchronisation), the
    gradients are multiplied by it, and ``_denormalise`` multiplies what the backward produced by ``1 / r`` — both exact in fp32.
    Returns (scratch, scaled gradients); scratch[0] = r, scratch[1] = 1 / r."""
    live = [g for g in grads if g is not None]
    dev = live[0].device
    scr = torch.empty(264, dtype=torch.float32, device=dev)
    st = _lib.stream_ptr()
    args = []
    for k in range(5):
        g = grads[k] if k < len(grads) else None
        args += [_lib.ptr(g), g.numel() if g is not None else 0]
    _lib.check(_lib.lib.lo_grad_scale_pick(*args, scr.data_ptr(), st), "lo_grad_scale_pick")
    out = []
    for g in grads:
        if g is None:
            out.append(None)
            continue
        d = torch.empty_like(g)
        _lib.check(_lib.lib.lo_scale_copy_dev(g.data_ptr(), d.data_ptr(), g.numel(), scr.data_ptr(), st), "lo_scale_copy_dev")
        out.append(d)
    return scr, out


def _denormalise(t: Optional[torch.Tensor], scr: torch.Tensor, eng: "_Engine") -> None:
    """t *= 1 / r (see ``_normalise_upstream``); a set rendezvous-failure word of the engine turns the result into NaN."""
    if t is not None:
        _lib.check(_lib.lib.lo_grad_unscale_dev(t.data_ptr(), t.numel(), scr.data_ptr() + 4, eng.sync_fail.data_ptr(), _lib.stream_ptr()),
                   "lo_grad_unscale_dev")


def _check_generation(eng: "_Engine", what: str, want) -> None:
    """The activations of a forward live in the engine's ONE workspace.  The reference's modules allow two forwards before a backward
    (or an eval forward of the same batch size inside a step); here the second forward overwrites what the first one's backward needs,
    and that backward must say so instead of returning gradients of the wrong activations."""
    have = (eng.gen_enc, eng.gen_dec)
    bad = [n for n, h, w in zip(("encoder", "decoder"), have, want) if w is not None and h != w]
    if bad:
        raise _lib.LunarisHipError(
            f"{what}: another forward of this module at the same batch size has overwritten the {' and '.join(bad)} activations this "
            "backward needs (one workspace per batch size). Run backward() before the next forward of that batch size, or run the "
            "other forward under a different batch size / on a copy of the module.")


class _EncoderFunction(torch.autograd.Function):
    """Encoder.forward / its backward as one native call each."""

    @staticmethod
    def forward(ctx, vae, x, *params):
        mu, logvar, s0, s1, s2, eng = vae._native_encode(x)
        ctx.vae, ctx.eng, ctx.nparam, ctx.gen = vae, eng, len(params), (eng.gen_enc, None)
        ctx.save_for_backward(x)
        return mu, logvar, s0, s1, s2

    @staticmethod
    def backward(ctx, g_mu, g_lv, g0, g1, g2):
        vae, eng = ctx.vae, ctx.eng
        _check_generation(eng, "Encoder backward", ctx.gen)
        (x,) = ctx.saved_tensors
        cont = lambda t: None if t is None else t.contiguous().float()
        ups = [cont(t) for t in (g_mu, g_lv, g0, g1, g2)]
        flat_g = torch.zeros_like(vae._flat)
        if all(t is None for t in ups):
            return (None, None) + tuple(flat_g[o:o + n].view(shape) for (o, n, shape) in vae._layout[:ctx.nparam])
        scr, (g_mu, g_lv, g0, g1, g2) = _normalise_upstream(ups)
        _lib.check(_lib.lib.lo_vae_encoder_backward(eng.handle, x.data_ptr(), vae._flat.data_ptr(), eng.ws.data_ptr(), _lib.ptr(g_mu),
                                                    _lib.ptr(g_lv), _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(g2), 1.0,
                                                    flat_g.data_ptr(), _lib.stream_ptr()), "lo_vae_encoder_backward")
        _denormalise(flat_g, scr, eng)
        grads = tuple(flat_g[o:o + n].view(shape) for (o, n, shape) in vae._layout[:ctx.nparam])
        return (None, None) + grads


class _DecoderFunction(torch.autograd.Function):
    """Decoder.forward / its backward as one native call each."""

    @staticmethod
    def forward(ctx, vae, n_skips, z, *rest):
        skips, params = rest[:n_skips], rest[n_skips:]
        recon, eng = vae._native_decode(z, skips)
        ctx.vae, ctx.eng, ctx.n_skips, ctx.nparam, ctx.gen = vae, eng, n_skips, len(params), (None, eng.gen_dec)
        ctx.save_for_backward(recon)
        return recon

    @staticmethod
    def backward(ctx, g_recon):
        vae, eng, ns = ctx.vae, ctx.eng, ctx.n_skips
        _check_generation(eng, "Decoder backward", ctx.gen)
        (recon,) = ctx.saved_tensors
        scr, (g_recon,) = _normalise_upstream([g_recon.contiguous().float()])
        B, dev = recon.shape[0], recon.device
        flat_g = torch.zeros_like(vae._flat)
        dz = torch.empty(B, vae.latent_dim, dtype=torch.float32, device=dev)
        dsk = [torch.empty(B, 64 << k, 64 >> k, 64 >> k, dtype=torch.float32, device=dev) if k < ns else None for k in range(3)]
        _lib.check(_lib.lib.lo_vae_decoder_backward(eng.handle, vae._flat.data_ptr(), eng.ws.data_ptr(), recon.data_ptr(), g_recon.data_ptr(),
                                                    1.0, dz.data_ptr(), _lib.ptr(dsk[0]), _lib.ptr(dsk[1]), _lib.ptr(dsk[2]),
                                                    flat_g.data_ptr(), _lib.stream_ptr()), "lo_vae_decoder_backward")
        for t in (flat_g, dz, *dsk):
            _denormalise(t, scr, eng)
        first = len(vae._layout) - ctx.nparam                    # the decoder's parameters are the tail of the state_dict order
        grads = tuple(flat_g[o:o + n].view(shape) for (o, n, shape) in vae._layout[first:])
        return (None, None, dz) + tuple(dsk[:ns]) + grads


class _Engine:
    """Native plan + workspace for one (batch, latent_dim, device)."""

    def __init__(self, batch: int, latent_dim: int, device: torch.device, flags: int = 0):
        self.handle = C.c_void_p()
        _lib.check(_lib.lib.lo_vae_create_ex(batch, latent_dim, flags, C.byref(self.handle)), "lo_vae_create_ex")
        self.batch, self.latent_dim, self.device = batch, latent_dim, device
        self.ws = torch.empty(_lib.lib.lo_vae_workspace_bytes(self.handle), dtype=torch.uint8, device=device)
        self.packed_version = None      # (explicit version, sum of the parameters' version counters) of the last pack
        self.fac_mode = 0               # lo_vae_set_linear_factored: 0 off, 1 single process, 2 data parallel (set by the stepper)
        self.gen_enc = self.gen_dec = 0 # bumped by every call that overwrites the encoder / decoder activations in the workspace
        off, nf = C.c_size_t(), C.c_int()
        _lib.check(_lib.lib.lo_vae_sync_fail_word(self.handle, C.byref(off), C.byref(nf)), "lo_vae_sync_fail_word")
        self.fused_gn_layers = nf.value
        self.sync_fail = self.ws[off.value:off.value + 4].view(torch.int32)     # set by a fused-GroupNorm workgroup whose wait ran out
        n8 = C.c_int()
        _lib.check(_lib.lib.lo_vae_fp8_layers(self.handle, C.byref(n8)), "lo_vae_fp8_layers")
        self.fp8_layers = n8.value      # forward conv layers on e4m3 operands at THIS batch size (0 unless mfma_precision="fp8")

    def __del__(self):
        try:
            if self.handle:
                _lib.lib.lo_vae_destroy(self.handle)
        except Exception:
            pass


class _VAEFunction(torch.autograd.Function):
    """Autograd bridge: forward/backward of the whole VAE are one native call each."""

    @staticmethod
    def forward(ctx, model, x, eps, *params):
        recon, mu, logvar, eng = model._native_forward(x, eps, target=None)
        ctx.model, ctx.eng, ctx.gen = model, eng, (eng.gen_enc, eng.gen_dec)
        ctx.save_for_backward(x, recon)
        return recon, mu, logvar

    @staticmethod
    def backward(ctx, g_recon, g_mu, g_logvar):
        model, eng = ctx.model, ctx.eng
        _check_generation(eng, "LunarisCoreVAE backward", ctx.gen)
        x, recon = ctx.saved_tensors
        flat_g = torch.empty_like(model._flat)
        cont = lambda t: None if t is None else t.contiguous().float()
        ups = [cont(g_recon), cont(g_mu), cont(g_logvar)]
        if all(t is None for t in ups):
            flat_g.zero_()
            return (None, None, None) + tuple(flat_g[o:o + n].view(shape) for (o, n, shape) in model._layout)
        # the upstream gradients may carry a foreign loss scale (torch.amp.GradScaler): normalised on the device, see _normalise_upstream
        scr, (g_recon, g_mu, g_logvar) = _normalise_upstream(ups)
        _lib.check(_lib.lib.lo_vae_backward(eng.handle, x.data_ptr(), model._flat.data_ptr(), eng.ws.data_ptr(), recon.data_ptr(),
                                            None, 0, _lib.ptr(g_recon), _lib.ptr(g_mu), _lib.ptr(g_logvar),
                                            1.0, flat_g.data_ptr(), _lib.stream_ptr()), "lo_vae_backward")
        _denormalise(flat_g, scr, eng)
        grads = tuple(flat_g[o:o + n].view(shape) for (o, n, shape) in model._layout)
        return (None, None, None) + grads


class LunarisCoreVAE(nn.Module):
    """Variational auto-encoder for 128x128 pixel art; see the module docstring for the drop-in contract."""

    #: operand formats of the forward convolutions: "fp16" (default; the parity-tested path) or "fp8" (BASELINE config 5:
    #: OCP e4m3 operands where the input channel count is a multiple of 128, fp16 backward; see include/lunaris_hip.h)
    MFMA_PRECISIONS = {"fp16": 0, "fp8": 1}

    def __init__(self, latent_dim: int = 256, mfma_precision: str = "fp16"):
        super().__init__()
        if mfma_precision not in self.MFMA_PRECISIONS:
            raise ValueError(f"mfma_precision must be one of {sorted(self.MFMA_PRECISIONS)}, got {mfma_precision!r}")
        self.mfma_precision = mfma_precision
        self.latent_dim = latent_dim
        self.encoder = Encoder(latent_dim=latent_dim)
        self.decoder = Decoder(latent_dim=latent_dim)
        for sub in (self.encoder, self.decoder):                 # not a registered attribute: no module cycle, nothing in state_dict
            object.__setattr__(sub, "_owner", weakref.ref(self))
        #: loss scale of the FUSED step (trainer.VAEStepper, which also runs GradScaler's policy on it).  The autograd path does not
        #: use it: there the upstream gradients are normalised on the device (`_normalise_upstream`), so a loop that brings its own
        #: torch.amp.GradScaler (train_hybrid.py:289-297, 899-923) needs no setting here
        self.loss_scale = 65536.0
        self._flat: Optional[torch.Tensor] = None
        self._layout: List[Tuple[int, int, torch.Size]] = []
        self._engines: Dict[Tuple[int, str], _Engine] = {}
        self._weights_version = 0      # bumped whenever the fp32 parameters may have changed
        self._seed: Optional[int] = None   # counter-RNG stream of the reparameterisation noise; derived at the first forward
        self.noise_calls = 0               # forwards drawn from the stream so far (checkpointed: a resumed run continues the stream)
        #: parity runs: N(0,1) noise [B, latent_dim] consumed by the NEXT forward that is not given `eps` explicitly (a caller whose
        #: step calls `vae(images)` like train_hybrid.py:850 cannot pass it; the fixture generator patches `randn_like` the same way)
        self.next_eps: Optional[torch.Tensor] = None

    # ---- flat parameter buffer ------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._flat = None              # parameters were re-created (e.g. .to(device)): re-flatten lazily
        self._engines.clear()
        return out

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self.mark_weights_changed()
        return out

    def mark_weights_changed(self) -> None:
        """The fp16 operand copies are re-packed before the next forward.  Needed only after writing the flat buffer through a raw
        pointer (the native optimizer step does that and calls this itself): in-place updates through the parameters — any
        ``torch.optim`` optimizer, ``p.data.add_(...)``, ``p.copy_(...)`` under ``no_grad`` — bump the parameters' version counters,
        which ``_engine`` checks."""
        self._weights_version += 1

    def _params_version(self) -> int:
        """Sum of the 72 parameters' autograd version counters: moves on every in-place update of any of them (the views of the
        flat buffer keep their own counters, and `torch.optim.AdamW.step()` bumps them; tests/test_module_boundary_gpu.py)."""
        return sum(p._version for p in self.parameters())

    def _current_version(self):
        """What `_Engine.packed_version` is compared with: (explicit counter, sum of the parameters' version counters)."""
        return (self._weights_version, self._params_version())

    # copies and pickles carry the parameters, not the native plans / flat buffer (rebuilt lazily), and get their own back-references
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_flat"], d["_layout"], d["_engines"] = None, [], {}
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        for sub in (self.encoder, self.decoder):
            object.__setattr__(sub, "_owner", weakref.ref(self))

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__({k: copy.deepcopy(v, memo) for k, v in self.__getstate__().items()})
        return new

    def _ensure_flat(self) -> None:
        params = list(self.parameters())
        dev = params[0].device
        if self._flat is not None and self._flat.device == dev:
            ok = all(p.data_ptr() == self._flat.data_ptr() + 4 * o for p, (o, _n, _s) in zip(params, self._layout))
            if ok:
                return
        _lib.require_gpu()
        if dev.type != "cuda":
            raise _lib.LunarisHipError("LunarisCoreVAE parameters must be on the GPU (model.to('cuda')); there is no CPU path")
        probe = C.c_void_p()
        _lib.check(_lib.lib.lo_vae_create(1, self.latent_dim, C.byref(probe)), "lo_vae_create")
        try:
            assert _lib.lib.lo_vae_num_params(probe) == len(params), "parameter table mismatch"
            layout = []
            for i, p in enumerate(params):
                n = _lib.lib.lo_vae_param_numel(probe, i)
                assert n == p.numel(), f"parameter {i}: numel {p.numel()} != {n}"
                layout.append((_lib.lib.lo_vae_param_offset(probe, i), n, p.shape))
            total = _lib.lib.lo_vae_flat_elems(probe)
        finally:
            _lib.lib.lo_vae_destroy(probe)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, (o, n, shape) in zip(params, layout):
                flat[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = flat[o:o + n].view(shape)
        self._flat, self._layout = flat, layout
        self._weights_version += 1
        self._engines.clear()

    def flat_parameters(self) -> torch.Tensor:
        """The single fp32 buffer all 72 parameters are views of (padding elements are zero)."""
        self._ensure_flat()
        return self._flat

    def _engine(self, batch: int) -> _Engine:
        self._ensure_flat()
        key = (batch, str(self._flat.device))
        eng = self._engines.get(key)
        if eng is None:
            eng = _Engine(batch, self.latent_dim, self._flat.device, self.MFMA_PRECISIONS[self.mfma_precision])
            self._engines[key] = eng
        version = self._current_version()
        if eng.packed_version != version:
            _lib.check(_lib.lib.lo_vae_pack(eng.handle, self._flat.data_ptr(), eng.ws.data_ptr(), _lib.stream_ptr()), "lo_vae_pack")
            eng.packed_version = version
        return eng

    # ---- forward --------------------------------------------------------------------------
    def _native_forward(self, x: torch.Tensor, eps: Optional[torch.Tensor], target: Optional[torch.Tensor]):
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 128, 128):
            raise ValueError(f"expected input of shape [B, 3, 128, 128], got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        B = x.shape[0]
        eng = self._engine(B)
        dev = x.device
        recon = torch.empty(B, 3, 128, 128, dtype=torch.float32, device=dev)
        mu = torch.empty(B, self.latent_dim, dtype=torch.float32, device=dev)
        logvar = torch.empty(B, self.latent_dim, dtype=torch.float32, device=dev)
        if eps is not None:
            eps = eps.detach().contiguous().float()
            if tuple(eps.shape) != (B, self.latent_dim):
                raise ValueError("eps must have shape [B, latent_dim]")
        if self._seed is None:
            # one stream per (torch seed, rank): `--seed` selects it (train_hybrid.py:1138-1141 seeds torch the same way) and
            # the ranks of a data-parallel job draw independent noise for their shards of the global batch
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x5EED + 0xD1B54A32D192ED03 * rank) & 0xFFFFFFFFFFFFFFFF
            self._seed = lcg_advance(self._seed, self.noise_calls)       # > 0 after a resume: this rank's stream, same position
        self._seed = (self._seed * _LCG_A + _LCG_C) & _M64
        self.noise_calls += 1
        eng.gen_enc += 1
        eng.gen_dec += 1
        _lib.check(_lib.lib.lo_vae_forward(eng.handle, x.data_ptr(), _lib.ptr(eps), self._seed, self._flat.data_ptr(),
                                           eng.ws.data_ptr(), recon.data_ptr(), mu.data_ptr(), logvar.data_ptr(),
                                           _lib.ptr(target), _lib.stream_ptr()), "lo_vae_forward")
        return recon, mu, logvar, eng

    def _native_encode(self, x: torch.Tensor):
        """Encoder.forward (lunar_generate.py:127-153): mu, logvar, the three skip maps (fp32 NCHW) and the engine that holds the
        activations."""
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 128, 128):
            raise ValueError(f"expected input of shape [B, 3, 128, 128], got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        B, dev = x.shape[0], x.device
        eng = self._engine(B)
        mu = torch.empty(B, self.latent_dim, dtype=torch.float32, device=dev)
        logvar = torch.empty_like(mu)
        sk = [torch.empty(B, 64 << k, 64 >> k, 64 >> k, dtype=torch.float32, device=dev) for k in range(3)]
        eng.gen_enc += 1
        _lib.check(_lib.lib.lo_vae_encode(eng.handle, x.data_ptr(), self._flat.data_ptr(), eng.ws.data_ptr(), mu.data_ptr(),
                                          logvar.data_ptr(), sk[0].data_ptr(), sk[1].data_ptr(), sk[2].data_ptr(), _lib.stream_ptr()),
                   "lo_vae_encode")
        return mu, logvar, sk[0], sk[1], sk[2], eng

    def _native_decode(self, z: torch.Tensor, skips: Sequence[torch.Tensor]):
        """Decoder.forward(z, skips) (lunar_generate.py:194-229) with 0..3 skip maps."""
        z = z.detach().contiguous().float()
        if z.dim() != 2 or z.shape[1] != self.latent_dim:
            raise ValueError("z must have shape [B, latent_dim]")
        B = z.shape[0]
        sk = []
        for k, t in enumerate(list(skips)[:3]):
            if tuple(t.shape) != (B, 64 << k, 64 >> k, 64 >> k):
                raise ValueError(f"skips[{k}] must have shape {(B, 64 << k, 64 >> k, 64 >> k)}, got {tuple(t.shape)}")
            sk.append(t.detach().contiguous().float())
        eng = self._engine(B)
        recon = torch.empty(B, 3, 128, 128, dtype=torch.float32, device=z.device)
        p = [_lib.ptr(sk[k]) if k < len(sk) else None for k in range(3)]
        eng.gen_dec += 1
        _lib.check(_lib.lib.lo_vae_decode_skips(eng.handle, z.data_ptr(), len(sk), p[0], p[1], p[2], self._flat.data_ptr(), eng.ws.data_ptr(),
                                                recon.data_ptr(), _lib.stream_ptr()), "lo_vae_decode_skips")
        return recon, eng

    def forward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """(reconstruction, mu, logvar), lunar_generate.py:263-276.  ``eps`` optionally injects the N(0,1) noise of
        ``reparameterize`` (parity runs); by default it is drawn on the device."""
        self._ensure_flat()
        x = x.detach().contiguous().float()
        if eps is None and self.next_eps is not None:
            eps, self.next_eps = self.next_eps, None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _VAEFunction.apply(self, x, eps, *self.parameters())
        recon, mu, logvar, _ = self._native_forward(x, eps, None)
        return recon, mu, logvar

    def reparameterize(self, mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
        """z = mu + eps * exp(0.5 logvar), lunar_generate.py:248-261 (tiny [B, L] op; PyTorch elementwise)."""
        std = torch.exp(0.5 * logvar)
        return mu + torch.randn_like(std) * std

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """Decoder without skip connections (`self.decoder(z, skips=[])`, lunar_generate.py:290) on the native path."""
        z = z.detach().contiguous().float()
        if z.dim() != 2 or z.shape[1] != self.latent_dim:
            raise ValueError("z must have shape [B, latent_dim]")
        eng = self._engine(z.shape[0])
        recon = torch.empty(z.shape[0], 3, 128, 128, dtype=torch.float32, device=z.device)
        eng.gen_dec += 1
        _lib.check(_lib.lib.lo_vae_decode(eng.handle, z.data_ptr(), self._flat.data_ptr(), eng.ws.data_ptr(), recon.data_ptr(),
                                          _lib.stream_ptr()), "lo_vae_decode")
        return recon

    def sample(self, num_samples: int) -> torch.Tensor:
        """lunar_generate.py:278-291: images from z ~ N(0, I), decoder without skips."""
        self._ensure_flat()
        z = torch.randn(num_samples, self.latent_dim, device=self._flat.device)
        return self.decode(z)
