"""LunarMoETeacher on MI355X: the reference's ``nn.Module`` surface, forward in liblunaris_hip.so.

Drop-in contract (reference: /root/reference/lunar_evaluator.py:278-462): same constructor signature, the same 252
parameters + 99 buffers under the same ``state_dict`` keys (``feature_extractor.conv1.0.weight`` ...
``prompt_net.6.bias``; BatchNorm ``running_mean/var/num_batches_tracked``; the attention buffer
``last_spatial_shapes``), the reference's initialisation (``kaiming_normal_(fan_out, leaky_relu)``, zero biases,
unit norms, ``rel_pos ~ N(0, 0.02)``, ``layer_scale = 0.1``), ``forward(x, prompt_embedding=None)`` returning the same
six-key dict.  The sub-modules are containers; ``forward`` is one native call (``lo_teacher_forward``) that reproduces
the forward AS EXECUTED — including the chunk-index write offset of ``PixelArtAttention``.

Dropout (``dropout_rate``, default 0.1 like the reference; lunar_evaluator.py:97-99,139-140,212,225,246,253,353-397): in
train mode all six sites are applied on the device with torch semantics (``nn.Dropout`` elementwise, ``nn.Dropout2d`` per
(sample, channel), kept values scaled by 1/(1-p)).  The masks come from a counter RNG keyed by (torch seed, rank, call,
site, element) — ``oracle/dropout_ref.py`` regenerates them bit-exactly, which is how the parity tests run the CPU oracle
and the reference itself on the same masks.  With dropout the native forward runs every 3x3 convolution in full (the
constant-field shortcuts of the p = 0 path do not survive ``proj_drop``); eval mode and ``dropout_rate=0`` take the
shortcut path.  ``quality_scores`` and ``expert_weights`` are autograd-visible outputs: with gradients enabled the forward is an
autograd node over the gate / quality-head parameters — the only ones that receive gradients in the reference, whose reentrant
checkpoints cut the experts and the feature extractor from the graph — so the reference's own
``teacher_loss.backward()`` (train_hybrid.py:891-904) fills their ``.grad`` (``lo_teacher_heads_backward_ex``, replaying the
call's dropout masks).  ``trainer.HybridStepper`` drives the same kernel directly (``lo_teacher_heads_backward``).  In-place
parameter updates by any optimizer are noticed through the parameters' version counters (re-pack before the next forward).

``feature_dim``: 128 (the CLI default: folded attention + constant-field shortcuts when no dropout is active) or 256 / 512 (the
README's High-End recipe, /root/reference/README.md:102-118: a generic path with every tensor at full resolution and the
``ExpertBlock.shortcut`` Conv1x1 + BatchNorm of the first block, lunar_evaluator.py:254-257).  ``feature_maps`` is always ``None``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib


class PixelArtFeatureExtractor(nn.Module):
    """Parameter container (lunar_evaluator.py:57-112)."""

    def __init__(self, in_channels=3, dropout_rate=0.1, feature_dim=128):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, 32, 3, padding=1), nn.LeakyReLU(0.2), nn.BatchNorm2d(32))

        def branch(k):
            return nn.Sequential(nn.Conv2d(32, 32, k, padding=k // 2, groups=32), nn.Conv2d(32, 64, 1), nn.LeakyReLU(0.2), nn.BatchNorm2d(64))
        self.edge_branch, self.color_branch, self.detail_branch = branch(3), branch(5), branch(3)
        self.dropout = nn.Dropout(dropout_rate)
        self.fusion = nn.Sequential(nn.Conv2d(192, feature_dim, 1), nn.LeakyReLU(0.2), nn.BatchNorm2d(feature_dim))


class PixelArtAttention(nn.Module):
    """Parameter container (lunar_evaluator.py:119-144)."""

    def __init__(self, in_channels, num_heads=8, rel_pos_size=8, dropout=0.1, chunk_size=64):
        super().__init__()
        self.num_heads, self.head_dim, self.chunk_size = num_heads, in_channels // num_heads, chunk_size
        self.qkv = nn.Conv2d(in_channels, in_channels * 3, 1)
        self.proj = nn.Conv2d(in_channels, in_channels, 1)
        self.rel_pos_h = nn.Parameter(torch.randn(1, num_heads, rel_pos_size, 1) * 0.02)
        self.rel_pos_w = nn.Parameter(torch.randn(1, num_heads, 1, rel_pos_size) * 0.02)
        self.attn_drop, self.proj_drop = nn.Dropout(dropout), nn.Dropout(dropout)
        self.register_buffer("rel_pos_cache", None)
        self.register_buffer("last_spatial_shapes", torch.zeros(2))


class ExpertBlock(nn.Module):
    """Parameter container (lunar_evaluator.py:234-258)."""

    def __init__(self, in_channels, out_channels, dropout_rate=0.1, rel_pos_size=8, layer_scale_init=0.1):
        super().__init__()
        def cb(ci):
            return nn.Sequential(nn.Conv2d(ci, out_channels, 3, padding=1), nn.LeakyReLU(0.2), nn.BatchNorm2d(out_channels), nn.Dropout2d(dropout_rate))
        self.conv1 = cb(in_channels)
        self.attention = PixelArtAttention(out_channels, rel_pos_size=rel_pos_size, dropout=dropout_rate)
        self.conv2 = cb(out_channels)
        self.shortcut = (nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
                         if in_channels != out_channels else nn.Identity())
        self.layer_scale = nn.Parameter(torch.ones(1, out_channels, 1, 1) * layer_scale_init)


class _TeacherEngine:
    """Native plan + workspace for one batch size; the handle is released with the object (lo_teacher_destroy)."""

    def __init__(self, model: "LunarMoETeacher", batch: int):
        h = C.c_void_p()
        _lib.check(_lib.lib.lo_teacher_create_ex(batch, model.num_experts, model.feature_dim, model.embedding_dim,
                                                 1 if model.mfma_precision == "fp8" else 0, C.byref(h)), "lo_teacher_create_ex")
        self.handle = h
        self.ws = torch.empty(_lib.lib.lo_teacher_workspace_bytes(h), dtype=torch.uint8, device=model._flat.device)
        self.packed_version = None

    def __iter__(self):          # (handle, workspace, packed version): the tuple form older call sites unpack
        return iter((self.handle, self.ws, self.packed_version))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h is not None:
            try:
                _lib.lib.lo_teacher_destroy(h)
            except Exception:
                pass


class _TeacherFunction(torch.autograd.Function):
    """LunarMoETeacher.forward as an autograd node over the gate / quality-head parameters (lunar_evaluator.py:353-373, 417,
    431-432): differentiable outputs quality_scores [B,4] and expert_weights [B,E]; the embeddings and the semantic score are
    returned without a graph (nothing in the reference step differentiates them).  The head inputs of THIS call (pooled
    features, pre-weighting logits, dropout stream) are copied out of the workspace, so later forward calls do not disturb it."""

    @staticmethod
    def forward(ctx, model, x, *live):
        full = model.all_parameters_live and model.training
        out, eng, p, seed = model._native_forward(x, keep=full)
        ctx.full, ctx.x = full, (x if full else None)
        offs, elems = (C.c_size_t * 3)(), (C.c_size_t * 3)()
        _lib.check(_lib.lib.lo_teacher_heads_saved(eng.handle, offs, elems), "lo_teacher_heads_saved")
        saved = [eng.ws[offs[i]:offs[i] + 4 * elems[i]].view(torch.float32).clone() for i in range(3)]
        ctx.model, ctx.eng, ctx.drop = model, eng, (p, seed)
        ctx.live_offsets = [(t.data_ptr() - model._flat.data_ptr()) // 4 for t in live]
        ctx.live_shapes = [t.shape for t in live]
        ctx.save_for_backward(out["expert_weights"], *saved)
        ctx.mark_non_differentiable(out["style_embedding"], out["prompt_embedding"], out["semantic_score"])
        return out["quality_scores"], out["expert_weights"], out["style_embedding"], out["prompt_embedding"], out["semantic_score"]

    @staticmethod
    def backward(ctx, gq, gw, *_):
        model, eng = ctx.model, ctx.eng
        w, pooled_f, pooled_e, raw_q = ctx.saved_tensors
        cont = lambda t: None if t is None else t.contiguous().float()
        gq, gw = cont(gq), cont(gw)
        b, e = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib.lo_teacher_grad_range(eng.handle, C.byref(b), C.byref(e)), "lo_teacher_grad_range")
        rows = torch.empty(w.shape[0] * (e.value - b.value), dtype=torch.float32, device=w.device)
        grads = torch.zeros_like(model._flat)
        if ctx.full and (gq is not None or gw is not None):
            # every parameter on the path (lo_teacher_full_backward_ex).  A foreign loss scale (GradScaler: 65 536) is divided out on the
            # device first, like at the VAE's boundary, so that the fp16 activation gradients see upstream values of order 1
            from .vae import _normalise_upstream
            scr, (gq, gw) = _normalise_upstream([gq, gw])
            B = w.shape[0]
            if getattr(eng, "bws", None) is None:
                eng.bws = torch.empty(_lib.lib.lo_teacher_full_backward_bytes(eng.handle), dtype=torch.uint8, device=w.device)
            _lib.check(_lib.lib.lo_teacher_full_backward_ex(eng.handle, ctx.x.data_ptr(), model._flat.data_ptr(), eng.ws.data_ptr(), eng.bws.data_ptr(),
                                                            pooled_f.data_ptr(), pooled_e.data_ptr(), raw_q.data_ptr(), w.data_ptr(), _lib.ptr(gq),
                                                            _lib.ptr(gw), float(ctx.drop[0]), int(ctx.drop[1]), float(2 ** 20), rows.data_ptr(),
                                                            grads.data_ptr(), _lib.stream_ptr()), "lo_teacher_full_backward_ex")
            _lib.check(_lib.lib.lo_grad_unscale_dev(grads.data_ptr(), grads.numel(), scr.data_ptr() + 4, None, _lib.stream_ptr()),
                       "lo_grad_unscale_dev")
        elif gq is not None or gw is not None:
            _lib.check(_lib.lib.lo_teacher_heads_backward_ex(eng.handle, model._flat.data_ptr(), pooled_f.data_ptr(), pooled_e.data_ptr(),
                                                             raw_q.data_ptr(), w.data_ptr(), _lib.ptr(gq), _lib.ptr(gw), float(ctx.drop[0]),
                                                             int(ctx.drop[1]), rows.data_ptr(), grads.data_ptr(), _lib.stream_ptr()),
                       "lo_teacher_heads_backward_ex")
        out = []
        for o, shape in zip(ctx.live_offsets, ctx.live_shapes):
            n = 1
            for d in shape:
                n *= int(d)
            out.append(grads[o:o + n].view(shape))
        return (None, None) + tuple(out)


class LunarMoETeacher(nn.Module):
    def __init__(self, num_experts=4, feature_dim=128, dropout_rate=0.1, rel_pos_size=8, use_checkpointing=True,
                 expert_layers=3, intermediate_dim=256, embedding_dim=64, mfma_precision: str = "fp16", full_backward: bool = False):
        super().__init__()
        # full_backward (an addition of this build): the autograd graph of `quality_scores` / `expert_weights` covers EVERY
        # parameter on their path -- experts and feature extractor included -- i.e. the reference with `use_reentrant=False` at its three
        # checkpoint calls (lunar_evaluator.py:194-197, 266-275, 411-414).  Default: the reference as it executes (gate + quality heads only).
        self.all_parameters_live = bool(full_backward)
        # mfma_precision (an addition of this build): "fp16" (default, parity-tested) or "fp8" = OCP e4m3 operands in the 24
        # full-resolution 3x3 convolutions of the train-mode dropout path (BASELINE config 5); eval mode is fp16 either way
        if mfma_precision not in ("fp16", "fp8"):
            raise ValueError(f"mfma_precision must be 'fp16' or 'fp8', got {mfma_precision!r}")
        self.mfma_precision = mfma_precision
        if feature_dim not in (128, 256, 512) or expert_layers != 3 or intermediate_dim != 256 or rel_pos_size != 8:
            raise NotImplementedError("built: feature_dim 128 (the CLI default), 256 or 512 (README High-End recipe) with expert_layers=3, "
                                      "intermediate_dim=256, rel_pos_size=8 (the CLI defaults)")
        if feature_dim != 128 and mfma_precision != "fp16":
            raise NotImplementedError("the fp8 operand mode covers feature_dim 128 only")
        self.num_experts, self.feature_dim, self.dropout_rate = num_experts, feature_dim, dropout_rate
        self.rel_pos_size, self.use_checkpointing, self.expert_layers = rel_pos_size, use_checkpointing, expert_layers
        self.intermediate_dim, self.embedding_dim = intermediate_dim, embedding_dim
        self.feature_extractor = PixelArtFeatureExtractor(3, dropout_rate, 128)

        def expert():
            blocks, cin = [], 128
            for _ in range(expert_layers):
                blocks.append(ExpertBlock(cin, feature_dim, dropout_rate, rel_pos_size))
                cin = feature_dim
            return nn.Sequential(*blocks)
        self.experts = nn.ModuleList([expert() for _ in range(num_experts)])
        self.gate = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(128, intermediate_dim), nn.LeakyReLU(0.2),
                                  nn.Dropout(dropout_rate), nn.Linear(intermediate_dim, num_experts), nn.Softmax(dim=1))

        def head(hidden, out, sigmoid=False):
            mods = [nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.LayerNorm(feature_dim), nn.Linear(feature_dim, hidden), nn.LeakyReLU(0.2),
                    nn.Dropout(dropout_rate), nn.Linear(hidden, out)]
            return nn.Sequential(*(mods + ([nn.Sigmoid()] if sigmoid else [])))
        self.quality_heads = nn.ModuleList([head(intermediate_dim // 4, 4) for _ in range(num_experts)])
        self.semantic_head = head(intermediate_dim // 2, 1, sigmoid=True)
        self.style_net = head(intermediate_dim // 2, embedding_dim)
        self.prompt_net = head(intermediate_dim // 2, embedding_dim)
        self.apply(self._init_weights)
        self._flat: Optional[torch.Tensor] = None
        self._nbt: Optional[torch.Tensor] = None      # the BatchNorm step counters, stacked (see _ensure_flat)
        self._engines: Dict[int, "_TeacherEngine"] = {}
        self._weights_version = 0
        self._drop_seed: Optional[int] = None     # counter-RNG stream of the dropout masks; derived at the first train-mode forward
        self.drop_calls = 0                        # train-mode forwards drawn from the stream so far (checkpointed)
        self.last_drop_seed: Optional[int] = None  # the seed the last forward ran with (tests regenerate its masks from it)

    @staticmethod
    def _init_weights(m):
        """lunar_evaluator.py:399-406."""
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.BatchNorm2d, nn.LayerNorm)):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    # ---- flat state ---------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._flat = None
        self._engines.clear()
        return out

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self._weights_version += 1
        return out

    def mark_weights_changed(self):
        self._weights_version += 1

    def _named_state(self):
        params = dict(self.named_parameters())
        bufs = dict(self.named_buffers())
        for k in self.state_dict().keys():
            yield k, (params[k] if k in params else bufs[k])

    def _ensure_flat(self):
        first = next(self.parameters())
        if (self._flat is not None and self._flat.device == first.device
                and self._flat.data_ptr() <= first.data_ptr() < self._flat.data_ptr() + 4 * self._flat.numel()):
            return
        _lib.require_gpu()
        if first.device.type != "cuda":
            raise _lib.LunarisHipError("LunarMoETeacher must be on the GPU (model.to('cuda')); there is no CPU path")
        h = C.c_void_p()
        _lib.check(_lib.lib.lo_teacher_create(1, self.num_experts, self.feature_dim, self.embedding_dim, C.byref(h)), "lo_teacher_create")
        try:
            state = list(self._named_state())
            assert _lib.lib.lo_teacher_num_tensors(h) == len(state), "teacher state table size mismatch"
            flat = torch.zeros(_lib.lib.lo_teacher_flat_elems(h), dtype=torch.float32, device=first.device)
            with torch.no_grad():
                for i, (k, t) in enumerate(state):
                    assert _lib.lib.lo_teacher_tensor_name(h, i).decode() == k, (i, k)
                    assert _lib.lib.lo_teacher_tensor_numel(h, i) == t.numel(), k
                    off = _lib.lib.lo_teacher_tensor_offset(h, i)
                    if off < 0:
                        continue                      # integer buffer (num_batches_tracked) stays a normal tensor
                    flat[off:off + t.numel()].copy_(t.detach().reshape(-1).float())
                    t.data = flat[off:off + t.numel()].view(t.shape)
        finally:
            _lib.lib.lo_teacher_destroy(h)
        # the 29 BatchNorm step counters become views of ONE int64 tensor: a train-mode forward bumps them with one launch
        # (29 separate `num_batches_tracked += 1` cost 29 launches = 0.13 ms of GPU time and more of host time per forward)
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        with torch.no_grad():
            nbt = torch.stack([m.num_batches_tracked.detach().reshape(()).to(first.device, torch.int64) for m in bns])
            for i, m in enumerate(bns):
                m.num_batches_tracked.data = nbt[i]
        self._nbt = nbt
        self._flat = flat
        self._weights_version += 1
        self._engines.clear()

    def _engine(self, batch: int):
        self._ensure_flat()
        eng = self._engines.get(batch)
        if eng is None:
            eng = _TeacherEngine(self, batch)
            self._engines[batch] = eng
        # in-place updates by any optimizer move the parameters' version counters (the native head update writes through raw
        # pointers instead and touches only tensors the head kernels read in fp32: nothing to re-pack)
        version = (self._weights_version, sum(p._version for p in self.parameters()))
        if eng.packed_version != version:
            _lib.check(_lib.lib.lo_teacher_pack(eng.handle, self._flat.data_ptr(), eng.ws.data_ptr(), _lib.stream_ptr()), "lo_teacher_pack")
            eng.packed_version = version
        return eng

    def live_parameters(self):
        """The parameters that receive gradients in the reference step (gate.*, quality_heads.*: SURVEY §3.2), state_dict order; with
        ``LunarMoETeacher(full_backward=True)`` every parameter on the path of quality_scores / expert_weights (all but the style / prompt / semantic heads)."""
        if self.all_parameters_live:
            return [p for k, p in self.named_parameters() if k.split(".")[0] not in ("semantic_head", "style_net", "prompt_net")]
        return [p for k, p in self.named_parameters() if k.startswith("gate.") or k.startswith("quality_heads.")]

    # ---- dropout stream -----------------------------------------------------------------------------------
    def set_dropout_stream(self, seed: int, exact_next: bool = False) -> None:
        """Restart the dropout mask stream from ``seed`` (otherwise derived from ``torch.initial_seed()`` and the rank).
        ``exact_next``: the next train-mode forward uses ``seed`` itself as its call seed (parity tests)."""
        self._drop_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._drop_exact = bool(exact_next)

    def _next_drop_seed(self) -> int:
        """One 64-bit stream value per forward call (the same scheme as the VAE's noise stream, vae.py): keyed by the torch
        seed (`--seed`, train_hybrid.py:1138-1141) and the rank, advanced by an LCG per call."""
        if self._drop_seed is None:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._drop_seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0xD209 + 0xD1B54A32D192ED03 * rank) & 0xFFFFFFFFFFFFFFFF
            from .vae import lcg_advance
            self._drop_seed = lcg_advance(self._drop_seed, self.drop_calls)     # > 0 after a resume: same stream, same position
        if getattr(self, "_drop_exact", False):
            self._drop_exact = False
        else:
            self._drop_seed = (self._drop_seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            self.drop_calls += 1
        self.last_drop_seed = self._drop_seed
        return self._drop_seed

    def last_path(self, batch: int) -> int:
        """0 = sparse shortcuts, 1 = dense (LO_T_DENSE=1), 2 = dropout path; -1 before the first forward of that batch size."""
        eng = self._engines.get(batch)
        return -1 if eng is None else int(_lib.lib.lo_teacher_last_path(eng.handle))

    def update_statistics_only(self, x: torch.Tensor) -> None:
        """The side effects of `forward(x)` in train mode without its outputs: every BatchNorm layer sees the batch
        (running statistics, `num_batches_tracked`), the pooling of the last ExpertBlocks and the heads are skipped.
        `_process_batch` calls the teacher once this way (train_hybrid.py:853-855: the result is overwritten unused)."""
        if not self.training:
            return
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 128, 128):
            raise ValueError(f"expected input of shape [B, 3, 128, 128], got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        eng = self._engine(x.shape[0])
        p = float(self.dropout_rate)
        _lib.check(_lib.lib.lo_teacher_forward(eng.handle, x.data_ptr(), self._flat.data_ptr(), eng.ws.data_ptr(), 1, p,
                                               self._next_drop_seed() if p > 0 else 0, None, None, None, None, None,
                                               _lib.stream_ptr()), "lo_teacher_forward(statistics only)")
        with torch.no_grad():
            self._nbt += 1

    def _native_forward(self, x: torch.Tensor, keep: bool = False):
        """One `lo_teacher_forward` (keep: `lo_teacher_forward_keep`, the train-mode forward of a step that ends in `full_backward`: every
        block's output stays in the backward's scratch).  Returns (outputs, engine, dropout_p, call seed)."""
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 128, 128):
            raise ValueError(f"expected input of shape [B, 3, 128, 128], got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        B = x.shape[0]
        eng = self._engine(B)
        dev = x.device
        q = torch.empty(B, 4, dtype=torch.float32, device=dev)
        w = torch.empty(B, self.num_experts, dtype=torch.float32, device=dev)
        st = torch.empty(B, self.embedding_dim, dtype=torch.float32, device=dev)
        pr = torch.empty(B, self.embedding_dim, dtype=torch.float32, device=dev)
        sem = torch.empty(B, 1, dtype=torch.float32, device=dev)
        p = float(self.dropout_rate) if self.training else 0.0
        seed = self._next_drop_seed() if p > 0 else 0
        if keep and self.training:
            if getattr(eng, "bws", None) is None:
                eng.bws = torch.empty(_lib.lib.lo_teacher_full_backward_bytes(eng.handle), dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib.lo_teacher_forward_keep(eng.handle, x.data_ptr(), self._flat.data_ptr(), eng.ws.data_ptr(), eng.bws.data_ptr(),
                                                        p, seed, q.data_ptr(), w.data_ptr(), st.data_ptr(), pr.data_ptr(), sem.data_ptr(),
                                                        _lib.stream_ptr()), "lo_teacher_forward_keep")
        else:
            _lib.check(_lib.lib.lo_teacher_forward(eng.handle, x.data_ptr(), self._flat.data_ptr(), eng.ws.data_ptr(), 1 if self.training else 0,
                                                   p, seed, q.data_ptr(), w.data_ptr(), st.data_ptr(), pr.data_ptr(), sem.data_ptr(),
                                                   _lib.stream_ptr()), "lo_teacher_forward")
        if self.training:
            with torch.no_grad():
                self._nbt += 1
        return ({"quality_scores": q, "expert_weights": w, "style_embedding": st, "prompt_embedding": pr,
                 "semantic_score": sem, "feature_maps": None}, eng, p, seed)

    def full_backward(self, x: torch.Tensor, expert_weights: torch.Tensor, coef: float, gscale: float = None) -> torch.Tensor:
        """SURVEY §8 row F2, second half: gradients of ``coef * -mean(quality_scores)`` (train_hybrid.py:891-892, coef =
        quality_weight / accumulation steps) for EVERY teacher parameter on the loss path -- what ``teacher_loss.backward()``
        yields when the model's three ``checkpoint`` calls (lunar_evaluator.py:194-197, 266-275, 411-414) are non-reentrant.
        Must follow the train-mode ``forward(x)`` whose loss it differentiates (same images; the call's dropout stream and head
        inputs are replayed).  Returns the flat gradient in the layout of ``self._flat`` (``parameter_grad_views`` slices it)."""
        x = x.detach().contiguous().float()
        B = x.shape[0]
        eng = self._engine(B)
        if getattr(eng, "bws", None) is None:
            eng.bws = torch.empty(_lib.lib.lo_teacher_full_backward_bytes(eng.handle), dtype=torch.uint8, device=x.device)
        b, e = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib.lo_teacher_grad_range(eng.handle, C.byref(b), C.byref(e)), "lo_teacher_grad_range")
        rows = torch.empty(B * (e.value - b.value), dtype=torch.float32, device=x.device)
        grads = torch.empty_like(self._flat)
        if gscale is None:
            gscale = 64.0 * B * 16384.0                      # a power of two for power-of-two batches; any positive value is exact enough
        _lib.check(_lib.lib.lo_teacher_full_backward(eng.handle, x.data_ptr(), self._flat.data_ptr(), eng.ws.data_ptr(), eng.bws.data_ptr(),
                                                     expert_weights.contiguous().float().data_ptr(), float(coef), float(gscale), rows.data_ptr(),
                                                     grads.data_ptr(), _lib.stream_ptr()), "lo_teacher_full_backward")
        return grads

    def parameter_grad_views(self, flat_grads: torch.Tensor):
        """name -> view of ``flat_grads`` (layout of ``self._flat``) for every parameter."""
        base = self._flat.data_ptr()
        return {k: flat_grads[(p.data_ptr() - base) // 4:(p.data_ptr() - base) // 4 + p.numel()].view(p.shape) for k, p in self.named_parameters()}

    def forward(self, x: torch.Tensor, prompt_embedding=None):
        """lunar_evaluator.py:408-462.  ``prompt_embedding`` is accepted and ignored exactly like the reference does
        (it is overwritten at :438 before any use).  With gradients enabled, ``quality_scores`` / ``expert_weights`` carry a
        graph over the gate / quality-head parameters (see `_TeacherFunction`)."""
        self._ensure_flat()
        if torch.is_grad_enabled():
            live = [p for p in self.live_parameters() if p.requires_grad]
            if live:
                q, w, st, pr, sem = _TeacherFunction.apply(self, x.detach(), *live)
                return {"quality_scores": q, "expert_weights": w, "style_embedding": st, "prompt_embedding": pr,
                        "semantic_score": sem, "feature_maps": None}
        return self._native_forward(x)[0]
