"""The VAE half of ``TrainingManager._process_batch`` (reference: /root/reference/train_hybrid.py:838-954) as one
native sequence on the current HIP stream: forward + MSE/KL partial sums -> loss finalize -> backward ->
global-norm clip + AdamW -> (host) cosine-warm-restart LR.  No host synchronisation inside a step; metrics are
returned as a device tensor and only converted on request.

What is mirrored, line by line:
  :841-842  zero_grad            -> every gradient element is overwritten by the backward (no accumulation state)
  :850      vae(images)          -> lo_vae_forward (eps drawn on device unless injected)
  :859,862  recon_loss, kl_loss  -> fused partial sums + lo_vae_loss
  :886-889  vae_loss             -> (recon_weight - mean_advantage)*MSE + kl_weight*KL   (pg_loss = -mean(adv)*MSE)
  :895      / accumulation steps
  :907      optimizer steps only when (batch_idx+1) % accum == 0 — with the reference's zero_grad placement the
            gradients of the other micro-batches are discarded, so their backward is skipped here
  :913      clip_grad_norm_(max_grad_norm)   :921 AdamW(lr, wd=0.01, betas=(0.9,0.999))   :925 scheduler.step()
The teacher's contribution is the detached scalar ``mean_advantage``; it is an input here.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Optional

import torch

from . import _lib
from .vae import LunarisCoreVAE


def cosine_warm_restarts_lr(base_lr: float, eta_min: float, t0: int, t_mult: int, epoch: int) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingWarmRestarts after ``epoch`` calls of step() (train_hybrid.py:516-521)."""
    t_i, t_cur = t0, epoch
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= t_mult
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t_cur / t_i)) / 2.0


class LossScalePolicy:
    """GradScaler's policy (torch.cuda.amp.GradScaler, used by train_hybrid.py:917-923) for observations that LAG the device.

    `observe(scale, skipped_total, at, now)`: `skipped_total` = the device's count of skipped updates (non-finite gradient norm)
    after optimizer step `at`; `now` = the step the host has enqueued up to.  The host enqueues a step or two ahead of what it
    observes, so after one genuine overflow the steps already in the queue overflow too — at the OLD scale.  Those are the same
    overflow, not new ones (counting them cost 2-3 halvings and lost updates per event where GradScaler.update() costs
    one): of the updates skipped since the last observation, as many as ran before the last scale change are ignored; any beyond
    that ran at the current scale and halve it once.  `growth_interval` steps without a skip double it again, up to `init`."""

    def __init__(self, init: float = 65536.0, growth_interval: int = 2000):
        self.init, self.growth_interval = init, growth_interval
        self.skipped_seen, self.changed_at, self.last_obs = 0, 0, 0

    def observe(self, scale: float, skipped_total: int, at: int, now: int) -> float:
        new = skipped_total - self.skipped_seen
        at_old_scale = max(0, min(at, self.changed_at) - self.last_obs)      # steps of this window that ran before the change
        self.skipped_seen, self.last_obs = max(skipped_total, self.skipped_seen), max(at, self.last_obs)
        if new > at_old_scale:
            self.changed_at = now               # every step enqueued from now on runs at the new scale
            return max(1.0, scale * 0.5)
        if new <= 0 and scale < self.init and now - self.changed_at >= self.growth_interval:
            self.changed_at = now
            return min(self.init, scale * 2.0)
        return scale


class VAEStepper:
    _DP_LAG = 2     # data parallel: an observation is evaluated this many optimizer steps after it was enqueued

    def __init__(self, vae: LunarisCoreVAE, lr: float = 1e-4, min_lr: float = 1e-6, scheduler_t0: int = 10,
                 weight_decay: float = 0.01, max_grad_norm: float = 1.0, recon_weight: float = 1.0, kl_weight: float = 0.1,
                 gradient_accumulation_steps: int = 1, betas=(0.9, 0.999), eps: float = 1e-8,
                 grad_sync: Optional[Callable[[torch.Tensor], None]] = None, pipeline_optimizer: bool = False):
        _lib.require_gpu()
        self.vae = vae
        # pipeline_optimizer: the update of the Linear / decoder parameters (87 % of AdamW's 1.7 GB) and their operand refresh run
        # on the library's side stream beside the NEXT step's encoder forward (lo_vae_optimizer_step).  Parameters may then still
        # be in flight when step() returns: call synchronize_parameters() before touching them outside this stepper
        # (state_dict(), checkpoints, your own torch ops); forward / decode / sample through the module order themselves.
        self.pipeline_optimizer = bool(pipeline_optimizer)
        self._pipelined_engine = None
        self.base_lr, self.min_lr, self.t0 = lr, min_lr, scheduler_t0
        self.weight_decay, self.max_grad_norm = weight_decay, max_grad_norm
        self.recon_weight, self.kl_weight = recon_weight, kl_weight
        self.accum = max(1, int(gradient_accumulation_steps))
        self.betas, self.eps = betas, eps
        self.grad_sync = grad_sync      # data parallel: averages the flat gradient buffer across ranks (RCCL)
        # The weight gradients of the three Linear layers (82 % of the parameters) are rank-B matrices: the fused step keeps them as
        # their factors -- norm from Gram matrices, gradient tiles formed inside the AdamW pass (csrc/lo_lowrank.hip) -- instead of
        # writing and re-reading 201 MB.  Single process, batch <= 128; LO_LINEAR_FACTORED=0: the materialised path (A/B).
        # `parameter_grads()` still returns all 72 gradients (the two matrices are written out on demand).
        self.linear_factored = grad_sync is None and os.environ.get("LO_LINEAR_FACTORED", "1") != "0"
        # data parallel: the ranks all-gather the FACTORS (8.6 MB of fp16 per rank, exact) instead of exchanging the 201 MB of Linear
        # weight gradients, and each forms the averaged gradient from the gathered blocks (FlatGradSync.begin_factored)
        self.dp_factored = hasattr(grad_sync, "begin_factored") and os.environ.get("LO_LINEAR_FACTORED", "1") != "0"
        self.dp_three_phase = True      # hand the encoder's last stage over before stages 3..1 run (False: one encoder range)
        flat = vae.flat_parameters()
        self.grads = torch.zeros_like(flat)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.scratch = torch.zeros(1028, dtype=torch.float32, device=flat.device)   # [1024:1027] = norm, clip coef, finite
        self.losses = torch.zeros(4, dtype=torch.float32, device=flat.device)       # recon, kl, vae_loss, pg_loss
        self.opt_steps = 0
        self.last = None
        self._scale_policy = LossScalePolicy()
        # GradScaler policy without a host sync: the device's skipped-update counter is copied to pinned memory after optimizer
        # steps (asynchronously) and evaluated when the copy has landed, i.e. one or two steps later.  Each observation carries the
        # optimizer-step index it was taken after, so that updates skipped at a scale that has since been changed are not counted
        # again (see _update_loss_scale).  Data parallel: the ranks must change the scale at the SAME step (the scale multiplies
        # fp16 activation gradients, a rank at a higher scale can overflow alone), so there the observation taken `_DP_LAG` steps
        # earlier is waited for instead of polled -- a fixed lag, identical on every rank (the counters are: every rank sees the
        # same averaged gradients).
        self._skip_slots = [torch.zeros(1, dtype=torch.float32).pin_memory() for _ in range(self._DP_LAG + 1)]
        self._skip_inflight = []      # (event, pinned slot, optimizer-step index), oldest first
        self._last_engine = None
        self._presummed_begin: Optional[int] = None   # set by a backward that left the early part of the gradient norm in the scratch

    @property
    def lr(self) -> float:
        return cosine_warm_restarts_lr(self.base_lr, self.min_lr, self.t0, 2, self.opt_steps)

    def step(self, images: torch.Tensor, batch_idx: int = 0, eps: Optional[torch.Tensor] = None,
             mean_advantage: float = 0.0, adv_dev: Optional[torch.Tensor] = None):
        """One micro-batch.  Returns (recon, mu, logvar); losses stay on the device in ``self.losses``."""
        vae = self.vae
        images = images.detach().contiguous().float()
        recon, mu, logvar, eng = vae._native_forward(images, eps, target=images)
        st = _lib.stream_ptr()
        _lib.check(_lib.lib.lo_vae_loss(eng.handle, eng.ws.data_ptr(), self.recon_weight, self.kl_weight, float(mean_advantage),
                                        _lib.ptr(adv_dev), float(self.accum), float(vae.loss_scale), self.losses.data_ptr(), st),
                   "lo_vae_loss")
        if (batch_idx + 1) % self.accum == 0:
            flat = vae._flat
            self._backward_and_exchange(eng, images, recon, st)
            lr = self.lr
            self.opt_steps += 1
            self._clip_adamw(flat, lr, st, eng)
            self._observe_skipped_updates()
        self.last = (recon, mu, logvar)
        return recon, mu, logvar

    def _observe_skipped_updates(self) -> None:
        """Runs the loss-scale policy every optimizer step (the reference's GradScaler.update(), train_hybrid.py:917-923) with no
        host synchronisation in the single-process case: the skipped-update counter of the step just enqueued travels to pinned
        memory behind it; whichever earlier copy has landed by now is evaluated.  The host runs a step or two ahead of the GPU, so
        an overflow is answered after that many skipped updates — not after `--log_every` of them, as when only metrics() looked."""
        dp = self.grad_sync is not None and getattr(self.grad_sync, "world", 1) > 1
        if dp:
            # fixed lag: wait for the observation of `_DP_LAG` steps ago (long finished unless the host runs further ahead than that)
            while len(self._skip_inflight) >= self._DP_LAG:
                ev, slot, at = self._skip_inflight.pop(0)
                ev.synchronize()
                self._update_loss_scale(float(slot[0]), at)
        else:
            while self._skip_inflight and self._skip_inflight[0][0].query():
                ev, slot, at = self._skip_inflight.pop(0)
                self._update_loss_scale(float(slot[0]), at)
        if len(self._skip_inflight) < len(self._skip_slots):
            busy = {id(s_) for _, s_, _ in self._skip_inflight}
            slot = next(s_ for s_ in self._skip_slots if id(s_) not in busy)
            slot.copy_(self.scratch[1027:1028], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._skip_inflight.append((ev, slot, self.opt_steps))

    def _backward_and_exchange(self, eng, images: torch.Tensor, recon: torch.Tensor, st) -> None:
        """Native backward of the fused loss into ``self.grads`` (+ the data-parallel exchange, overlapped with it)."""
        vae = self.vae
        flat = vae._flat
        self._last_engine = eng
        bargs = (images.data_ptr(), flat.data_ptr(), eng.ws.data_ptr(), recon.data_ptr(), images.data_ptr(), 1, None, None, None,
                 float(vae.loss_scale), self.grads.data_ptr(), st)
        if self.grad_sync is not None and hasattr(self.grad_sync, "begin"):
            # data parallel: everything from fc_mu.weight to the end of the flat buffer (Linear layers, decoder and final
            # convs: 90 % of the bytes) is final after phase 1 and is exchanged while the encoder backward runs; the
            # encoder's last stage (94 % of the encoder bytes) is final after phase 3 and is exchanged during stages 3..1;
            # only the small remainder (7.7 MB) is exchanged with nothing left to hide it
            import ctypes as C
            b, e, b4, e4 = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
            kw = {}
            _lib.check(_lib.lib.lo_vae_phase1_grad_range(eng.handle, C.byref(b), C.byref(e)), "lo_vae_phase1_grad_range")
            _lib.check(_lib.lib.lo_vae_stage4_grad_range(eng.handle, C.byref(b4), C.byref(e4)), "lo_vae_stage4_grad_range")
            assert e4.value == b.value and e.value == self.grads.numel()
            # FlatGradSync runs the exchange on its own stream: the phases then hand their range over as an event for THAT stream
            # (lo_vae_set_async_handover) instead of holding this one up until the side stream's weight gradients have finished
            active = getattr(self.grad_sync, "world", 1) > 1 or getattr(self.grad_sync, "force", False)
            dpf = self.dp_factored and active and eng.batch <= 128
            if eng.fac_mode != (2 if dpf else 0):
                _lib.check(_lib.lib.lo_vae_set_linear_factored(eng.handle, 2 if dpf else 0), "lo_vae_set_linear_factored")
                eng.fac_mode = 2 if dpf else 0
            hooks = getattr(self.grad_sync, "supports_then", False)
            _lib.check(_lib.lib.lo_vae_set_async_handover(eng.handle, 1 if hooks else 0), "lo_vae_set_async_handover")
            def wait_range():
                _lib.check(_lib.lib.lo_vae_wait_handover(eng.handle, _lib.stream_ptr()), "lo_vae_wait_handover")
            kw = {"pre": wait_range} if hooks else {}
            _lib.check(_lib.lib.lo_vae_backward_phase(eng.handle, 1, *bargs), "lo_vae_backward_phase(1)")
            # ... and so is the sum of squares of that range for clip_grad_norm_ (of the AVERAGED gradients): taken right behind the
            # exchange on the communication stream, beside the encoder backward; the tail of the step reads only the encoder range
            early = os.environ.get("LO_EARLY_NORM", "1") != "0" and hooks
            if dpf:
                # the Linear weight gradients of this range travel as factors: gather, form the averaged matrices locally, average the
                # rest of the range (head biases; decoder.fc.bias, decoder and final convs), then the range's share of the norm
                fo, fb = C.c_size_t(), C.c_size_t()
                _lib.check(_lib.lib.lo_vae_factor_block(eng.handle, C.byref(fo), C.byref(fb)), "lo_vae_factor_block")
                factors = eng.ws[fo.value:fo.value + fb.value]
                L = vae.latent_dim
                h_end, d_beg, d_end = b.value + 2 * L * 32768, None, None
                lb, le = C.c_size_t(), C.c_size_t()
                _lib.check(_lib.lib.lo_vae_linear_grad_range(eng.handle, C.byref(lb), C.byref(le)), "lo_vae_linear_grad_range")
                d_end = le.value - 32768                 # decoder.fc.bias is the last tensor of the Linear range (32768 elements, aligned)
                d_beg = d_end - 32768 * L
                pieces = [self.grads[h_end:d_beg], self.grads[d_end:e.value]]

                def materialize(gathered, world, eng=eng):
                    _lib.check(_lib.lib.lo_vae_materialize_gathered_linear_grads(eng.handle, gathered.data_ptr(), world, self.grads.data_ptr(),
                                                                                 _lib.stream_ptr()), "lo_vae_materialize_gathered_linear_grads")

                def norm_of_range():
                    _lib.check(_lib.lib.lo_gradnorm_early_range(self.grads.data_ptr(), b.value, e.value, self.scratch.data_ptr(), _lib.stream_ptr()),
                               "lo_gradnorm_early_range")
                if self.grad_sync.begin_factored(pieces, factors, materialize, then=norm_of_range if early else None, **kw) and early:
                    self._presummed_begin = b.value
            elif early:
                if self.grad_sync.begin(self.grads[b.value:e.value], sumsq_scratch=self.scratch.data_ptr(), **kw):
                    self._presummed_begin = b.value
            else:
                self.grad_sync.begin(self.grads[b.value:e.value], **kw)
            if self.dp_three_phase:
                _lib.check(_lib.lib.lo_vae_backward_phase(eng.handle, 3, *bargs), "lo_vae_backward_phase(3)")
                self.grad_sync.begin(self.grads[b4.value:e4.value], **kw)
                _lib.check(_lib.lib.lo_vae_backward_phase(eng.handle, 4, *bargs), "lo_vae_backward_phase(4)")
                self.grad_sync.begin(self.grads[:b4.value])
            else:                    # two-call form: the whole encoder range after phase 2
                _lib.check(_lib.lib.lo_vae_backward_phase(eng.handle, 2, *bargs), "lo_vae_backward_phase(2)")
                self.grad_sync.begin(self.grads[:b.value])
            self.grad_sync.finish()
        else:
            # single process: the backward takes the sum of squares of everything from fc_mu.weight on as soon as it is final,
            # beside the encoder backward; _clip_adamw then reads only the encoder range for the norm
            fac = self.linear_factored and eng.batch <= 128
            if eng.fac_mode != (1 if fac else 0):
                _lib.check(_lib.lib.lo_vae_set_linear_factored(eng.handle, 1 if fac else 0), "lo_vae_set_linear_factored")
                eng.fac_mode = 1 if fac else 0
            early = self.grad_sync is None and (fac or os.environ.get("LO_EARLY_NORM", "1") != "0")       # LO_EARLY_NORM=0: A/B knob
            _lib.check(_lib.lib.lo_vae_set_gradnorm_scratch(eng.handle, self.scratch.data_ptr() if early else None),
                       "lo_vae_set_gradnorm_scratch")
            _lib.check(_lib.lib.lo_vae_backward(eng.handle, *bargs), "lo_vae_backward")
            if self.grad_sync is not None:
                self.grad_sync(self.grads)
            elif _lib.lib.lo_vae_gradnorm_presummed(eng.handle):
                import ctypes as C
                b, e = C.c_size_t(), C.c_size_t()
                _lib.check(_lib.lib.lo_vae_phase1_grad_range(eng.handle, C.byref(b), C.byref(e)), "lo_vae_phase1_grad_range")
                self._presummed_begin = b.value

    def synchronize_parameters(self) -> None:
        """Make the current stream wait for a pipelined optimizer step's side-stream work (no-op otherwise)."""
        if self._pipelined_engine is not None:
            _lib.check(_lib.lib.lo_vae_join(self._pipelined_engine.handle, _lib.stream_ptr()), "lo_vae_join")

    def _clip_adamw(self, flat: torch.Tensor, lr: float, st, eng=None) -> None:
        """clip_grad_norm_ + AdamW on the VAE's flat buffers (train_hybrid.py:913,921)."""
        args = (float(self.max_grad_norm), float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                float(self.weight_decay), self.opt_steps, self.scratch.data_ptr(), st)
        pre, self._presummed_begin = self._presummed_begin, None
        if eng is not None:
            # through the engine in both orders (LO_OPT_SERIAL = norm, AdamW, re-pack in order on this stream): its clip kernel reads the
            # plan's rendezvous-failure word and skips the update on the device when a fused-GroupNorm wait of this step ran out
            flags = (1 if pre is not None else 0) | (0 if self.pipeline_optimizer else 2)
            _lib.check(_lib.lib.lo_vae_optimizer_step(eng.handle, flat.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                                      self.exp_avg_sq.data_ptr(), eng.ws.data_ptr(), *args[:7], self.scratch.data_ptr(),
                                                      flags, st), "lo_vae_optimizer_step")
            self.vae.mark_weights_changed()
            eng.packed_version = self.vae._current_version()     # this engine's operand copies were refreshed by the call itself
            if self.pipeline_optimizer:
                self._pipelined_engine = eng
            return
        if pre is not None:
            _lib.check(_lib.lib.lo_clip_adamw_step_presummed(flat.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                                             self.exp_avg_sq.data_ptr(), flat.numel(), pre, *args), "lo_clip_adamw_step_presummed")
        else:
            _lib.check(_lib.lib.lo_clip_adamw_step(flat.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                   flat.numel(), *args), "lo_clip_adamw_step")

    def decode_sprites(self, u8_hwc: torch.Tensor) -> torch.Tensor:
        """uint8 [B,128,128,3] on the device -> normalised float32 [B,3,128,128] (train_hybrid.py:181-182), native kernel."""
        if u8_hwc.dtype != torch.uint8 or tuple(u8_hwc.shape[1:]) != (128, 128, 3) or not u8_hwc.is_cuda:
            raise ValueError("expected a CUDA uint8 tensor of shape [B,128,128,3]")
        u8_hwc = u8_hwc.contiguous()
        out = torch.empty(u8_hwc.shape[0], 3, 128, 128, dtype=torch.float32, device=u8_hwc.device)
        _lib.check(_lib.lib.lo_decode_sprites_u8(u8_hwc.data_ptr(), out.data_ptr(), u8_hwc.shape[0], _lib.stream_ptr()),
                   "lo_decode_sprites_u8")
        return out

    def _update_loss_scale(self, skipped_total: float, observed_after_step: Optional[int] = None) -> None:
        """Feed one observation of the device's skipped-update counter to the loss-scale policy (`LossScalePolicy`)."""
        at = self.opt_steps if observed_after_step is None else int(observed_after_step)
        self.vae.loss_scale = self._scale_policy.observe(self.vae.loss_scale, int(skipped_total), at, self.opt_steps)

    def metrics(self) -> Dict[str, float]:
        """Host copy of the last step's scalars (this synchronises the stream)."""
        v = torch.cat([self.losses, self.scratch[1024:1028], self._sync_fail_word()]).cpu().tolist()
        self._check_sync(v[8])
        self._update_loss_scale(v[7])
        return {"recon_loss": v[0], "kl_loss": v[1], "vae_loss": v[2], "pg_loss": v[3], "grad_norm": v[4],
                "clip_coef": v[5], "grads_finite": v[6], "lr": self.lr, "skipped_steps": v[7], "loss_scale": self.vae.loss_scale}

    def _sync_fail_word(self) -> torch.Tensor:
        """The engines' "a fused-GroupNorm rendezvous ran out" words (lo_vae_sync_fail_word), summed, as one float."""
        words = [e.sync_fail for e in self.vae._engines.values()]
        if not words:
            return torch.zeros(1, dtype=torch.float32, device=self.losses.device)
        return torch.stack([w.float().sum() for w in words]).sum().reshape(1)

    @staticmethod
    def _check_sync(word: float) -> None:
        if word != 0:
            raise _lib.LunarisHipError("a fused GroupNorm epilogue gave up waiting for the other workgroups of its sample (a launch of a "
                                       "step did not complete).  No parameter update has been applied since: the optimizer's clip kernel "
                                       "reads the same word and skips the update on the device.  LO_GNB_APPLY_FUSE=0 / LO_GN_FUSE=0 run "
                                       "the separate passes")

    def check_device_health(self) -> None:
        """Host-synchronising look at the rendezvous-failure word (checkpoints call it before they write anything)."""
        self._check_sync(float(self._sync_fail_word().item()))

    def flat_grads(self) -> torch.Tensor:
        """The flat gradient buffer with every gradient of the last step in it (in the factored mode the two Linear weight
        gradients are written out first; `self.grads` alone does not hold them then)."""
        self.parameter_grads()
        return self.grads

    def parameter_grads(self):
        """Views of the flat gradient buffer, one per parameter (state_dict order).  In the factored mode the two Linear weight
        gradients of the last step are written out first (`lo_vae_materialize_linear_grads`: the tiles the AdamW pass formed)."""
        for eng in self.vae._engines.values():
            if eng is self._last_engine and _lib.lib.lo_vae_linear_factored(eng.handle):
                _lib.check(_lib.lib.lo_vae_materialize_linear_grads(eng.handle, eng.ws.data_ptr(), self.grads.data_ptr(), _lib.stream_ptr()),
                           "lo_vae_materialize_linear_grads")
        return [self.grads[o:o + n].view(shape) for (o, n, shape) in self.vae._layout]


class HybridStepper(VAEStepper):
    """Full `_process_batch` (train_hybrid.py:838-954): VAE step + the teacher, as the reference EXECUTES it.

    Per micro-batch: VAE forward; teacher(images) (its result is dead in the reference — the prompt embedding it
    produces is overwritten before use, lunar_evaluator.py:438 — but it runs in train mode, so its BatchNorm
    running-statistics update is a real side effect and is kept; `run_dead_teacher_call=False` skips it); losses;
    teacher(recon.detach()); reward / EMA baseline / advantage on the device (`lo_hybrid_reward`); the VAE objective
    with the detached mean advantage; VAE backward + clip + AdamW; teacher_loss backward for the only parameters that
    receive gradients in the reference (gate, quality_heads: SURVEY §3.2) + clip + AdamW on exactly those.
    Teacher dropout (the module's ``dropout_rate``, 0.1 like the reference unless constructed otherwise) is applied in both
    teacher calls, each with its own call seed; the heads' backward replays the masks of the evaluated call."""

    def __init__(self, vae: LunarisCoreVAE, teacher, teacher_lr: float = 1e-4, quality_weight: float = 0.5, reward_scale: float = 0.1,
                 semantic_weight: float = 0.5, baseline_momentum: float = 0.9, run_dead_teacher_call: bool = True,
                 teacher_full_backward: bool = False, **kw):
        super().__init__(vae, **kw)
        self.teacher = teacher
        # the teacher trained "as documented" -- every parameter on the path of the teacher loss gets its gradient
        # (lo_teacher_full_backward: the reference with non-reentrant checkpoints) and its AdamW update; off = the reference as it
        # executes (gate + quality heads only)
        self.teacher_full_backward = bool(teacher_full_backward)
        self.teacher_base_lr = teacher_lr
        self.quality_weight, self.reward_scale = quality_weight, reward_scale
        self.semantic_weight, self.baseline_momentum = semantic_weight, baseline_momentum
        self.run_dead_teacher_call = run_dead_teacher_call
        dev = vae.flat_parameters().device
        self.reward_state = torch.zeros(2, dtype=torch.float32, device=dev)     # baseline, initialised flag
        self.reward_out = torch.zeros(8, dtype=torch.float32, device=dev)
        self.adv_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._t_ready = False

    def _teacher_setup(self, batch: int):
        import ctypes as C
        t = self.teacher
        h, ws, _ = t._engine(batch)
        b, e = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib.lo_teacher_grad_range(h, C.byref(b), C.byref(e)), "lo_teacher_grad_range")
        if not self._t_ready:
            self.t_heads_range = (b.value, e.value)
            self.t_range = (0, t._flat.numel()) if self.teacher_full_backward else (b.value, e.value)
            b, e = C.c_size_t(self.t_range[0]), C.c_size_t(self.t_range[1])
            n = e.value - b.value
            self.t_grads = torch.zeros_like(t._flat)
            self.t_m = torch.zeros(n, dtype=torch.float32, device=t._flat.device)
            self.t_v = torch.zeros(n, dtype=torch.float32, device=t._flat.device)
            self.t_scratch = torch.zeros(1028, dtype=torch.float32, device=t._flat.device)
            self._t_ready = True
            pending = getattr(self, "_pending_teacher_opt", None)
            if pending is not None:                       # optimizer state of a checkpoint (hostside.restore_checkpoint)
                from collections import OrderedDict
                from .hostside import flat_offsets, load_adamw_state_dict
                tp = OrderedDict(t.named_parameters())
                offs = flat_offsets(tp, t._flat)
                rel = {k: (o - b.value if b.value <= o < e.value else 0) for k, o in offs.items()}
                names = list(tp.keys())
                live = {"state": {i: st for i, st in pending.get("state", {}).items() if b.value <= offs[names[int(i)]] < e.value}}
                load_adamw_state_dict(live, tp, self.t_m, self.t_v, rel)
                self._pending_teacher_opt = None
        hb, he = self.t_heads_range
        rows = getattr(self, "_t_rows", None)
        if rows is None or rows.numel() != batch * (he - hb):
            self._t_rows = torch.empty(batch * (he - hb), dtype=torch.float32, device=t._flat.device)
        return h, ws

    def step(self, images: torch.Tensor, batch_idx: int = 0, eps: Optional[torch.Tensor] = None, **_):
        vae, t = self.vae, self.teacher
        images = images.detach().contiguous().float()
        B = images.shape[0]
        st = _lib.stream_ptr()
        recon, mu, logvar, eng = vae._native_forward(images, eps, target=images)
        if self.run_dead_teacher_call:
            t.update_statistics_only(images)            # train_hybrid.py:853-855 (side effects only)
        # train_hybrid.py:865 (no autograd node: the head gradients are taken below).  Full-backward mode: the plain forward that keeps
        # every block's output for lo_teacher_full_backward (only on the micro-batch whose gradients are used)
        tout = t._native_forward(recon, keep=self.teacher_full_backward and (batch_idx + 1) % self.accum == 0)[0]
        self.last_teacher_out = tout
        h, ws = self._teacher_setup(B)
        q_rows, s_rows, n_rows = tout["quality_scores"], tout["semantic_score"], B
        if self.grad_sync is not None and (getattr(self.grad_sync, "world", 1) > 1 or getattr(self.grad_sync, "force", False)):
            # data parallel: the baseline / advantage are functions of the batch means only (train_hybrid.py:870-883);
            # one 5-float all-reduce keeps them identical on every rank
            means = torch.cat([q_rows.mean(dim=0), s_rows.mean(dim=0)]).contiguous()
            self.grad_sync.average_small(means)
            self._reward_rows = (means[:4].reshape(1, 4).contiguous(), means[4:5].reshape(1, 1).contiguous())
            q_rows, s_rows = self._reward_rows
            n_rows = 1
        _lib.check(_lib.lib.lo_hybrid_reward(q_rows.data_ptr(), s_rows.data_ptr(), n_rows,
                                             float(self.semantic_weight), float(self.reward_scale), float(self.baseline_momentum),
                                             float(self.quality_weight), float(self.accum), self.reward_state.data_ptr(),
                                             self.reward_out.data_ptr(), self.adv_dev.data_ptr(), st), "lo_hybrid_reward")
        _lib.check(_lib.lib.lo_vae_loss(eng.handle, eng.ws.data_ptr(), self.recon_weight, self.kl_weight, 0.0, self.adv_dev.data_ptr(),
                                        float(self.accum), float(vae.loss_scale), self.losses.data_ptr(), st), "lo_vae_loss")
        if (batch_idx + 1) % self.accum == 0:
            flat = vae._flat
            self._backward_and_exchange(eng, images, recon, st)     # data parallel: exchange overlapped with the backward
            lr = self.lr
            t_lr = cosine_warm_restarts_lr(self.teacher_base_lr, self.min_lr, self.t0, 2, self.opt_steps)
            self.opt_steps += 1
            self._clip_adamw(flat, lr, st, eng)
            self._observe_skipped_updates()
            if self.teacher_full_backward:
                self._teacher_full_update(t, recon, tout, t_lr, st)
                self.last = (recon, mu, logvar)
                return recon, mu, logvar
            # teacher: gate + quality heads only (train_hybrid.py:891-904, 914, 922)
            b, e = self.t_range
            _lib.check(_lib.lib.lo_teacher_heads_backward(h, t._flat.data_ptr(), ws.data_ptr(), tout["expert_weights"].data_ptr(),
                                                          float(self.quality_weight) / float(self.accum), self._t_rows.data_ptr(),
                                                          self.t_grads.data_ptr(), st), "lo_teacher_heads_backward")
            if self.grad_sync is not None:
                self.grad_sync(self.t_grads[b:e])
            _lib.check(_lib.lib.lo_clip_adamw_step(t._flat[b:e].data_ptr(), self.t_grads[b:e].data_ptr(), self.t_m.data_ptr(),
                                                   self.t_v.data_ptr(), e - b, float(self.max_grad_norm), float(t_lr),
                                                   float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                                   float(self.weight_decay), self.opt_steps, self.t_scratch.data_ptr(), st),
                       "lo_clip_adamw_step(teacher)")
            # the gate / head weights are read in fp32 by the head kernels: no re-pack needed
        self.last = (recon, mu, logvar)
        return recon, mu, logvar

    def _teacher_full_update(self, t, recon, tout, t_lr, st):
        """teacher_loss.backward() + clip + AdamW with every teacher parameter live (train_hybrid.py:891-904, 914, 922 under
        non-reentrant checkpoints): lo_teacher_full_backward on the evaluated images, then lo_teacher_clip_adamw_full."""
        eng = t._engine(recon.shape[0])
        if getattr(eng, "bws", None) is None:
            eng.bws = torch.empty(_lib.lib.lo_teacher_full_backward_bytes(eng.handle), dtype=torch.uint8, device=recon.device)
        gscale = 64.0 * recon.shape[0] * 16384.0
        _lib.check(_lib.lib.lo_teacher_full_backward(eng.handle, recon.data_ptr(), t._flat.data_ptr(), eng.ws.data_ptr(), eng.bws.data_ptr(),
                                                     tout["expert_weights"].data_ptr(), float(self.quality_weight) / float(self.accum),
                                                     gscale, self._t_rows.data_ptr(), self.t_grads.data_ptr(), st), "lo_teacher_full_backward")
        if self.grad_sync is not None:
            self.grad_sync(self.t_grads)
        _lib.check(_lib.lib.lo_teacher_clip_adamw_full(eng.handle, t._flat.data_ptr(), self.t_grads.data_ptr(), self.t_m.data_ptr(),
                                                       self.t_v.data_ptr(), float(self.max_grad_norm), float(t_lr), float(self.betas[0]),
                                                       float(self.betas[1]), float(self.eps), float(self.weight_decay), self.opt_steps,
                                                       self.t_scratch.data_ptr(), st), "lo_teacher_clip_adamw_full")
        t.mark_weights_changed()              # conv / attention operands are packed fp16 copies: re-packed before the next forward

    def metrics(self) -> Dict[str, float]:
        """The 12 scalars of train_hybrid.py:929-942 (+ grad norm / lr); one host copy."""
        v = torch.cat([self.losses, self.scratch[1024:1027], self.reward_out[:7], self.scratch[1027:1028], self._sync_fail_word()]).cpu().tolist()
        self._check_sync(v[15])
        recon, kl, vae_loss, pg = v[0:4]
        q_loss, sem_r, q_r, baseline, adv, t_loss, q_mean = v[7:14]
        self._update_loss_scale(v[14])
        return {"recon_loss": recon, "kl_loss": kl, "quality_loss": q_loss, "pg_loss": pg, "semantic_reward": sem_r,
                "quality_reward": q_r, "baseline": baseline, "advantage": adv, "vae_loss": vae_loss, "teacher_loss": t_loss,
                "total_loss": vae_loss + t_loss, "quality_scores": q_mean, "grad_norm": v[4], "clip_coef": v[5],
                "grads_finite": v[6], "lr": self.lr, "skipped_steps": v[14], "loss_scale": self.vae.loss_scale}
