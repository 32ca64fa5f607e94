"""Host-side pieces around the step that the reference keeps in `train_hybrid.py`: early stopping,
the comparison / sample PNGs, and checkpoints in the reference's dictionary layout (`vae_optimizer` / `teacher_optimizer`
as `torch.optim.AdamW.state_dict()`s, so a checkpoint written here resumes in the reference and vice versa).

Nothing here touches the GPU kernels; tensors arrive as CPU copies or are copied once."""
from __future__ import annotations

import time
from collections import OrderedDict
from pathlib import Path
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch


class EarlyStopping:
    """train_hybrid.py:206-225 — patience on a loss that must improve by more than `min_delta`."""

    def __init__(self, patience: int = 7, min_delta: float = 0.0):
        self.patience, self.min_delta = patience, min_delta
        self.counter, self.best_loss, self.early_stop = 0, None, False

    def __call__(self, loss: float) -> None:
        if self.best_loss is None:
            self.best_loss = loss
        elif loss > self.best_loss + self.min_delta:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_loss, self.counter = loss, 0


def to_uint8_hwc(images: torch.Tensor) -> np.ndarray:
    """[-1, 1] CHW float -> [0, 255] HWC uint8, the reference's `((x + 1) * 127.5).clamp(0, 255)` (train_hybrid.py:636, 745)."""
    x = ((images.detach().float().cpu() + 1.0) * 127.5).clamp(0, 255).numpy().astype(np.uint8)
    return np.transpose(x, (0, 2, 3, 1))


def save_comparison(path: Path, originals: torch.Tensor, recons: torch.Tensor, quality_scores: Optional[torch.Tensor] = None,
                    semantic_scores: Optional[torch.Tensor] = None) -> Path:
    """Original | generated side by side for up to 4 samples, with the teacher's scores (train_hybrid.py:718-789)."""
    from PIL import Image, ImageDraw, ImageFont
    n = min(4, len(recons))
    orig, gen = to_uint8_hwc(originals[:n]), to_uint8_hwc(recons[:n])
    img = Image.new("RGB", (2 * 128 + 10, n * 128 + (n - 1) * 10 + 30), color="white")
    draw = ImageDraw.Draw(img)
    font = ImageFont.load_default()
    for i in range(n):
        y = i * (128 + 10)
        img.paste(Image.fromarray(orig[i]), (0, y))
        img.paste(Image.fromarray(gen[i]), (128 + 10, y))
        text = "Generated"
        if quality_scores is not None:
            text += f" | Quality: {float(quality_scores[i].float().mean()):.3f}"
        if semantic_scores is not None:
            text += f" | Semantic: {float(semantic_scores[i].reshape(-1)[0]):.3f}"
        draw.text((0, y + 128 + 2), "Original", fill="black", font=font)
        draw.text((128 + 10, y + 128 + 2), text, fill="black", font=font)
    path.parent.mkdir(parents=True, exist_ok=True)
    img.save(path)
    return path


def save_samples(out_dir: Path, images: torch.Tensor, global_step: int) -> Iterable[Path]:
    """One PNG per decoded sample (train_hybrid.py:626-649; the decode itself is `LunarisCoreVAE.sample`)."""
    from PIL import Image
    stamp = time.strftime("%Y%m%d_%H%M%S")
    out_dir.mkdir(parents=True, exist_ok=True)
    paths = []
    for i, im in enumerate(to_uint8_hwc(images)):
        p = out_dir / f"sample_{global_step}_{i}_{stamp}.png"
        Image.fromarray(im).save(p)
        paths.append(p)
    return paths


# ---- torch.optim.AdamW state_dict <-> flat (exp_avg, exp_avg_sq, step) --------------------------------------------------
def adamw_state_dict(named_params: "OrderedDict[str, torch.Tensor]", exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                     offsets: Dict[str, int], step: int, lr: float, initial_lr: float, betas: Tuple[float, float], eps: float,
                     weight_decay: float, only: Optional[Iterable[str]] = None) -> dict:
    """The dictionary `torch.optim.AdamW(params).state_dict()` would hold after `step` updates, built from the flat moment
    buffers: per-parameter `step` / `exp_avg` / `exp_avg_sq` in parameter order.  `only`: names that have state (the
    reference's teacher optimizer only ever steps the gate / quality heads; the rest has no state entry)."""
    only = set(only) if only is not None else None
    state = {}
    for i, (name, p) in enumerate(named_params.items()):
        if only is not None and name not in only:
            continue
        if step == 0:
            continue
        o, n = offsets[name], p.numel()
        state[i] = {"step": torch.tensor(float(step)), "exp_avg": exp_avg[o:o + n].detach().cpu().reshape(p.shape).clone(),
                    "exp_avg_sq": exp_avg_sq[o:o + n].detach().cpu().reshape(p.shape).clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
             "initial_lr": initial_lr, "params": list(range(len(named_params)))}
    return {"state": state, "param_groups": [group]}


def load_adamw_state_dict(sd: dict, named_params: "OrderedDict[str, torch.Tensor]", exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                          offsets: Dict[str, int]) -> int:
    """Inverse of `adamw_state_dict`: fills the flat moment buffers, returns the step count (0 when the dict has no state)."""
    step = 0
    names = list(named_params.keys())
    for i, st in sd.get("state", {}).items():
        name = names[int(i)]
        p = named_params[name]
        o, n = offsets[name], p.numel()
        exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1).to(exp_avg.dtype))
        exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1).to(exp_avg_sq.dtype))
        step = max(step, int(float(st["step"])))
    return step


def scheduler_state_dict(t0: int, t_mult: int, eta_min: float, base_lr: float, steps: int, last_lr: float) -> dict:
    """`CosineAnnealingWarmRestarts.state_dict()` after `steps` calls of `.step()` (train_hybrid.py:516-527)."""
    t_i, t_cur = t0, steps
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= t_mult
    return {"T_0": t0, "T_i": t_i, "T_mult": t_mult, "eta_min": eta_min, "T_cur": t_cur, "base_lrs": [base_lr], "last_epoch": steps,
            "_step_count": steps + 1, "_get_lr_called_within_step": False, "_is_initial": False, "_last_lr": [last_lr]}


def flat_offsets(named_params: "OrderedDict[str, torch.Tensor]", flat: torch.Tensor) -> Dict[str, int]:
    """Element offset of every parameter inside the flat fp32 buffer it is a view of."""
    base = flat.data_ptr()
    return {k: (p.data_ptr() - base) // 4 for k, p in named_params.items()}


_DEFAULT_TEACHER: dict = {}


def _default_teacher(args):
    """(state_dict, named parameters, initial lr) of a default-initialised CPU teacher for the run's flags; built once per shape
    without touching the global RNG stream."""
    a = args if isinstance(args, dict) else {}
    key = (int(a.get("num_experts", 4)), int(a.get("feature_dim", 128)), int(a.get("embedding_dim", 64)))
    if key not in _DEFAULT_TEACHER:
        from .teacher import LunarMoETeacher
        # fork_rng() with its default device list saves and restores the CPU generator AND every CUDA generator: manual_seed
        # below reseeds all of them (with devices=[] the CUDA streams of the run were reseeded for good)
        with torch.random.fork_rng():
            torch.manual_seed(int(a.get("seed", 42)))
            m = LunarMoETeacher(num_experts=key[0], feature_dim=key[1], embedding_dim=key[2])
        _DEFAULT_TEACHER[key] = ({k: v.detach().clone() for k, v in m.state_dict().items()},
                                 OrderedDict((k, p.detach()) for k, p in m.named_parameters()))
    sd, tp = _DEFAULT_TEACHER[key]
    return sd, tp, float(a.get("teacher_lr", 1e-4))


def prune_periodic_checkpoints(ckpt_dir: Path, keep: int) -> list:
    """`--keep_n_checkpoints` (train_hybrid.py:1115, parsed and never read by the reference): of the periodic `step_<N>.pt` files
    written under `--save_every`, keep the `keep` with the highest step numbers; returns the paths removed.  `latest.pt` /
    `best.pt` are never touched."""
    found = []
    for p in Path(ckpt_dir).glob("step_*.pt"):
        try:
            found.append((int(p.stem.split("_", 1)[1]), p))
        except ValueError:
            continue
    found.sort()
    removed = []
    for _, p in found[:max(0, len(found) - max(0, int(keep)))]:
        p.unlink(missing_ok=True)
        removed.append(p)
    return removed


def checkpoint_dict(stepper, vae, teacher, global_step: int, best_loss: float, args: dict) -> dict:
    """The reference's checkpoint dictionary (train_hybrid.py:594-605) from the native state.  Extra key
    `lunaris_amd_extra` (ignored by the reference) keeps what the reference loses on resume: the reward baseline and the
    fp16 loss scale."""
    vp = OrderedDict(vae.named_parameters())
    ck = {"global_step": global_step,
          "vae_state_dict": {k: v.detach().cpu().clone() for k, v in vae.state_dict().items()},
          "teacher_state_dict": {}, "teacher_optimizer": {"state": {}, "param_groups": []}, "teacher_scheduler": {},
          "best_loss": best_loss, "args": args}
    ck["vae_optimizer"] = adamw_state_dict(vp, stepper.exp_avg, stepper.exp_avg_sq, flat_offsets(vp, vae.flat_parameters()),
                                           stepper.opt_steps, stepper.lr, stepper.base_lr, stepper.betas, stepper.eps, stepper.weight_decay)
    ck["vae_scheduler"] = scheduler_state_dict(stepper.t0, 2, stepper.min_lr, stepper.base_lr, stepper.opt_steps, stepper.lr)
    extra = {}
    if teacher is not None:
        ck["teacher_state_dict"] = {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()}
        tp = OrderedDict(teacher.named_parameters())
        from .trainer import cosine_warm_restarts_lr
        t_lr = cosine_warm_restarts_lr(stepper.teacher_base_lr, stepper.min_lr, stepper.t0, 2, stepper.opt_steps)
        if getattr(stepper, "_t_ready", False):
            b, e = stepper.t_range
            offs = flat_offsets(tp, teacher._flat)
            live = [k for k, o in offs.items() if b <= o < e]
            rel = {k: offs[k] - b for k in live}
            ck["teacher_optimizer"] = adamw_state_dict(tp, stepper.t_m, stepper.t_v, {**{k: 0 for k in tp}, **rel}, stepper.opt_steps, t_lr,
                                                       stepper.teacher_base_lr, stepper.betas, stepper.eps, stepper.weight_decay, only=live)
        else:
            ck["teacher_optimizer"] = adamw_state_dict(tp, torch.zeros(1), torch.zeros(1), {k: 0 for k in tp}, 0, t_lr, stepper.teacher_base_lr,
                                                       stepper.betas, stepper.eps, stepper.weight_decay)
        ck["teacher_scheduler"] = scheduler_state_dict(stepper.t0, 2, stepper.min_lr, stepper.teacher_base_lr, stepper.opt_steps, t_lr)
        extra["reward_state"] = stepper.reward_state.detach().cpu().clone()
    else:
        # VAE-only run (teacher never built): the reference's `_load_checkpoint` still calls `teacher_optimizer.load_state_dict`
        # and `teacher_scheduler.load_state_dict` (train_hybrid.py:808-822) and gives up on a dict without parameter groups, so
        # write the state of a freshly constructed teacher: default-initialised weights, one parameter group over all of its
        # parameters with no moments yet, the scheduler at step 0
        try:
            t0_, tp, t_lr = _default_teacher(args)
        except NotImplementedError as e:
            # a VAE-only run never builds the teacher, so its flags were never validated (e.g. --feature_dim 64): the run must not
            # die at its first save.  Keep the empty teacher entries; the reference cannot resume its teacher from
            # this file, the VAE half loads either side
            import warnings
            warnings.warn(f"checkpoint written without teacher entries: {e}")
            extra["teacher_untrained"] = True
            extra["teacher_entries_omitted"] = str(e)
        else:
            ck["teacher_state_dict"] = t0_
            ck["teacher_optimizer"] = adamw_state_dict(tp, torch.zeros(1), torch.zeros(1), {k: 0 for k in tp}, 0, getattr(stepper, "teacher_base_lr", 1e-4),
                                                       float(args.get("teacher_lr", 1e-4)) if isinstance(args, dict) else 1e-4, stepper.betas, stepper.eps,
                                                       stepper.weight_decay)
            ck["teacher_scheduler"] = scheduler_state_dict(stepper.t0, 2, stepper.min_lr, float(args.get("teacher_lr", 1e-4)) if isinstance(args, dict) else 1e-4, 0, t_lr)
            extra["teacher_untrained"] = True
    extra["loss_scale"] = float(vae.loss_scale)          # the reference's GradScaler state is not checkpointed either side: keep ours
    # positions in the two counter-RNG streams (reparameterisation noise, teacher dropout masks): calls drawn so far.  Each stream
    # is keyed by (torch seed, rank), so on resume every rank re-derives its own stream and advances it by the recorded count
    # (vae.lcg_advance) -- the run continues with the noise / masks it would have drawn uninterrupted (the reference does not
    # checkpoint its RNG state: SURVEY §5)
    extra["noise_calls"] = int(getattr(vae, "noise_calls", 0))
    if teacher is not None:
        extra["drop_calls"] = int(getattr(teacher, "drop_calls", 0))
    ck["lunaris_amd_extra"] = extra
    return ck


def restore_checkpoint(ck: dict, stepper, vae, teacher) -> Tuple[int, float]:
    """Load a checkpoint written by `checkpoint_dict` OR by the reference's `_save_checkpoint`.  Returns
    (global_step, best_loss).  Models first (strict=False like train_hybrid.py:798-803), then moments and step counts."""
    vae.load_state_dict(ck["vae_state_dict"], strict=False)
    if teacher is not None and ck.get("teacher_state_dict"):
        teacher.load_state_dict(ck["teacher_state_dict"], strict=False)
    vp = OrderedDict(vae.named_parameters())
    flat = vae.flat_parameters()
    opt = ck.get("vae_optimizer") or {}
    if opt.get("state"):
        stepper.opt_steps = load_adamw_state_dict(opt, vp, stepper.exp_avg, stepper.exp_avg_sq, flat_offsets(vp, flat))
    elif ck.get("vae_scheduler", {}).get("last_epoch") is not None:
        stepper.opt_steps = int(ck["vae_scheduler"]["last_epoch"])
    if teacher is not None and (ck.get("teacher_optimizer") or {}).get("state"):
        stepper._pending_teacher_opt = ck["teacher_optimizer"]          # applied once the teacher buffers exist (first step)
    extra = ck.get("lunaris_amd_extra") or {}
    if "loss_scale" in extra:
        vae.loss_scale = float(extra["loss_scale"])
    if teacher is not None and "reward_state" in extra:
        stepper.reward_state.copy_(extra["reward_state"])
    if "noise_calls" in extra:
        vae._seed, vae.noise_calls = None, int(extra["noise_calls"])          # re-derived (and advanced) at the next forward
    if teacher is not None and "drop_calls" in extra:
        teacher._drop_seed, teacher.drop_calls = None, int(extra["drop_calls"])
    return int(ck.get("global_step", 0)), float(ck.get("best_loss", float("inf")))
