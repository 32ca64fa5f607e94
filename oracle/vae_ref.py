"""CPU oracle for the Lunaris-Orion VAE training step.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) *restatement* of the arithmetic of the reference's
hot path.  It is the checker the HIP path is compared with; it is never the thing measured or
shipped.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package (``lunaris_orion_amd``) must never import anything from
``oracle/`` and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own model classes from
``/root/reference`` (in the build container only), checks this restatement against them, and
emits the small fixtures under ``tests/golden/`` that ``tests/test_oracle_golden.py`` re-checks on
every run (the reference itself never travels to the GPU box).

Reference lines restated here (paths relative to /root/reference):
  mish                        lunar_generate.py:24-26
  ResBlock.forward            lunar_generate.py:49-53   (conv-GN-Mish x2, identity shortcut, mish)
  Encoder.forward             lunar_generate.py:127-153
  Decoder.forward             lunar_generate.py:194-229
  reparameterize              lunar_generate.py:248-261 (eps is an explicit argument here)
  LunarisCoreVAE.forward      lunar_generate.py:263-276
  losses                      train_hybrid.py:859 (MSE), :862 (KL), :886-889 (vae_loss)
  clip_grad_norm_ + AdamW     train_hybrid.py:913, :921, :504-509
  CosineAnnealingWarmRestarts train_hybrid.py:516-521, :925 (stepped once per optimizer step)
  input normalisation         train_hybrid.py:181-182
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

GN_GROUPS = 8
GN_EPS = 1e-5

# (name, in_ch, out_ch) of the four encoder stages, lunar_generate.py:94-120
ENC_STAGES = (("down1", 3, 64), ("down2", 64, 128), ("down3", 128, 256), ("down4", 256, 512))
# (name, in_ch, out_ch) of the four decoder stages, lunar_generate.py:168-190
DEC_STAGES = (("up1", 512, 256), ("up2", 256, 128), ("up3", 128, 64), ("up4", 64, 32))


def param_shapes(latent_dim: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """The 72 parameter tensors of LunarisCoreVAE in ``state_dict`` order (lunar_generate.py:91-125,162-192)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for name, cin, cout in ENC_STAGES:
        p = f"encoder.{name}"
        s[f"{p}.0.weight"] = (cout, cin, 3, 3)
        s[f"{p}.0.bias"] = (cout,)
        s[f"{p}.1.weight"] = (cout,)
        s[f"{p}.1.bias"] = (cout,)
        for cv in ("conv1", "conv2"):
            s[f"{p}.3.{cv}.0.weight"] = (cout, cout, 3, 3)
            s[f"{p}.3.{cv}.0.bias"] = (cout,)
            s[f"{p}.3.{cv}.1.weight"] = (cout,)
            s[f"{p}.3.{cv}.1.bias"] = (cout,)
    s["encoder.fc_mu.weight"] = (latent_dim, 512 * 8 * 8)
    s["encoder.fc_mu.bias"] = (latent_dim,)
    s["encoder.fc_logvar.weight"] = (latent_dim, 512 * 8 * 8)
    s["encoder.fc_logvar.bias"] = (latent_dim,)
    s["decoder.fc.weight"] = (512 * 8 * 8, latent_dim)
    s["decoder.fc.bias"] = (512 * 8 * 8,)
    for name, cin, cout in DEC_STAGES:
        p = f"decoder.{name}"
        s[f"{p}.0.weight"] = (cin, cout, 4, 4)  # ConvTranspose2d layout [C_in, C_out, kH, kW]
        s[f"{p}.0.bias"] = (cout,)
        s[f"{p}.1.weight"] = (cout,)
        s[f"{p}.1.bias"] = (cout,)
    s["decoder.final_conv.weight"] = (3, 32, 3, 3)
    s["decoder.final_conv.bias"] = (3,)
    return s


def closed_form_uniform(name: str, n: int, salt: int = 0) -> torch.Tensor:
    """n float64 values in [-1, 1): an integer hash of (name, salt, index).  Integer arithmetic only
    (int64 with wraparound) up to the final division, so it is bit-identical on every platform."""
    h0 = 1469598103934665603
    for ch in (name + f"#{salt}").encode():
        h0 = ((h0 ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    idx = torch.arange(n, dtype=torch.int64)
    x = idx * -7046029254386353131 + (h0 - (1 << 64) if h0 >= (1 << 63) else h0)
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * -4658895280553007687
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * -7723592293110705685
    x = x ^ ((x >> 31) & 0x1FFFFFFFF)
    u = ((x >> 11) & ((1 << 53) - 1)).to(torch.float64) / float(1 << 53)  # [0,1)
    return u * 2.0 - 1.0


def closed_form_tensor(name: str, shape: Tuple[int, ...], salt: int = 0) -> torch.Tensor:
    """Deterministic, platform-independent pseudo-random parameter fill used by fixtures and tests.

    143 MB of random weights cannot be committed, so fixtures use weights that any party can
    regenerate exactly: ``closed_form_uniform`` scaled like PyTorch's default initialisers (bound
    1/sqrt(fan_in)); GroupNorm weight ~ 1 +- 0.25; biases small.
    """
    n = 1
    for d in shape:
        n *= d
    u = closed_form_uniform(name, n, salt)
    if len(shape) == 1:
        t = 1.0 + 0.25 * u if name.endswith(".1.weight") else 0.05 * u
    else:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        t = u / math.sqrt(fan_in)
    return t.to(torch.float32).reshape(shape)


_PARAM_CACHE: "dict" = {}


def closed_form_params(latent_dim: int, salt: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """The closed-form parameter set (61 M values at latent 512: 1.3-1.9 s of hashing).  A test run asks for the same (latent_dim, salt)
    dozens of times: the generated tensors are kept and every call gets its own copies (0.1 s), so a caller may still update them in place."""
    key = (int(latent_dim), int(salt))
    if key not in _PARAM_CACHE:
        if len(_PARAM_CACHE) >= 4:
            _PARAM_CACHE.clear()
        _PARAM_CACHE[key] = OrderedDict((k, closed_form_tensor(k, shp, salt)) for k, shp in param_shapes(latent_dim).items())
    return OrderedDict((k, v.clone()) for k, v in _PARAM_CACHE[key].items())


def closed_form_sprites(n: int, salt: int = 0) -> torch.Tensor:
    """uint8 sprites [n,128,128,3] (HWC, like sprites_*.npy) made of flat-colour 8x8 blocks + a ramp."""
    yy = torch.arange(128).view(1, 128, 1, 1)
    xx = torch.arange(128).view(1, 1, 128, 1)
    cc = torch.arange(3).view(1, 1, 1, 3)
    nn_ = torch.arange(n).view(n, 1, 1, 1) + salt
    block = ((yy // 8) * 7 + (xx // 8) * 13 + cc * 29 + nn_ * 31) * 37
    ramp = (yy * 3 + xx * 5 + cc * 11 + nn_ * 17)
    v = (block + (ramp // 4)) % 256
    return v.to(torch.uint8)


def normalise_sprites(u8_hwc: torch.Tensor) -> torch.Tensor:
    """train_hybrid.py:181-182: uint8 HWC -> float32 / 127.5 - 1, CHW."""
    return (u8_hwc.to(torch.float32) / 127.5 - 1.0).permute(0, 3, 1, 2).contiguous()


def closed_form_eps(batch: int, latent_dim: int, salt: int = 0) -> torch.Tensor:
    """Explicit N(0,1)-like noise for the reparameterisation (Box-Muller over the closed-form hash)."""
    u1 = closed_form_uniform(f"eps.u1.{salt}", batch * latent_dim)
    u2 = closed_form_uniform(f"eps.u2.{salt}", batch * latent_dim)
    u1 = (u1 + 1.0) * 0.5 * (1 - 1e-9) + 1e-9
    u2 = (u2 + 1.0) * 0.5
    z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
    return z.to(torch.float32).reshape(batch, latent_dim)


# ----------------------------------------------------------------------------------------------
# forward restatement
# ----------------------------------------------------------------------------------------------

def mish(x: torch.Tensor) -> torch.Tensor:
    """lunar_generate.py:24-26 (x * tanh(softplus(x)), softplus threshold 20)."""
    return x * torch.tanh(F.softplus(x))


def _conv_gn_mish(x, P, prefix_conv, prefix_gn, stride, acts=None, tag=None, transposed=False):
    if transposed:
        v = F.conv_transpose2d(x, P[prefix_conv + ".weight"], P[prefix_conv + ".bias"], stride=2, padding=1)
    else:
        v = F.conv2d(x, P[prefix_conv + ".weight"], P[prefix_conv + ".bias"], stride=stride, padding=1)
    u = F.group_norm(v, GN_GROUPS, P[prefix_gn + ".weight"], P[prefix_gn + ".bias"], GN_EPS)
    a = F.mish(u)  # nn.Mish
    if acts is not None:
        acts[tag + ".conv"] = v
        acts[tag + ".gn"] = u
        acts[tag + ".act"] = a
    return a


def res_block(x, P, prefix, acts=None):
    """lunar_generate.py:49-53."""
    out = _conv_gn_mish(x, P, prefix + ".conv1.0", prefix + ".conv1.1", 1, acts, prefix + ".conv1")
    out = _conv_gn_mish(out, P, prefix + ".conv2.0", prefix + ".conv2.1", 1, acts, prefix + ".conv2")
    y = mish(out + x)
    if acts is not None:
        acts[prefix + ".out"] = y
    return y


def encoder_forward(x, P, acts=None):
    """lunar_generate.py:127-153."""
    skips = []
    h = x
    for i, (name, _cin, _cout) in enumerate(ENC_STAGES):
        p = f"encoder.{name}"
        h = _conv_gn_mish(h, P, p + ".0", p + ".1", 2, acts, p + ".0")
        h = res_block(h, P, p + ".3", acts)
        if i < 3:
            skips.append(h)
    flat = h.flatten(1)
    mu = F.linear(flat, P["encoder.fc_mu.weight"], P["encoder.fc_mu.bias"])
    logvar = F.linear(flat, P["encoder.fc_logvar.weight"], P["encoder.fc_logvar.bias"])
    return mu, logvar, skips


def decoder_forward(z, skips, P, acts=None):
    """lunar_generate.py:194-229."""
    B = z.shape[0]
    h = F.linear(z, P["decoder.fc.weight"], P["decoder.fc.bias"]).view(B, 512, 8, 8)
    if acts is not None:
        acts["decoder.fc"] = h
    for i, (name, _cin, _cout) in enumerate(DEC_STAGES):
        p = f"decoder.{name}"
        h = _conv_gn_mish(h, P, p + ".0", p + ".1", 2, acts, p + ".0", transposed=True)
        need = 3 - i  # up1 needs len>=3 and uses skips[2], up2 -> skips[1], up3 -> skips[0]
        if i < 3 and len(skips) >= need:
            h = h + skips[need - 1]
    v = F.conv2d(h, P["decoder.final_conv.weight"], P["decoder.final_conv.bias"], padding=1)
    if acts is not None:
        acts["decoder.final_conv"] = v
    return torch.tanh(v)


def vae_forward(x, eps, P, acts=None):
    """lunar_generate.py:263-276 with the noise of :260 passed in explicitly."""
    mu, logvar, skips = encoder_forward(x, P, acts)
    std = torch.exp(0.5 * logvar)
    z = mu + eps * std
    recon = decoder_forward(z, skips, P, acts)
    return recon, mu, logvar


def vae_losses(recon, x, mu, logvar):
    """train_hybrid.py:859, :862."""
    recon_loss = F.mse_loss(recon, x, reduction="mean")
    kl_loss = -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
    return recon_loss, kl_loss


def vae_objective(recon_loss, kl_loss, recon_weight=1.0, kl_weight=0.1, mean_advantage=0.0, accum=1):
    """train_hybrid.py:886-889, :895 — pg_loss = -(advantage*recon_loss).mean() = -mean(adv)*recon_loss."""
    pg_loss = -mean_advantage * recon_loss
    vae_loss = (recon_weight * recon_loss + kl_weight * kl_loss + pg_loss) / accum
    return vae_loss, pg_loss


# ----------------------------------------------------------------------------------------------
# optimizer restatement (train_hybrid.py:504-521, 913, 921, 925)
# ----------------------------------------------------------------------------------------------

def clip_coef(grads: List[torch.Tensor], max_norm: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch.nn.utils.clip_grad_norm_: total L2 norm, coef = max_norm/(norm+1e-6) clamped to <=1."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, coef


def adamw_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
    """torch.optim.AdamW single-tensor update (decoupled weight decay), in place.  ``step`` is 1-based."""
    p.mul_(1.0 - lr * weight_decay)
    m.lerp_(g, 1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)


def cosine_warm_restarts_lr(base_lr: float, eta_min: float, t0: int, t_mult: int, epoch: int) -> float:
    """LR after ``epoch`` calls of CosineAnnealingWarmRestarts.step() (epoch 0 = initial LR)."""
    t_i, t_cur = t0, epoch
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= t_mult
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t_cur / t_i)) / 2.0


class OracleTrainer:
    """VAE-only restatement of TrainingManager._process_batch (train_hybrid.py:838-954).

    The teacher contributes one detached scalar, ``mean_advantage``; with
    ``--reward_scale 0`` it is exactly 0.  Everything else of the step is here: forward, losses,
    backward, clip, AdamW, scheduler.
    """

    def __init__(self, params: Dict[str, torch.Tensor], lr=1e-4, min_lr=1e-6, t0=10, weight_decay=0.01,
                 max_grad_norm=1.0, recon_weight=1.0, kl_weight=0.1, accum=1):
        self.P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        self.base_lr, self.min_lr, self.t0 = lr, min_lr, t0
        self.weight_decay, self.max_grad_norm = weight_decay, max_grad_norm
        self.recon_weight, self.kl_weight, self.accum = recon_weight, kl_weight, accum
        self.opt_steps = 0

    def lr(self) -> float:
        return cosine_warm_restarts_lr(self.base_lr, self.min_lr, self.t0, 2, self.opt_steps)

    def step(self, x, eps, mean_advantage: float = 0.0, do_update: bool = True):
        for p in self.P.values():
            p.grad = None
        recon, mu, logvar = vae_forward(x, eps, self.P)
        recon_loss, kl_loss = vae_losses(recon, x, mu, logvar)
        vae_loss, pg_loss = vae_objective(recon_loss, kl_loss, self.recon_weight, self.kl_weight,
                                          mean_advantage, self.accum)
        vae_loss.backward()
        grads = [p.grad for p in self.P.values()]
        total, coef = clip_coef(grads, self.max_grad_norm)
        out = dict(recon=recon.detach(), mu=mu.detach(), logvar=logvar.detach(),
                   recon_loss=float(recon_loss.detach()), kl_loss=float(kl_loss.detach()), vae_loss=float(vae_loss.detach()),
                   pg_loss=float(pg_loss.detach()), grad_norm=float(total), lr=self.lr(),
                   grads=OrderedDict((k, p.grad.detach().clone()) for k, p in self.P.items()))
        if do_update:
            lr = self.lr()
            self.opt_steps += 1
            with torch.no_grad():
                for k, p in self.P.items():
                    g = p.grad * coef
                    adamw_step(p, g, self.m[k], self.v[k], self.opt_steps, lr, weight_decay=self.weight_decay)
        return out


def self_attention_2d(x, wq, bq, wk, bk, wv, bv, gamma):
    """SelfAttention2d.forward, lunar_generate.py:68-78 (module is defined but never instantiated)."""
    B, C, H, W = x.shape
    q = F.conv2d(x, wq, bq).view(B, -1, H * W)
    k = F.conv2d(x, wk, bk).view(B, -1, H * W)
    energy = torch.bmm(q.permute(0, 2, 1), k)
    att = F.softmax(energy, dim=-1)
    v = F.conv2d(x, wv, bv).view(B, -1, H * W)
    out = torch.bmm(v, att.permute(0, 2, 1)).view(B, C, H, W)
    return gamma * out + x
