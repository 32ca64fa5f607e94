"""CPU oracle for LunarMoETeacher.forward AS EXECUTED by the reference.  TEST INFRASTRUCTURE ONLY (see vae_ref.py).

Plain-PyTorch fp32 restatement of /root/reference/lunar_evaluator.py:57-462 on a flat dict of the reference's
``state_dict`` tensors (252 parameters + BatchNorm buffers).  The quirks of the reference are reproduced, not fixed
(SURVEY §3.4):
  * PixelArtAttention (:188-227): tokens are split into chunks of 32, a chunk attends only to itself, the relative
    position term is constant along the key axis (softmax-invariant), and chunk outputs are written at offset = chunk
    INDEX, so the final map holds chunk_p.row0 at p <= 511, chunk_511.rows 1..31 at p = 512..542 and zeros elsewhere;
  * ``prompt_embedding`` passed by the caller is overwritten before use (:438), cosine similarity with itself is 1;
  * BatchNorm2d runs with batch statistics in training mode (and updates its running stats, returned separately).
Dropout is stochastic in the reference; the oracle applies it at the reference's six sites (:97-99,108; :139-140,212,225;
:246,253; :353-397) from an explicit mask provider (``oracle/dropout_ref.TeacherMasks``: the masks the native library draws
for a given call seed), or not at all (``masks=None`` = ``dropout_rate 0``).  Parity status: PINNED by oracle/make_golden.py
(fixtures tests/golden/teacher_*.npz; the dropout fixture runs the reference itself on the same injected masks).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5


def teacher_param_shapes(num_experts=4, feature_dim=128, embedding_dim=64, expert_layers=3, intermediate_dim=256,
                         rel_pos_size=8) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameters AND buffers in the reference's state_dict order (lunar_evaluator.py:57-112,119-144,234-258,291-397)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(p, co, ci, k, groups=1):
        s[p + ".weight"] = (co, ci // groups, k, k)
        s[p + ".bias"] = (co,)

    def bn(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)
        s[p + ".running_mean"] = (c,)
        s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    def lin(p, o, i):
        s[p + ".weight"] = (o, i)
        s[p + ".bias"] = (o,)

    fe = "feature_extractor"
    conv(fe + ".conv1.0", 32, 3, 3)
    bn(fe + ".conv1.2", 32)
    for br, k in (("edge_branch", 3), ("color_branch", 5), ("detail_branch", 3)):
        conv(f"{fe}.{br}.0", 32, 32, k, groups=32)
        conv(f"{fe}.{br}.1", 64, 32, 1)
        bn(f"{fe}.{br}.3", 64)
    conv(fe + ".fusion.0", 128, 192, 1)
    bn(fe + ".fusion.2", 128)
    for e in range(num_experts):
        cin = 128
        for l in range(expert_layers):
            p = f"experts.{e}.{l}"
            s[p + ".layer_scale"] = (1, feature_dim, 1, 1)
            conv(p + ".conv1.0", feature_dim, cin, 3)
            bn(p + ".conv1.2", feature_dim)
            s[p + ".attention.rel_pos_h"] = (1, 8, rel_pos_size, 1)
            s[p + ".attention.rel_pos_w"] = (1, 8, 1, rel_pos_size)
            s[p + ".attention.last_spatial_shapes"] = (2,)
            conv(p + ".attention.qkv", 3 * feature_dim, feature_dim, 1)
            conv(p + ".attention.proj", feature_dim, feature_dim, 1)
            conv(p + ".conv2.0", feature_dim, feature_dim, 3)
            bn(p + ".conv2.2", feature_dim)
            if cin != feature_dim:
                conv(p + ".shortcut.0", feature_dim, cin, 1)
                bn(p + ".shortcut.1", feature_dim)
            cin = feature_dim
    lin("gate.2", intermediate_dim, 128)
    lin("gate.5", num_experts, intermediate_dim)
    for e in range(num_experts):
        p = f"quality_heads.{e}"
        s[p + ".2.weight"] = (feature_dim,)
        s[p + ".2.bias"] = (feature_dim,)
        lin(p + ".3", intermediate_dim // 4, feature_dim)
        lin(p + ".6", 4, intermediate_dim // 4)
    for name, out in (("semantic_head", 1), ("style_net", embedding_dim), ("prompt_net", embedding_dim)):
        s[name + ".2.weight"] = (feature_dim,)
        s[name + ".2.bias"] = (feature_dim,)
        lin(name + ".3", intermediate_dim // 2, feature_dim)
        lin(name + ".6", out, intermediate_dim // 2)
    return s


def closed_form_teacher_state(salt: int = 0, **kw) -> "OrderedDict[str, torch.Tensor]":
    """Closed-form state (see vae_ref.closed_form_tensor): weights ~ U(-1,1)/sqrt(fan_in), BN/LN weights ~ 1 +- 0.25,
    running_mean small, running_var in [0.75, 1.25], layer_scale 0.1 +- 0.025."""
    from .vae_ref import closed_form_uniform
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in teacher_param_shapes(**kw).items():
        n = 1
        for d in shp:
            n *= d
        u = closed_form_uniform("teacher." + k, max(n, 1), salt)
        if k.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.int64)
        elif k.endswith("last_spatial_shapes"):
            t = torch.zeros(2)
        elif k.endswith("running_var"):
            t = (1.0 + 0.25 * u).float().reshape(shp)
        elif k.endswith("running_mean"):
            t = (0.05 * u).float().reshape(shp)
        elif k.endswith("layer_scale"):
            t = (0.1 + 0.025 * u).float().reshape(shp)
        elif len(shp) == 1:
            is_norm_w = k.endswith(".weight") and (".conv1.2." in k or ".conv2.2." in k or ".3.weight" in k and "branch" in k
                                                   or "fusion.2" in k or "shortcut.1" in k or k.split(".")[-2] == "2")
            t = ((1.0 + 0.25 * u) if is_norm_w else 0.05 * u).float().reshape(shp)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = (u / (fan_in ** 0.5)).float().reshape(shp)
        out[k] = t
    return out


def _bn(x, S, p, training, new_stats):
    w, b, rm, rv = S[p + ".weight"], S[p + ".bias"], S[p + ".running_mean"], S[p + ".running_var"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() / x.shape[1]
        new_stats[p + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
        new_stats[p + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / (n - 1)
    else:
        mean, var = rm, rv
    return (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS) * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def feature_extractor(x, S, training, new_stats, masks=None):
    """lunar_evaluator.py:105-112."""
    p = "feature_extractor"
    h = _bn(F.leaky_relu(F.conv2d(x, S[p + ".conv1.0.weight"], S[p + ".conv1.0.bias"], padding=1), 0.2), S, p + ".conv1.2", training, new_stats)
    outs = []
    for br, pad in (("edge_branch", 1), ("color_branch", 2), ("detail_branch", 1)):
        q = f"{p}.{br}"
        t = F.conv2d(h, S[q + ".0.weight"], S[q + ".0.bias"], padding=pad, groups=32)
        t = F.conv2d(t, S[q + ".1.weight"], S[q + ".1.bias"])
        outs.append(_bn(F.leaky_relu(t, 0.2), S, q + ".3", training, new_stats))
    c = torch.cat(outs, dim=1)
    if masks is not None:
        from .dropout_ref import DS_FE
        c = c * masks.elementwise_nchw(DS_FE, 192)                       # self.dropout(combined), :108
    f = F.conv2d(c, S[p + ".fusion.0.weight"], S[p + ".fusion.0.bias"])
    return _bn(F.leaky_relu(f, 0.2), S, p + ".fusion.2", training, new_stats)


def attention_as_executed(x, S, p, num_heads=8, chunk=32, att_mask=None, proj_mask=None):
    """PixelArtAttention.forward (lunar_evaluator.py:188-227) in closed form (see module docstring).  att_mask: [B, 543,
    heads, chunk] multiplicative attn_drop mask of the rows that reach the output; proj_mask: [B, C, H, W] (proj_drop)."""
    B, C, H, W = x.shape
    N, hd = H * W, C // num_heads
    qkv = F.conv2d(x, S[p + ".qkv.weight"], S[p + ".qkv.bias"])
    qkv = qkv.reshape(B, 3, num_heads, hd, N).permute(0, 1, 2, 4, 3)      # [B,3,heads,N,hd]
    nchunk = (N + chunk - 1) // chunk
    q = qkv[:, 0].reshape(B, num_heads, nchunk, chunk, hd)
    k = qkv[:, 1].reshape(B, num_heads, nchunk, chunk, hd)
    v = qkv[:, 2].reshape(B, num_heads, nchunk, chunk, hd)
    att = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)   # rel-pos term: softmax-invariant
    if att_mask is not None:                                                           # attn_drop (:212) on the surviving rows
        att = att.clone()
        att[:, :, :, 0, :] = att[:, :, :, 0, :] * att_mask[:, :nchunk].permute(0, 2, 1, 3)
        att[:, :, nchunk - 1, 1:, :] = att[:, :, nchunk - 1, 1:, :] * att_mask[:, nchunk:].permute(0, 2, 1, 3)
    co = torch.matmul(att, v)                                                          # [B,heads,nchunk,chunk,hd]
    out = torch.zeros(B, num_heads, N, hd, dtype=x.dtype, device=x.device)
    out[:, :, :nchunk] = co[:, :, :, 0]                      # p <= nchunk-1 : row 0 of chunk p
    out[:, :, nchunk:nchunk + chunk - 1] = co[:, :, nchunk - 1, 1:]   # rows 1..31 of the last chunk
    out = out.permute(0, 1, 3, 2).reshape(B, C, H, W)
    out = F.conv2d(out, S[p + ".proj.weight"], S[p + ".proj.bias"])
    return out if proj_mask is None else out * proj_mask                               # proj_drop (:225)


def expert_block(x, S, p, training, new_stats, masks=None, e=0, l=0):
    """ExpertBlock.forward (lunar_evaluator.py:260-275)."""
    m1 = am = pm = m2 = None
    if masks is not None:
        from .dropout_ref import ds_block
        C = S[p + ".conv1.0.weight"].shape[0]
        m1, m2 = masks.channelwise(ds_block(e, l, 0), C), masks.channelwise(ds_block(e, l, 3), C)
        am, pm = masks.attention(ds_block(e, l, 1)), masks.elementwise_nchw(ds_block(e, l, 2), C)
    if (p + ".shortcut.0.weight") in S:
        idt = _bn(F.conv2d(x, S[p + ".shortcut.0.weight"], S[p + ".shortcut.0.bias"]), S, p + ".shortcut.1", training, new_stats)
    else:
        idt = x
    o = _bn(F.leaky_relu(F.conv2d(x, S[p + ".conv1.0.weight"], S[p + ".conv1.0.bias"], padding=1), 0.2), S, p + ".conv1.2", training, new_stats)
    if m1 is not None:
        o = o * m1                                                                     # Dropout2d (:246)
    o = attention_as_executed(o, S, p + ".attention", att_mask=am, proj_mask=pm)
    o = _bn(F.leaky_relu(F.conv2d(o, S[p + ".conv2.0.weight"], S[p + ".conv2.0.bias"], padding=1), 0.2), S, p + ".conv2.2", training, new_stats)
    if m2 is not None:
        o = o * m2                                                                     # Dropout2d (:253)
    return F.leaky_relu(o * S[p + ".layer_scale"] + idt, 0.2)


def _head(pooled, S, p, final=None, mask=None):
    """AdaptiveAvgPool -> Flatten -> LayerNorm -> Linear -> LeakyReLU -> Dropout -> Linear [-> Sigmoid]."""
    h = F.layer_norm(pooled, (pooled.shape[1],), S[p + ".2.weight"], S[p + ".2.bias"], LN_EPS)
    h = F.leaky_relu(F.linear(h, S[p + ".3.weight"], S[p + ".3.bias"]), 0.2)
    if mask is not None:
        h = h * mask
    h = F.linear(h, S[p + ".6.weight"], S[p + ".6.bias"])
    return torch.sigmoid(h) if final == "sigmoid" else h


def teacher_forward(x, S: Dict[str, torch.Tensor], training: bool = True, num_experts=4, expert_layers=3, masks=None):
    """LunarMoETeacher.forward (lunar_evaluator.py:408-462).  masks: dropout_ref.TeacherMasks (train mode with dropout) or
    None (no dropout).  Returns (outputs, new BN running stats)."""
    from . import dropout_ref as D
    if not training:
        masks = None
    hm = (lambda site, width: masks.rows(site, width)) if masks is not None else (lambda site, width: None)
    new_stats: Dict[str, torch.Tensor] = {}
    feats = feature_extractor(x, S, training, new_stats, masks)
    pooled = feats.mean(dim=(2, 3))
    g = F.leaky_relu(F.linear(pooled, S["gate.2.weight"], S["gate.2.bias"]), 0.2)
    if masks is not None:
        g = g * masks.rows(D.DS_GATE, g.shape[1])                                      # gate Dropout (:358)
    w = torch.softmax(F.linear(g, S["gate.5.weight"], S["gate.5.bias"]), dim=1)
    q_all, pooled_e = [], []
    sem_feat = None
    for e in range(num_experts):
        h = feats
        for l in range(expert_layers):
            h = expert_block(h, S, f"experts.{e}.{l}", training, new_stats, masks, e, l)
        pe = h.mean(dim=(2, 3))
        pooled_e.append(pe)
        q_all.append(_head(pe, S, f"quality_heads.{e}", mask=hm(D.ds_quality(e), S[f"quality_heads.{e}.3.weight"].shape[0])))
        if e == 0:
            sem_feat = pe
    qt = torch.stack(q_all, dim=1)
    weighted_q = (qt * w.unsqueeze(-1)).sum(dim=1)
    comb = (torch.stack(pooled_e, dim=1) * w.unsqueeze(-1)).sum(dim=1)
    style = _head(comb, S, "style_net", mask=hm(D.DS_STYLE, S["style_net.3.weight"].shape[0]))
    prompt = _head(comb, S, "prompt_net", mask=hm(D.DS_PROMPT, S["prompt_net.3.weight"].shape[0]))
    sem = _head(sem_feat, S, "semantic_head", final="sigmoid", mask=hm(D.DS_SEM, S["semantic_head.3.weight"].shape[0])) * 1.0     # cosine_similarity(p, p.detach()) == 1
    out = {"quality_scores": torch.sigmoid(weighted_q), "expert_weights": w, "style_embedding": style,
           "prompt_embedding": prompt, "semantic_score": sem}
    return out, new_stats
