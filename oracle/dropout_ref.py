"""Dropout masks of the native teacher, regenerated on the CPU.  TEST INFRASTRUCTURE ONLY (see vae_ref.py).

The reference draws its dropout masks (``nn.Dropout`` / ``nn.Dropout2d``, /root/reference/lunar_evaluator.py:97-99,139-140,
212,225,246,253,353-397) from torch's global generator, which no other implementation can reproduce.  The native library
draws them from a counter RNG instead (lunaris_orion_amd/csrc/lo_common.h: ``lo_splitmix64``, ``lo_drop_site_keys``,
``lo_drop_word``, ``lo_drop_keep``); this file restates those integer functions in numpy, bit for bit, and lays the masks out
in the reference's tensor shapes, so that
  * ``oracle/teacher_ref.teacher_forward(..., masks=TeacherMasks(seed, p, B))`` runs the CPU oracle, and
  * ``oracle/make_golden.py`` runs the REFERENCE ``LunarMoETeacher(dropout_rate=p)`` in train mode (forward hooks on its own
    ``nn.Dropout`` modules replace the module output by ``input * mask / (1 - p)``)
on exactly the masks the GPU applies for the same 64-bit call seed.  torch's dropout semantics are the documented ones:
Bernoulli(1 - p) keep mask, kept values scaled by 1 / (1 - p); ``Dropout2d`` draws one decision per (sample, channel).

Element index order (the library's own, stated in lo_common.h):
  site FE            : [B][HW][192]  (NHWC)         gate / head hidden layers : [B][width]
  conv1 / conv2 drop : [B][C]                       proj_drop                 : [B][HW][C]  (NHWC)
  attn_drop          : [B][543][8][32] = (sample, written position p, head, key); p < 512 is row 0 of chunk p, p >= 512 is
                       row p - 511 of chunk 511 (the only rows of the attention matrix that reach the output)
"""
from __future__ import annotations

import numpy as np
import torch

M64 = (1 << 64) - 1

DS_FE, DS_GATE = 0, 1
DS_SEM, DS_STYLE, DS_PROMPT = 110, 111, 112


def ds_block(e: int, l: int, k: int) -> int:
    """k: 0 conv1 Dropout2d, 1 attn_drop, 2 proj_drop, 3 conv2 Dropout2d."""
    return 2 + (e * 3 + l) * 4 + k


def ds_quality(e: int) -> int:
    return 100 + e


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def site_keys(call_seed: int, site: int):
    k = splitmix64((call_seed ^ ((0xD1B54A32D192ED03 * (site + 1)) & M64)) & M64)
    return k & 0xFFFFFFFF, k >> 32


def drop_words(k0: int, k1: int, pairs: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = pairs.astype(np.uint32) ^ np.uint32(k0)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
        x += np.uint32(k1)
        x *= np.uint32(0x9E3779B1)
        x ^= x >> np.uint32(15)
    return x


def threshold(p: float) -> int:
    return max(1, int(np.rint(np.float32(p) * np.float32(65536.0)))) if p > 0 else 0


def keep_flat(call_seed: int, site: int, n: int, p: float) -> np.ndarray:
    """keep[i] for i < n (bool)."""
    k0, k1 = site_keys(call_seed, site)
    npair = (n + 1) // 2
    w = drop_words(k0, k1, np.arange(npair, dtype=np.uint32))
    bits = np.empty(npair * 2, dtype=np.uint32)
    bits[0::2] = w & np.uint32(0xFFFF)
    bits[1::2] = w >> np.uint32(16)
    return bits[:n] >= np.uint32(threshold(p))


class TeacherMasks:
    """Multiplicative masks (keep / (1 - p), float32 torch tensors in the reference's layouts) of one teacher forward."""

    def __init__(self, call_seed: int, p: float, B: int, H: int = 128, W: int = 128, C: int = 128, device=None):
        self.seed, self.p, self.B, self.H, self.W, self.C = int(call_seed), float(p), B, H, W, C
        self.scale = 1.0 / (1.0 - float(np.float32(p)))
        self.device = device        # where the mask tensors live (None = CPU); the bits always come from the numpy restatement

    def _t(self, keep: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(keep.astype(np.float32)) * np.float32(self.scale)
        return t if self.device is None else t.to(self.device)

    def elementwise_nchw(self, site: int, C: int) -> torch.Tensor:
        """[B, C, H, W] mask of an elementwise dropout on a feature map (library index order: NHWC)."""
        k = keep_flat(self.seed, site, self.B * self.H * self.W * C, self.p).reshape(self.B, self.H, self.W, C)
        return self._t(k).permute(0, 3, 1, 2).contiguous()

    def channelwise(self, site: int, C: int) -> torch.Tensor:
        """[B, C, 1, 1] mask of a Dropout2d."""
        return self._t(keep_flat(self.seed, site, self.B * C, self.p)).reshape(self.B, C, 1, 1)

    def rows(self, site: int, width: int) -> torch.Tensor:
        """[B, width] mask of a hidden layer."""
        return self._t(keep_flat(self.seed, site, self.B * width, self.p)).reshape(self.B, width)

    def attention(self, site: int, heads: int = 8, chunk: int = 32) -> torch.Tensor:
        """[B, 543, heads, chunk]: masks of the attention rows that reach the output (see module docstring)."""
        n = 512 + chunk - 1
        return self._t(keep_flat(self.seed, site, self.B * n * heads * chunk, self.p)).reshape(self.B, n, heads, chunk)

    def attention_chunk(self, site: int, i: int, heads: int = 8, chunk: int = 32) -> torch.Tensor:
        """Mask for the reference's ``attn_drop`` call of chunk i: [B, heads, chunk(query), chunk(key)].  Rows that never
        reach the output (every row but 0, except in the last chunk) get the identity."""
        a = getattr(self, "_att_cache", None)
        if a is None or a[0] != site:
            a = (site, self.attention(site, heads, chunk))
            self._att_cache = a
        full = a[1]
        m = torch.ones(self.B, heads, chunk, chunk, device=full.device)
        m[:, :, 0, :] = full[:, i]
        if i == 511:
            m[:, :, 1:, :] = full[:, 512:].permute(0, 2, 1, 3)
        return m
