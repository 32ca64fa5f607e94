"""Data-parallel gradient exchange for the VAE step: one process per GPU, RCCL over xGMI via torch.distributed.

The reference has no distributed code (no `distributed`, `nccl`, `LOCAL_RANK` anywhere), so there is no
behaviour to match except: N ranks at per-rank batch B give the single-process result at global batch N*B.
GroupNorm is per-sample and MSE / KL are means, so averaging the per-rank gradients is exactly that (VAE-only).

All 72 gradients live in ONE flat fp32 buffer (lunaris_orion_amd.vae), so the exchange is a single collective on a
contiguous buffer — no bucketing logic, no per-tensor launches.  `FlatGradSync`:
  * backend "nccl" (= RCCL on ROCm): `all_reduce(AVG)` on a side stream that waits for the backward's event, joined
    before clip+AdamW; optional fp16 compression of the payload (halves the xGMI bytes; the sum is still taken in
    the wire dtype by RCCL, so it is off by default to keep DP == single-process to fp32 rounding);
  * backend "gloo" (CPU tests): SUM then divide (gloo has no AVG).
The module is pure host logic on top of torch.distributed and is exercised by world-size-2 gloo tests on CPU.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, group: Optional["dist.ProcessGroup"] = None, compress_fp16: bool = False, side_stream: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.compress = compress_fp16
        self.stream = None
        self._side = side_stream
        self._wire = None

    def __call__(self, flat_grads: torch.Tensor) -> None:
        """Average `flat_grads` in place across ranks; returns when the result is ordered on the current stream."""
        if self.world == 1:
            return
        if flat_grads.is_cuda:
            self._cuda(flat_grads)
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
            flat_grads.div_(self.world)

    def _cuda(self, g: torch.Tensor) -> None:
        cur = torch.cuda.current_stream()
        if self._side and self.stream is None:
            self.stream = torch.cuda.Stream()
        st = self.stream if self._side else cur
        if st is not cur:
            st.wait_stream(cur)                 # the backward that produced `g` ran on `cur`
        with torch.cuda.stream(st):
            if self.compress:
                if self._wire is None or self._wire.numel() != g.numel():
                    self._wire = torch.empty_like(g, dtype=torch.float16)
                self._wire.copy_(g)
                dist.all_reduce(self._wire, op=dist.ReduceOp.AVG, group=self.group)
                g.copy_(self._wire)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
        if st is not cur:
            cur.wait_stream(st)                 # clip + AdamW on `cur` see the averaged gradients


def shard_batch(global_batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard of a global batch for this rank (drop_last semantics like train_hybrid.py:569)."""
    per = global_batch.shape[0] // world
    return global_batch[rank * per:(rank + 1) * per]
