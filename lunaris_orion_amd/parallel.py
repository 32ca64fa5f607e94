"""Data-parallel gradient exchange for the VAE step: one process per GPU, RCCL over xGMI via torch.distributed.

The reference has no distributed code (no `distributed`, `nccl`, `LOCAL_RANK` anywhere), so there is no
behaviour to match except: N ranks at per-rank batch B give the single-process result at global batch N*B.
GroupNorm is per-sample and MSE / KL are means, so averaging the per-rank gradients is exactly that (VAE-only).

All 72 gradients live in ONE flat fp32 buffer (lunaris_orion_amd.vae), so the exchange is a single collective on a
contiguous buffer — no bucketing logic, no per-tensor launches.  `FlatGradSync`:
  * backend "nccl" (= RCCL on ROCm): asynchronous `all_reduce(AVG)` (RCCL's own stream, ordered after the producing
    kernels), joined before clip+AdamW.  VAEStepper splits the backward in three native calls so that the exchange of
    everything from fc_mu.weight to the end of the flat buffer (Linear layers, decoder and final convs: 90 % of the
    bytes, final after phase 1) overlaps the encoder backward, and that of the encoder's last stage (94 % of the rest)
    overlaps stages 3..1; only their 7.7 MB are exchanged after the backward has ended.  Optional
    fp16 wire format (halves the xGMI bytes; g * 1024 on the wire so that small gradients stay in the normal fp16
    range; the sum is then taken in fp16, so it is off by default to keep DP == single-process to fp32 rounding);
  * backend "gloo" (CPU tests): SUM then divide (gloo has no AVG).
The module is pure host logic on top of torch.distributed and is exercised by world-size-2 gloo tests on CPU.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class FlatGradSync:
    """Averages (slices of) the flat gradient buffer across ranks.

    `sync(flat)` = one blocking-in-stream-order exchange.  `begin(slice)` ... `finish()` = asynchronous exchanges
    (torch.distributed `async_op=True`: the collective is ordered after the work already enqueued on the current
    stream and runs on the backend's own stream; `finish()` makes the current stream wait for all of them), used by
    VAEStepper to overlap the exchange of the Linear-layer gradients with the encoder backward.
    """

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, compress_fp16: bool = False, force: bool = False,
                 wire_scale: float = 1024.0):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.compress = compress_fp16
        # fp16 wire: typical gradient elements (1e-5 .. 1e-3) sit at the bottom of the fp16 range; the wire carries
        # g * wire_scale (overflow needs an element > 64) and the result is divided again
        self.wire_scale = float(wire_scale)
        self.force = force              # tests: issue the collectives even in a one-rank group
        self._pending = []          # (work, wire_or_None, destination)
        self._wire = {}

    def _op(self):
        # gloo has no AVG: SUM, divide in finish()
        return dist.ReduceOp.AVG if self.backend == "nccl" else dist.ReduceOp.SUM

    def begin(self, g: torch.Tensor) -> None:
        if (self.world == 1 and not self.force) or g.numel() == 0:
            return
        wire = None
        if self.compress and g.is_cuda:
            key = (g.data_ptr(), g.numel())
            wire = self._wire.get(key)
            if wire is None:
                wire = self._wire[key] = torch.empty(g.numel(), dtype=torch.float16, device=g.device)
            torch.mul(g, self.wire_scale, out=wire)
        work = dist.all_reduce(wire if wire is not None else g, op=self._op(), group=self.group, async_op=True)
        self._pending.append((work, wire, g))

    def finish(self) -> None:
        for work, wire, g in self._pending:
            work.wait()                      # NCCL: the current stream waits; gloo: the host waits
            if wire is not None:
                g.copy_(wire)
                g.div_(self.wire_scale)
            if self.backend != "nccl":
                g.div_(self.world)
        self._pending.clear()

    def average_small(self, t: torch.Tensor) -> None:
        """Average a few fp32 scalars in place (never compressed): the batch means behind the reward baseline."""
        if self.world == 1 and not self.force:
            return
        dist.all_reduce(t, op=self._op(), group=self.group)
        if self.backend != "nccl":
            t.div_(self.world)

    def __call__(self, flat_grads: torch.Tensor) -> None:
        """Average `flat_grads` in place across ranks; the result is ordered on the current stream."""
        self.begin(flat_grads)
        self.finish()


def shard_batch(global_batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard of a global batch for this rank (drop_last semantics like train_hybrid.py:569)."""
    per = global_batch.shape[0] // world
    return global_batch[rank * per:(rank + 1) * per]
