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
  * backend "gloo" (CPU tests): SUM then divide (gloo has no AVG);
  * `mode="direct"`: the xGMI-shaped exchange of SURVEY §8(e).  xGMI is a full mesh of point-to-point links (7 per GPU), so a
    ring all-reduce moves 2(N-1)/N of the buffer over ONE link per GPU (2.8 ms for 244 MB at 8 ranks).  Direct form: every rank
    sends chunk j of its buffer straight to rank j (`all_to_all_single`: N-1 concurrent peer transfers, one per link), sums the N
    chunks it received (its share of the reduce-scatter), and the shares are gathered back (`all_gather_into_tensor`): each
    link carries 2/N of the buffer.  Behind a flag (`bench.py --dp-exchange direct`) until an 8-GPU node has measured it.
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
                 wire_scale: float = 1024.0, mode: str = "allreduce", time_exposed: bool = False):
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
        if mode not in ("allreduce", "direct"):
            raise ValueError(f"unknown exchange mode {mode!r}")
        self.mode = mode
        self._stage = {}            # direct mode: (receive buffer, share) per exchanged slice
        # bookkeeping for the bench line: bytes handed over per begin() of a step, HIP events around finish()
        self.time_exposed = time_exposed
        self._phase_bytes, self._phase_log = [], None
        self._events, self._finishes = [], 0

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
        buf = wire if wire is not None else g
        self._phase_bytes.append(buf.numel() * buf.element_size())
        if self.mode == "direct" and (self.world > 1 or self.force):
            self._pending.append((self._begin_direct(buf), wire, g))
            return
        work = dist.all_reduce(buf, op=self._op(), group=self.group, async_op=True)
        self._pending.append((work, wire, g))

    def _begin_direct(self, buf: torch.Tensor):
        """all-to-all reduce-scatter + all-gather on the largest prefix divisible by the world size (+ a tiny all-reduce for
        the remainder).  Returns the list of waits finish() has to perform, in order."""
        n, w = buf.numel(), self.world
        chunk = n // w
        body = buf[: chunk * w]
        key = (buf.data_ptr(), n, buf.dtype)
        st = self._stage.get(key)
        if st is None:
            st = self._stage[key] = (torch.empty(chunk * w, dtype=buf.dtype, device=buf.device), torch.empty(chunk, dtype=buf.dtype, device=buf.device))
        recv, share = st
        steps = []
        if chunk > 0:
            a2a = dist.all_to_all_single(recv, body, group=self.group, async_op=True)      # recv[j] = rank j's chunk `rank`
            steps.append(("a2a", a2a, recv, share, body))
        if n > chunk * w:
            steps.append(("tail", dist.all_reduce(buf[chunk * w:], op=self._op(), group=self.group, async_op=True), None, None, None))
        return steps

    def finish(self) -> None:
        ev = None
        if self.time_exposed and self._pending and torch.cuda.is_available() and self._pending[0][2].is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for work, wire, g in self._pending:
            if isinstance(work, list):                       # direct mode
                buf = wire if wire is not None else g
                for kind, wk, recv, share, body in work:
                    wk.wait()
                    if kind == "a2a":                        # this rank's share of the reduce-scatter, then the gather
                        torch.sum(recv.view(self.world, -1), dim=0, out=share)
                        share.div_(self.world)
                        dist.all_gather_into_tensor(body, share, group=self.group)
                    elif self.backend != "nccl":             # remainder: plain all-reduce (gloo: SUM, so divide)
                        buf[(buf.numel() // self.world) * self.world:].div_(self.world)
            else:
                work.wait()                      # NCCL: the current stream waits; gloo: the host waits
            if wire is not None:
                g.copy_(wire)
                g.div_(self.wire_scale)
            if self.backend != "nccl" and not isinstance(work, list):
                g.div_(self.world)
        if ev is not None:
            ev[1].record()
            if len(self._events) < 4096:
                self._events.append(ev)
        if self._pending:
            self._finishes += 1
            self._phase_log, self._phase_bytes = self._phase_bytes, []
        self._pending.clear()

    # ---- bench bookkeeping ----------------------------------------------------------------------------------
    def reset_timing(self) -> None:
        self._events, self._finishes = [], 0

    def exposed_ms_per_step(self) -> Optional[float]:
        """Mean time the current stream spent between entering finish() and having every exchanged range back (HIP events on
        the stream; includes the wait for the collectives and the fp16-wire unpack): the part of the exchange the backward did
        not hide.  Call after torch.cuda.synchronize()."""
        if not self._events:
            return None
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)

    def bytes_per_phase(self):
        """Bytes handed to the exchange by each begin() of the last step, in hand-over order."""
        return list(self._phase_log or [])

    def average_small(self, t: torch.Tensor) -> None:
        """Average a few fp32 scalars in place (never compressed): the batch means behind the reward baseline."""
        if self.world == 1 and not self.force:
            return
        dist.all_reduce(t, op=self._op(), group=self.group)
        if self.backend != "nccl":
            t.div_(self.world)

    def max_small(self, t: torch.Tensor) -> None:
        """Element-wise maximum over ranks, in place (control flags: e.g. "some rank received SIGINT")."""
        if self.world == 1 and not self.force:
            return
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def __call__(self, flat_grads: torch.Tensor) -> None:
        """Average `flat_grads` in place across ranks; the result is ordered on the current stream."""
        self.begin(flat_grads)
        self.finish()


def shard_batch(global_batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard of a global batch for this rank (drop_last semantics like train_hybrid.py:569)."""
    per = global_batch.shape[0] // world
    return global_batch[rank * per:(rank + 1) * per]
