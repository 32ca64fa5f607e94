"""Data-parallel gradient exchange for the VAE step: one process per GPU, RCCL over xGMI via torch.distributed.

The reference has no distributed code (no `distributed`, `nccl`, `LOCAL_RANK` anywhere), so there is no
behaviour to match except: N ranks at per-rank batch B give the single-process result at global batch N*B.
GroupNorm is per-sample and MSE / KL are means, so averaging the per-rank gradients is exactly that (VAE-only).

All 72 gradients live in ONE flat fp32 buffer (lunaris_orion_amd.vae), so a hand-over is a contiguous range of it — no
bucketing logic, no per-tensor launches.  VAEStepper splits the backward into three native calls and hands each range over
(`begin`) the moment it is final: everything from fc_mu.weight to the end of the buffer (Linear layers, decoder and final convs:
90 % of the bytes) after phase 1, the encoder's last stage (94 % of the rest) after phase 3, the remaining 7.7 MB at the end.

`FlatGradSync.begin(range)` enqueues the WHOLE exchange of that range on a communication stream that waits only for the work
already on the compute stream (one event): wire pack -> collective(s) -> share sum -> gather -> unpack, every step chained in
stream order, nothing left for `finish()` but one event wait per range.  So the exchange of a range overlaps whatever the
backward still has to do, and only the last range (7.7 MB) is exposed.  Two forms:
  * `mode="direct"` (default for N > 1 in bench.py / train_hybrid.py): the xGMI-shaped exchange of SURVEY §8(e).  xGMI is a full
    mesh of point-to-point links (7 per GPU, ≈153 GB/s each), so a ring all-reduce moves 2(N-1)/N of the buffer over ONE link per
    GPU.  Direct form: every rank sends chunk j of the range straight to rank j (`all_to_all_single`: N-1 concurrent peer
    transfers, one per link), sums the N chunks it received in fp32 (`lo_dp_sum_shares`: its share of the reduce-scatter, ranks in
    order -> reproducible), and the shares are gathered back (`all_gather_into_tensor`): each link carries 2/N of the range.
  * `mode="allreduce"`: one RCCL `all_reduce(AVG)` per range (ring / tree as RCCL chooses).
  fp16 wire format (`compress_fp16`, default with "direct"): g * 1024 in fp16 on the wire (`lo_dp_pack_f16` / `lo_dp_unpack_f16`;
  typical gradient elements 1e-5 .. 1e-3 sit at the bottom of the fp16 range), halves the xGMI bytes; the share sum accumulates in
  fp32 either way.  fp32 wire keeps DP == single-process to fp32 rounding.
DESIGN.md §6 has the bandwidth model (bytes per link per phase, expected exposed time at N = 2 / 4 / 8).
Backend "gloo" with CPU tensors (tests only): the same sequence of collectives, synchronously, with torch ops in place of the
library kernels (gloo has no AVG: SUM then divide) — the world-size-2 tests exercise the ordering and the arithmetic.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class FlatGradSync:
    """Averages ranges of the flat gradient buffer across ranks.

    `begin(range)` ... `finish()`: each `begin` enqueues the complete exchange of one range behind the work already on the current
    stream; `finish()` makes the current stream wait for all of them.  `sync(flat)` = `begin` + `finish`."""

    supports_then = True        # begin(range, then=...) runs a callable behind the exchange (VAEStepper: early gradient norm)

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, compress_fp16: bool = False, force: bool = False,
                 wire_scale: float = 1024.0, mode: str = "allreduce", time_exposed: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.compress = compress_fp16
        self.wire_scale = float(wire_scale)      # overflow on the wire needs a gradient element > 64
        self.compress_min_elems = 4 << 20        # ranges below 16 MB (fp32) stay on the fp32 wire
        self.direct_min_elems = 4 << 20          # ... and go through ONE all-reduce whatever `mode` says
        self._mode_log, self._mode_last = [], []   # exchange form of every range of the current / the last finished step
        self.force = force                       # tests: issue the collectives even in a one-rank group
        if mode not in ("allreduce", "direct"):
            raise ValueError(f"unknown exchange mode {mode!r}")
        self.mode = mode
        self.mode_used = None       # what the last exchange ran as (gloo cannot carry all-to-all / all-gather on GPU tensors)
        self._pending = []          # per begin(): a CUDA event (GPU path) or None (CPU path: already complete)
        self._wire, self._stage = {}, {}
        self._comm: Optional[torch.cuda.Stream] = None
        # bookkeeping for the bench line: bytes handed over per begin() of a step, HIP events around finish()
        self.time_exposed = time_exposed
        self._phase_bytes, self._phase_log = [], None
        self._events, self._finishes = [], 0

    # ---- the pieces: library kernels on the GPU, torch ops for the CPU tests -------------------------------------------------
    def _avg_op(self):
        return dist.ReduceOp.AVG if self.backend == "nccl" else dist.ReduceOp.SUM       # gloo has no AVG: SUM, divide afterwards

    def _pack(self, g: torch.Tensor) -> torch.Tensor:
        key = (g.data_ptr(), g.numel())
        wire = self._wire.get(key)
        if wire is None:
            wire = self._wire[key] = torch.empty(g.numel(), dtype=torch.float16, device=g.device)
        if g.is_cuda:
            from . import _lib
            _lib.check(_lib.lib.lo_dp_pack_f16(g.data_ptr(), wire.data_ptr(), g.numel(), self.wire_scale, _lib.stream_ptr()), "lo_dp_pack_f16")
        else:
            torch.mul(g, self.wire_scale, out=wire)
        return wire

    def _unpack(self, wire: torch.Tensor, g: torch.Tensor, sumsq_scratch: Optional[int] = None) -> None:
        if g.is_cuda:
            from . import _lib
            if sumsq_scratch is not None:        # the early part of the gradient norm in the same pass over the range
                _lib.check(_lib.lib.lo_dp_unpack_f16_sumsq(wire.data_ptr(), g.data_ptr(), g.numel(), 1.0 / self.wire_scale, sumsq_scratch,
                                                           _lib.stream_ptr()), "lo_dp_unpack_f16_sumsq")
                return
            _lib.check(_lib.lib.lo_dp_unpack_f16(wire.data_ptr(), g.data_ptr(), g.numel(), 1.0 / self.wire_scale, _lib.stream_ptr()), "lo_dp_unpack_f16")
        else:
            g.copy_(wire)
            g.div_(self.wire_scale)

    def _sum_shares(self, recv: torch.Tensor, share: torch.Tensor) -> None:
        if recv.is_cuda:
            from . import _lib
            _lib.check(_lib.lib.lo_dp_sum_shares(recv.data_ptr(), share.data_ptr(), self.world, share.numel(),
                                                 1 if recv.dtype == torch.float16 else 0, _lib.stream_ptr()), "lo_dp_sum_shares")
        else:
            acc = recv.view(self.world, -1).float().sum(dim=0) / self.world
            share.copy_(acc)

    def _exchange(self, buf: torch.Tensor) -> None:
        """The collectives of one range, issued on the CURRENT stream (sync-in-stream calls: the stream, not the host, waits)."""
        # gloo implements all_to_all / all_gather for CPU tensors only: the two-rank rehearsals on one GPU (gloo + GPU tensors) fall
        # back to the all-reduce form; RCCL and the CPU tests run what was asked for
        # ... and ranges below `direct_min_elems` (the teacher's head gradients, the exposed last 7.7 MB of the encoder) are latency-
        # bound: one all-reduce instead of all-to-all + share sum + all-gather + tail all-reduce
        direct = self.mode == "direct" and not (self.backend == "gloo" and buf.is_cuda) and buf.numel() >= self.direct_min_elems
        self.mode_used = "direct" if direct else "allreduce"
        self._mode_log.append(self.mode_used)
        if direct:
            n, w = buf.numel(), self.world
            chunk = n // w
            if chunk > 0:
                key = (buf.data_ptr(), n, buf.dtype)
                st = self._stage.get(key)
                if st is None:
                    st = self._stage[key] = (torch.empty(chunk * w, dtype=buf.dtype, device=buf.device),
                                             torch.empty(chunk, dtype=buf.dtype, device=buf.device))
                recv, share = st
                body = buf[: chunk * w]
                dist.all_to_all_single(recv, body, group=self.group)          # recv[j] = rank j's chunk `rank`
                self._sum_shares(recv, share)                                 # this rank's share of the reduce-scatter
                dist.all_gather_into_tensor(body, share, group=self.group)
            if n > chunk * w:                                                 # remainder (< world elements): plain all-reduce
                tail = buf[chunk * w:]
                dist.all_reduce(tail, op=self._avg_op(), group=self.group)
                if self.backend != "nccl":
                    tail.div_(w)
        else:
            dist.all_reduce(buf, op=self._avg_op(), group=self.group)
            if self.backend != "nccl":
                buf.div_(self.world)

    def _run(self, g: torch.Tensor, sumsq_scratch: Optional[int] = None) -> None:
        # small ranges stay on the fp32 wire: the last range of a step (7.7 MB) is exposed and latency-bound, and two launches
        # less on its chain are worth more than 4 MB less on the links
        wire = self._pack(g) if self.compress and g.numel() >= self.compress_min_elems else None
        self._exchange(wire if wire is not None else g)
        if wire is not None:
            self._unpack(wire, g, sumsq_scratch)
        elif sumsq_scratch is not None and g.is_cuda:
            from . import _lib
            _lib.check(_lib.lib.lo_gradnorm_early_range(g.data_ptr(), 0, g.numel(), sumsq_scratch, _lib.stream_ptr()), "lo_gradnorm_early_range")

    # ---- public ---------------------------------------------------------------------------------------------------------------
    def begin(self, g: torch.Tensor, then=None, pre=None, sumsq_scratch: Optional[int] = None) -> bool:
        """Enqueue the exchange of `g`.  `then` (GPU path only): a callable run right behind the exchange with the communication
        stream current, i.e. on the averaged values and still beside the backward (the early part of the gradient norm).  `pre`
        (GPU path only): a callable run in front of the exchange with the communication stream current -- it makes that stream wait
        for whoever finishes the range (lo_vae_wait_handover: the library's side stream; the compute stream is then not held up).
        `sumsq_scratch` (GPU path only): device address of the optimizer's scratch -- the sum of squares of the averaged range is
        left in scratch[512..1024) (lo_gradnorm_early_range), in the unpack pass itself on the fp16 wire.
        Returns whether the hooks were enqueued (False: nothing was exchanged, or CPU tensors)."""
        if (self.world == 1 and not self.force) or g.numel() == 0:
            return False
        self._phase_bytes.append(g.numel() * (2 if (self.compress and g.numel() >= self.compress_min_elems) else g.element_size()))
        if not g.is_cuda:
            self._run(g)                                  # CPU tensors (gloo tests): synchronous
            self._pending.append(None)
            return False
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=g.device)
        ready = torch.cuda.Event()
        ready.record()                                    # `g` is final once the work enqueued so far has run
        self._comm.wait_event(ready)
        with torch.cuda.stream(self._comm):
            if pre is not None:
                pre()
            self._run(g, sumsq_scratch)
            if then is not None:
                then()
            done = torch.cuda.Event()
            done.record()
        self._pending.append(done)
        return True

    def gather(self, block: torch.Tensor) -> torch.Tensor:
        """All-gather of one flat tensor: [world * numel], rank r's block at r * numel (issued on the CURRENT stream)."""
        key = ("gather", block.data_ptr(), block.numel(), block.dtype)
        out = self._stage.get(key)
        if out is None:
            out = self._stage[key] = torch.empty(self.world * block.numel(), dtype=block.dtype, device=block.device)
        if self.backend == "gloo" and block.is_cuda:     # rehearsals: gloo carries all-gather for CPU tensors only
            host = torch.empty(self.world * block.numel(), dtype=block.dtype)
            dist.all_gather_into_tensor(host, block.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, block, group=self.group)
        return out

    def begin_factored(self, pieces, factors: torch.Tensor, materialize, then=None, pre=None) -> bool:
        """The first hand-over range of a step with the Linear layers' weight gradients kept as factors (VAEStepper, mode 2 of
        lo_vae_set_linear_factored): `factors` = this rank's factor block (8.6 MB of fp16 instead of 201 MB of gradients) is
        all-gathered -- exact, no wire rounding -- and `materialize(gathered, world)` writes the AVERAGED gradients of the two matrices
        from the gathered blocks; `pieces` (the rest of the range: biases, decoder convs) are averaged the usual way.  `pre` / `then`
        as for begin().  Everything is chained on the communication stream behind the work already on the current stream."""
        if self.world == 1 and not self.force:
            return False
        self._phase_bytes.append(factors.numel() * factors.element_size() + sum(p.numel() * p.element_size() for p in pieces))
        if not factors.is_cuda:                           # CPU tensors (gloo tests): synchronous
            materialize(self.gather(factors), self.world)
            self._mode_log.append("factors")
            for p in pieces:
                self._run(p)
            self._pending.append(None)
            return False
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=factors.device)
        ready = torch.cuda.Event()
        ready.record()
        self._comm.wait_event(ready)
        with torch.cuda.stream(self._comm):
            if pre is not None:
                pre()
            materialize(self.gather(factors), self.world)
            self._mode_log.append("factors")
            for p in pieces:
                self._run(p)
            if then is not None:
                then()
            done = torch.cuda.Event()
            done.record()
        self._pending.append(done)
        return True

    def finish(self) -> None:
        ev = None
        if self.time_exposed and any(p is not None for p in self._pending):
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for done in self._pending:
            if done is not None:
                torch.cuda.current_stream().wait_event(done)
        if ev is not None:
            ev[1].record()
            if len(self._events) < 4096:
                self._events.append(ev)
        if self._pending:
            self._finishes += 1
            self._phase_log, self._phase_bytes = self._phase_bytes, []
            self._mode_last, self._mode_log = self._mode_log, []
        self._pending.clear()

    # ---- bench bookkeeping ----------------------------------------------------------------------------------
    def reset_timing(self) -> None:
        self._events, self._finishes = [], 0

    def exposed_ms_per_step(self) -> Optional[float]:
        """Mean time the current stream spent inside finish() waiting for exchanges that had not completed yet (HIP events on the
        stream): the part of the exchange the backward did not hide.  Call after torch.cuda.synchronize()."""
        if not self._events:
            return None
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)

    def modes_per_phase(self):
        """Exchange form ("direct" / "allreduce") each range of the last step actually ran as, in hand-over order."""
        return list(self._mode_last)

    def bytes_per_phase(self):
        """Bytes handed to the exchange (on the wire format) by each begin() of the last step, in hand-over order."""
        return list(self._phase_log or [])

    def _small(self, t: torch.Tensor, op) -> None:
        """A few scalars, blocking in stream order on the CURRENT stream (never the communication stream: its queue may hold
        gradient ranges of this step, and the scalars are needed now)."""
        dist.all_reduce(t, op=op, group=self.group)

    def average_small(self, t: torch.Tensor) -> None:
        """Average a few fp32 scalars in place (never compressed): the batch means behind the reward baseline."""
        if self.world == 1 and not self.force:
            return
        self._small(t, self._avg_op())
        if self.backend != "nccl":
            t.div_(self.world)

    def max_small(self, t: torch.Tensor) -> None:
        """Element-wise maximum over ranks, in place (control flags: e.g. "some rank received SIGINT")."""
        if self.world == 1 and not self.force:
            return
        self._small(t, dist.ReduceOp.MAX)

    def __call__(self, flat_grads: torch.Tensor) -> None:
        """Average `flat_grads` in place across ranks; the result is ordered on the current stream."""
        self.begin(flat_grads)
        self.finish()


def shard_batch(global_batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard of a global batch for this rank (drop_last semantics like train_hybrid.py:569)."""
    per = global_batch.shape[0] // world
    return global_batch[rank * per:(rank + 1) * per]
