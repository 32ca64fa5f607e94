"""Sprite input path: the step BEFORE `_process_batch`.

Reference: `PixelArtDataset` (train_hybrid.py:100-201) memory-maps `sprites*.npy` (uint8 [N,128,128,3]), checks the row
count of `labels*.csv`, and per sample does `x/127.5 - 1` + HWC->CHW on the CPU inside DataLoader workers
(`:529-585`: 90/10 `random_split`, `drop_last=True`).  Here the CPU only gathers raw uint8 rows; normalisation and the
layout change run on the GPU (`lo_decode_sprites_u8`), and the gather + host->device copy of batch k+1.. overlap the
training step of batch k:

    SpriteShards   global index over the memory-mapped shards (same files, same checks, same error texts)
    SpriteFeeder   background thread: gather -> pinned uint8 buffer -> async copy on a side HIP stream -> event;
                   the consumer makes the compute stream wait on that event (no host synchronisation)
"""
from __future__ import annotations

import glob
import os
import queue
import threading
from typing import Iterable, Iterator, Optional

import numpy as np
import torch


class SpriteShards:
    """sprites*.npy (uint8 [N,128,128,3], memory-mapped) + labels*.csv; only the images feed the step (train_hybrid.py:995)."""

    def __init__(self, data_dir: str):
        files = sorted(glob.glob(os.path.join(data_dir, "sprites*.npy")))
        labels = sorted(glob.glob(os.path.join(data_dir, "labels*.csv")))
        if not files or not labels:
            raise ValueError(f"No sprites or labels files found in {data_dir}")
        self.shards = [np.load(f, mmap_mode="r") for f in files]
        for f, s in zip(files, self.shards):
            if s.ndim != 4 or s.shape[1:] != (128, 128, 3):
                raise ValueError(f"Expected 128x128x3 images in {f}, got {s.shape[1:]}")
            if s.dtype != np.uint8:
                raise ValueError(f"Expected uint8 sprites in {f}, got {s.dtype}")
        self.cum = np.cumsum([0] + [len(s) for s in self.shards])
        n_rows = 0
        for lf in labels:
            with open(lf, "rb") as fh:
                n_rows += max(0, sum(1 for _ in fh) - 1)
        if n_rows != len(self):
            raise AssertionError(f"Mismatch between total sprites ({len(self)}) and labels ({n_rows})")

    def __len__(self) -> int:
        return int(self.cum[-1])

    def gather_into(self, idx: np.ndarray, out: np.ndarray) -> None:
        """out[j] = sprite idx[j] (uint8 HWC); rows of one shard are copied with one fancy-index read."""
        idx = np.asarray(idx, dtype=np.int64)
        shard = np.searchsorted(self.cum, idx, side="right") - 1
        for f in np.unique(shard):
            sel = np.nonzero(shard == f)[0]
            out[sel] = self.shards[int(f)][idx[sel] - self.cum[int(f)]]

    def batch_u8(self, idx: np.ndarray) -> torch.Tensor:
        out = np.empty((len(idx), 128, 128, 3), dtype=np.uint8)
        self.gather_into(idx, out)
        return torch.from_numpy(out)


def split_indices(n: int, train_fraction: float = 0.9, generator: Optional[torch.Generator] = None):
    """`random_split(dataset, [int(0.9 n), n - int(0.9 n)])` (train_hybrid.py:552-556): one permutation, two slices."""
    n_train = int(train_fraction * n)
    perm = torch.randperm(n, generator=generator).numpy()
    return perm[:n_train], perm[n_train:]


def steps_per_epoch(n_train: int, batch: int, world: int = 1) -> int:
    """Optimizer micro-batches per epoch and rank (the same on every rank)."""
    return n_train // (world * batch)


def epoch_batches(indices: np.ndarray, batch: int, rank: int = 0, world: int = 1, shuffle: bool = True,
                  rng: Optional[np.random.Generator] = None) -> Iterator[np.ndarray]:
    """Index batches of one epoch for this rank: shuffle, `drop_last` at the GLOBAL batch (train_hybrid.py:563-570), stride by
    rank.  Every rank gets exactly ``steps_per_epoch(len(indices), batch, world)`` batches: the tail that does not fill one
    more global batch of world * batch sprites is dropped BEFORE striding, so no rank can run one gradient exchange more
    than the others (mismatched collectives hang)."""
    order = (rng.permutation(indices) if rng is not None else np.random.permutation(indices)) if shuffle else np.asarray(indices)
    order = order[: steps_per_epoch(len(order), batch, world) * world * batch][rank::world]
    for b in range(len(order) // batch):
        yield np.sort(order[b * batch:(b + 1) * batch])      # sorted: sequential reads inside a shard


class SpriteFeeder:
    """Iterates device uint8 batches [B,128,128,3]; `depth` batches are gathered / copied ahead of the consumer."""

    def __init__(self, shards: SpriteShards, batches: Iterable[np.ndarray], batch: int, device="cuda", depth: int = 3):
        if not torch.cuda.is_available():
            raise RuntimeError("SpriteFeeder needs a GPU (pinned staging buffers + a copy stream); there is no CPU path")
        dev = torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.shards, self.batch, self.device, self.depth = shards, batch, dev, max(2, depth)
        self._pinned = [torch.empty((batch, 128, 128, 3), dtype=torch.uint8).pin_memory() for _ in range(self.depth)]
        self._dev = [torch.empty((batch, 128, 128, 3), dtype=torch.uint8, device=self.device) for _ in range(self.depth)]
        self._copied = [torch.cuda.Event() for _ in range(self.depth)]
        self._consumed = [torch.cuda.Event() for _ in range(self.depth)]
        self._stream = torch.cuda.Stream(device=self.device)
        self._free: "queue.Queue[int]" = queue.Queue()
        self._ready: "queue.Queue[Optional[int]]" = queue.Queue()
        for i in range(self.depth):
            self._free.put(i)
        self._last: Optional[int] = None
        self._err: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._produce, args=(iter(batches),), daemon=True)
        self._thread.start()

    def _produce(self, it):
        try:
            torch.cuda.set_device(self.device)
            for idx in it:
                if len(idx) != self.batch:
                    raise ValueError(f"batch of {len(idx)} indices, feeder built for {self.batch}")
                slot = self._free.get()
                if slot is None:
                    return
                self._consumed[slot].synchronize()                    # the step that read this slot has finished
                self.shards.gather_into(idx, self._pinned[slot].numpy())
                with torch.cuda.stream(self._stream):
                    self._dev[slot].copy_(self._pinned[slot], non_blocking=True)
                    self._copied[slot].record(self._stream)
                self._ready.put(slot)
        except BaseException as e:                                    # surfaced in the consumer thread
            self._err = e
        finally:
            self._ready.put(None)

    def __iter__(self):
        return self

    def __next__(self) -> torch.Tensor:
        if self._last is not None:                                     # hand the previous slot back once its reader is enqueued
            self._consumed[self._last].record(torch.cuda.current_stream(self.device))
            self._free.put(self._last)
            self._last = None
        slot = self._ready.get()
        if slot is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        torch.cuda.current_stream(self.device).wait_event(self._copied[slot])
        self._last = slot
        return self._dev[slot]

    def close(self):
        self._free.put(None)
