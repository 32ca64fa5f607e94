#!/usr/bin/env python
"""End-to-end rate of the sprite input path + VAE step: uint8 shards on disk (page cache) -> pinned
buffers -> PCIe -> on-GPU decode -> training step, against the same steps on a batch already resident in HBM.

  python tools/feed_bench.py [--batch 64] [--latent 512] [--steps 60]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lunaris_orion_amd.data import SpriteFeeder, SpriteShards, epoch_batches  # noqa: E402
from lunaris_orion_amd.trainer import VAEStepper  # noqa: E402
from lunaris_orion_amd.vae import LunarisCoreVAE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--latent", type=int, default=512)
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    B, n = a.batch, a.batch * (a.steps + 12)
    with tempfile.TemporaryDirectory() as d:
        rng = np.random.default_rng(0)
        for i in range(2):
            np.save(os.path.join(d, f"sprites_{i:03d}.npy"), rng.integers(0, 256, (n // 2, 128, 128, 3), dtype=np.uint8))
            with open(os.path.join(d, f"labels_{i:03d}.csv"), "w") as f:
                f.write("h\n" + "x\n" * (n // 2))
        shards = SpriteShards(d)
        torch.manual_seed(42)
        st = VAEStepper(LunarisCoreVAE(a.latent).to("cuda"), gradient_accumulation_steps=1)
        # gather-only rate of the host side
        out = np.empty((B, 128, 128, 3), dtype=np.uint8)
        bl = list(epoch_batches(np.arange(n), B, rng=np.random.default_rng(1)))
        t0 = time.perf_counter()
        for idx in bl[:20]:
            shards.gather_into(idx, out)
        t_gather = (time.perf_counter() - t0) / 20
        feeder = SpriteFeeder(shards, bl, B)
        t0 = None
        for k, u8 in enumerate(feeder):
            if k == 10:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            st.step(st.decode_sprites(u8), batch_idx=k)
            if k == 10 + a.steps - 1:
                break
        torch.cuda.synchronize()
        dt_feed = (time.perf_counter() - t0) / a.steps
        feeder.close()
        x = st.decode_sprites(shards.batch_u8(bl[0]).cuda())
        for k in range(5):
            st.step(x, batch_idx=k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(a.steps):
            st.step(x, batch_idx=k)
        torch.cuda.synchronize()
        dt_res = (time.perf_counter() - t0) / a.steps
    print(f"batch {B} latent {a.latent}: host gather {t_gather * 1e3:.2f} ms/batch ({B / t_gather:.0f} sprites/s, one thread); "
          f"fed from shards over PCIe {B / dt_feed:.0f} sprites/s ({dt_feed * 1e3:.2f} ms/step); "
          f"resident batch {B / dt_res:.0f} sprites/s ({dt_res * 1e3:.2f} ms/step)")


if __name__ == "__main__":
    main()
