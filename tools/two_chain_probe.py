#!/usr/bin/env python
"""Would two independent half-batch chains on two streams fill the bubbles of one batch-64 chain?  (design probe, not a product path)
Runs (a) one stepper at batch 64, (b) one at batch 32, (c) two steppers at batch 32 concurrently from two host threads on two
streams, and prints sprites/s of each.  The optimizer work is duplicated in (c), so (c) understates a real split-batch step."""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_sprites
from lunaris_orion_amd.trainer import VAEStepper
from lunaris_orion_amd.vae import LunarisCoreVAE


def make(B):
    torch.manual_seed(42)
    m = LunarisCoreVAE(latent_dim=512).to("cuda")
    return VAEStepper(m, pipeline_optimizer=True), synth_sprites(B, 1).cuda()


def run(st, x, n, stream=None):
    if stream is None:
        for i in range(n):
            st.step(x, i)
    else:
        with torch.cuda.stream(stream):
            for i in range(n):
                st.step(x, i)


N = 200
for B in (64, 32):
    st, x = make(B)
    run(st, x, 100)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(st, x, N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / N
    print(f"single chain B={B}: {1e3 * dt:.3f} ms/step  {B / dt:.0f} sprites/s", flush=True)
    del st, x
pairs = [make(32) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for (st, x), s in zip(pairs, streams):
    run(st, x, 50, s)
torch.cuda.synchronize()
t = time.perf_counter()
th = [threading.Thread(target=run, args=(st, x, N, s)) for (st, x), s in zip(pairs, streams)]
for h in th:
    h.start()
for h in th:
    h.join()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / N
print(f"two chains B=32+32 (two threads, two streams): {1e3 * dt:.3f} ms per pair of steps  {64 / dt:.0f} sprites/s", flush=True)
