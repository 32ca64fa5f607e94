#!/usr/bin/env python
"""Does a non_blocking device -> pinned-host copy of one float return at once while the GPU is busy?  (The loss-scale policy of
VAEStepper._observe_skipped_updates relies on it; a blocking copy would drain the queue once per step.)"""
import time
import torch
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
src = torch.zeros(2048, device="cuda")
pin = torch.zeros(1).pin_memory()
torch.cuda.synchronize()
for name, fn in (("slice -> pinned copy_(non_blocking=True)", lambda: pin.copy_(src[1027:1028], non_blocking=True)),
                 ("slice -> pinned copy_(non_blocking=False)", lambda: pin.copy_(src[1027:1028]))):
    for _ in range(20):
        b = a @ a                      # ~0.9 ms each: a queue of ~18 ms
    t = time.perf_counter()
    fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"{name}: host time {1e6 * dt:.0f} us with ~18 ms of GPU work queued in front", flush=True)
