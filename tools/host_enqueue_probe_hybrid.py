#!/usr/bin/env python
"""Host enqueue time vs wall time of the full hybrid step (see tools/host_enqueue_probe.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_sprites
from lunaris_orion_amd.teacher import LunarMoETeacher
from lunaris_orion_amd.trainer import HybridStepper
from lunaris_orion_amd.vae import LunarisCoreVAE

torch.manual_seed(42)
m = LunarisCoreVAE(latent_dim=512).to("cuda")
t = LunarMoETeacher(num_experts=4, feature_dim=128, embedding_dim=256).to("cuda").train()
hs = HybridStepper(m, t, gradient_accumulation_steps=1, pipeline_optimizer=True)
x = synth_sprites(64, 0).cuda()
for i in range(10):
    hs.step(x, i)
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for i in range(N):
    hs.step(x, i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"hybrid: host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step   wall {1e3 * (t2 - t0) / N:.3f} ms/step")
# burst from an idle GPU: the host's own cost of enqueuing a hybrid step (no queue back-pressure in these numbers)
for n in (1, 2, 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        hs.step(x, i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"burst of {n} hybrid steps from idle: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, drained {1e3 * (t2 - t1):.2f} ms later")
