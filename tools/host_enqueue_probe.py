#!/usr/bin/env python
"""How long does the HOST need to enqueue one training step?  If this is not well below the GPU time per step, the step is
launch-bound on that host and the GPU idles between kernels (the bench line then measures the host, not the kernels)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_sprites
from lunaris_orion_amd.trainer import VAEStepper
from lunaris_orion_amd.vae import LunarisCoreVAE

torch.manual_seed(42)
m = LunarisCoreVAE(latent_dim=512).to("cuda")
st = VAEStepper(m)
x = synth_sprites(64, 0).cuda()
for i in range(10):
    st.step(x, i)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for i in range(N):
    st.step(x, i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step   wall {1e3 * (t2 - t0) / N:.3f} ms/step   (queue drained {1e3 * (t2 - t1):.1f} ms after the last enqueue)")
print("cpus", len(os.sched_getaffinity(0)))
# burst from an idle GPU: no back-pressure from a full queue can be in these numbers
for n in (2, 5, 10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        st.step(x, i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"burst of {n:2d} steps from idle: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, drained {1e3 * (t2 - t1):.2f} ms later")
