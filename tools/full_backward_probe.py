#!/usr/bin/env python
"""Three timed hybrid steps with `--teacher_full_backward` semantics (HybridStepper(teacher_full_backward=True)) at batch 64, teacher dropout
0.1: wall / host time per step and peak device memory.  The target of `rocprofv3 --kernel-trace --stats -- python3 tools/full_backward_probe.py`
(per-kernel durations of the teacher's full backward: profiles/r04_teacher_full_backward_kernel_stats.csv).

  python tools/full_backward_probe.py [feature_dim]
"""
import sys, torch, time
sys.path.insert(0, "/root/repo")
from lunaris_orion_amd.teacher import LunarMoETeacher
from lunaris_orion_amd.trainer import HybridStepper
from lunaris_orion_amd.vae import LunarisCoreVAE
torch.manual_seed(1)
B = 64
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
t = LunarMoETeacher(num_experts=4, feature_dim=F, embedding_dim=256, dropout_rate=0.1).to("cuda").train()
v = LunarisCoreVAE(latent_dim=512).to("cuda")
hs = HybridStepper(v, t, teacher_full_backward=True)
x = torch.rand(B, 3, 128, 128, device="cuda") * 2 - 1
for i in range(2): hs.step(x, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3): hs.step(x, i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("feature_dim", F, "host ms/step", (t1 - t0) / 3 * 1e3, "wall ms/step", (t2 - t0) / 3 * 1e3, "max mem GB", torch.cuda.max_memory_allocated() / 2**30)
