import sys, time, torch
sys.path.insert(0, '.')
from lunaris_orion_amd.vae import LunarisCoreVAE
from lunaris_orion_amd.teacher import LunarMoETeacher
from lunaris_orion_amd.trainer import HybridStepper
from lunaris_orion_amd import _lib
import bench
B, L = 64, 512
torch.manual_seed(42)
vae = LunarisCoreVAE(L).to('cuda'); t = LunarMoETeacher(embedding_dim=256).to('cuda').train()
hs = HybridStepper(vae, t, gradient_accumulation_steps=1)
x = bench.synth_sprites(B, 1).cuda()
for i in range(3): hs.step(x, i)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for i in range(K): hs.step(x, i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print('hybrid ms/step', dt * 1e3, 'sprites/s', B / dt, hs.metrics())
_lib.lib.lo_prof_enable(1); hs.step(x, 0); torch.cuda.synchronize()
rows = bench.collect_profile(_lib.lib); _lib.lib.lo_prof_enable(0)
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"{k:32s} {r[0]:9.3f} ms n={r[1]:3d} {(r[2]/(r[0]*1e-3)/1e12 if r[0]>0 else 0):8.1f} TF {(r[3]/(r[0]*1e-3)/1e9 if r[0]>0 else 0):8.0f} GB/s  avg {1e3*r[0]/max(r[1],1):7.1f} us")
print('sum', sum(r[0] for r in rows.values()))
