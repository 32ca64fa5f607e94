import sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_teacher_fullgrad_gpu as TT
from oracle import vae_ref as R
B = 2
x = R.normalise_sprites(R.closed_form_sprites(B))
ora, _ = TT._oracle_grads(x, False, device="cuda")
for gs in (None, 64.0 * B * 16384 * 16, 64.0 * B * 16384 / 16):
    m = TT._teacher(False)
    with torch.no_grad():
        out = m(x.cuda())
    flat = m.full_backward(x.cuda(), out["expert_weights"], TT.QW, gscale=gs)
    torch.cuda.synchronize()
    got = {k: v.detach().cpu() for k, v in m.parameter_grad_views(flat).items()}
    print("gscale", gs)
    rows = []
    for k, g in got.items():
        o = ora.get(k)
        if o is None: continue
        rows.append((k, (g - o).norm().item() / (o.norm().item() + 1e-30), o.norm().item(), torch.nn.functional.cosine_similarity(g.flatten().double(), o.flatten().double(), dim=0).item(), g.norm().item() / (o.norm().item() + 1e-30)))
    if gs is None:
        for r in rows:
            if r[0].startswith("experts.0.") or r[0].startswith("feature") or r[0].startswith("gate") or r[0].startswith("quality_heads.0"):
                print("  %-55s rel %.4f  |o| %.3e  cos %.6f  ratio %.4f" % r)
    print("  mean rel", np.mean([r[1] for r in rows]), "max", max(rows, key=lambda r: r[1])[:2])
