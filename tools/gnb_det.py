"""Run-to-run bitwise reproducibility of the batch-64 gradient, N repetitions (used with LO_GNB_FUSE=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vae_ref as R
from lunaris_orion_amd.vae import LunarisCoreVAE
from lunaris_orion_amd.trainer import VAEStepper

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
for B in (16, 32, 64):
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    eps = R.closed_form_eps(B, 512, salt=0).cuda()
    ref = None
    for i in range(N):
        m = LunarisCoreVAE(latent_dim=512)
        m.load_state_dict(R.closed_form_params(512, 0))
        m = m.to("cuda")
        st = VAEStepper(m, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)
        st.step(x, 0, eps)
        torch.cuda.synchronize()
        g = [t.detach().clone() for t in st.parameter_grads()]
        if ref is None:
            ref = g
            names = [k for k, _ in m.named_parameters()]
        else:
            for a, b, k in zip(ref, g, names):
                if not torch.equal(a, b):
                    bad += 1
                    print(f"B={B} run {i}: {k} differs, max rel {((a - b).abs().max() / (a.abs().max() + 1e-30)).item():.2e}", flush=True)
    print(f"B={B}: {N} runs compared", flush=True)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
