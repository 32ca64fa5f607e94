"""GroupNorm inside the convolution epilogues through a per-sample rendezvous (csrc/lo_common.h: LoGnFuse; lo_internal.h:
LoGnBwdFuse::dv -- the workgroups that hold one sample's tiles exchange their partial sums and wait for each other) against the
separate passes: the forward form (LO_GN_FUSE=1, opt-in) against lo_gn_fwd, the backward apply (default) against lo_gn_bwd_apply.
Same statistics, same arithmetic, so every output, every saved activation and every gradient of a training step must agree BIT FOR
BIT, at the oracle's batch size and at BASELINE's batch 64 (where the grids are longer than the chip and workgroups of different
samples interleave); the bounded wait never runs out."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CACHE = {}


def _run(tmp_path, tag, B, L, env_extra):
    """Two optimizer steps of a fresh model under the given knobs, in THIS process: the three knobs involved (LO_GN_FUSE,
    LO_GNB_FUSE, LO_GNB_APPLY_FUSE) are read by lo_vae_create_ex for every plan it makes, so a fresh model under a changed
    environment is a fresh configuration (round 3 started a Python process per configuration: 12 of them, 90 s of the suite).
    Results are cached per configuration: the "separate passes" run is shared by the two tests below."""
    import ctypes as C

    from oracle import vae_ref as R
    from lunaris_orion_amd import _lib
    from lunaris_orion_amd.trainer import VAEStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    knobs = ("LO_GN_FUSE", "LO_GNB_FUSE", "LO_GNB_APPLY_FUSE")
    defaults = {"LO_GN_FUSE": "0", "LO_GNB_FUSE": "1", "LO_GNB_APPLY_FUSE": "1"}
    key = (B, L, tuple(env_extra.get(k, defaults[k]) for k in knobs))
    if key in _CACHE:
        return _CACHE[key]
    saved = {k: os.environ.get(k) for k in knobs}
    try:
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env_extra)
        m = LunarisCoreVAE(L); m.load_state_dict(R.closed_form_params(L)); m = m.to("cuda")
        x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
        st = VAEStepper(m, gradient_accumulation_steps=1)
        out = {}
        for s in range(2):
            recon, mu, logvar = st.step(x, s, R.closed_form_eps(B, L, salt=s).cuda())
            met = st.metrics()                       # raises if a rendezvous ran out
            out[f"recon{s}"], out[f"mu{s}"], out[f"logvar{s}"] = recon.cpu(), mu.cpu(), logvar.cpu()
            for (name, _), g_ in zip(m.named_parameters(), st.parameter_grads()):
                out[f"grad{s}/{name}"] = g_.detach().cpu().clone()
            out[f"losses{s}"] = torch.tensor([met["recon_loss"], met["kl_loss"], met["grad_norm"]], dtype=torch.float64)
        eng = m._engine(B)

        # intermediate tensors of the last forward, read back from the workspace: a failure names the first layer that differs
        def dbg(which, s_, k_):
            off, dims = C.c_size_t(), (C.c_int * 4)()
            _lib.check(_lib.lib.lo_vae_debug_tensor(eng.handle, which, s_, k_, C.byref(off), dims))
            n = dims[0] * dims[1] * dims[2] * dims[3]
            return eng.ws[off.value:off.value + 2 * n].view(torch.float16).clone().cpu()
        for s_ in range(4):
            for k_ in range(3):
                out[f"a_enc{s_}_conv{k_}_raw"] = dbg(0, s_, k_)
            out[f"b_enc{s_}_out"] = dbg(2, s_, 0)
        for s_ in range(4):
            out[f"c_dec{s_}_raw"] = dbg(1, s_, 0)
            out[f"d_dec{s_}_act"] = dbg(3, s_, 0)
        out["fused_layers"] = torch.tensor(eng.fused_gn_layers)
        out["sync_fail"] = eng.sync_fail.cpu().clone()
        st.synchronize_parameters()
        torch.cuda.synchronize()
        out["params"] = m.flat_parameters().detach().cpu().clone()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    _CACHE[key] = out
    return out


@pytest.mark.parametrize("B,L", [(2, 256), (5, 256), (64, 512)])
def test_fused_groupnorm_epilogue_equals_the_separate_pass_bit_for_bit(tmp_path, B, L):
    fused = _run(tmp_path, "fused", B, L, {"LO_GN_FUSE": "1", "LO_GNB_APPLY_FUSE": "0"})      # opt-in: 2 % slower on the step
    plain = _run(tmp_path, "plain", B, L, {"LO_GN_FUSE": "0", "LO_GNB_APPLY_FUSE": "0"})
    assert int(plain["fused_layers"]) == 0 and int(fused["fused_layers"]) >= 6, (int(fused["fused_layers"]), int(plain["fused_layers"]))
    assert int(fused["sync_fail"].item()) == 0
    bad = [(k, (fused[k].double() - plain[k].double()).abs().max().item()) for k in sorted(fused)
           if k not in ("fused_layers", "sync_fail") and not torch.equal(fused[k], plain[k])]
    assert not bad, bad


@pytest.mark.parametrize("B,L", [(2, 256), (5, 256), (64, 512)])
def test_fused_groupnorm_backward_apply_equals_the_separate_pass(tmp_path, B, L):
    """The GroupNorm-backward APPLY inside the data-gradient epilogue of the consuming layer (lo_internal.h: LoGnBwdFuse::dv; 11 of
    the 16 layers) against the separate lo_gn_bwd_apply launches (LO_GNB_APPLY_FUSE=0): the same group sums in the same order and the
    same element arithmetic, so every dv -- hence every weight, GroupNorm and Linear gradient, every loss and every updated
    parameter of a second step -- is bitwise the same, EXCEPT the conv bias gradients of the fused layers: those are sums of dv whose
    partial rows are per tile instead of per chunk (another summation order: 1e-6 relative)."""
    fused = _run(tmp_path, "fused", B, L, {})
    plain = _run(tmp_path, "plain", B, L, {"LO_GNB_APPLY_FUSE": "0"})
    assert int(fused["sync_fail"].item()) == 0
    bad = []
    for k in sorted(fused):
        if k in ("fused_layers", "sync_fail"):
            continue
        a, b = fused[k].double(), plain[k].double()
        conv_bias = k.startswith("grad") and k.endswith(".0.bias")
        if conv_bias or k.startswith(("grad1", "params", "recon1", "mu1", "logvar1", "losses1")) or k[:2] in ("a_", "b_", "c_", "d_"):
            # conv bias gradients, and everything downstream of the first update (which has used them)
            # second step: AdamW's first update is lr * g / (|g| + eps), so a 1e-6 difference in a tiny bias gradient moves that
            # parameter by a visible fraction of lr, and the second step's tensors differ by up to 2e-4 of their range (measured)
            second = k.startswith(("grad1", "params", "recon1", "mu1", "logvar1", "losses1")) or k[:2] in ("a_", "b_", "c_", "d_")
            tol = (5e-3 if second else 2e-6) * max(b.abs().max().item(), 1e-30)
            if (a - b).abs().max().item() > tol:
                bad.append((k, (a - b).abs().max().item(), tol))
        elif not torch.equal(fused[k], plain[k]):
            bad.append((k, (a - b).abs().max().item(), 0.0))
    first_step = [t for t in bad if t[0].startswith(("grad0", "recon0", "mu0", "logvar0", "losses0"))]
    assert not bad, (first_step or bad)[:12]


def test_a_failed_rendezvous_skips_the_update_on_the_device_and_poisons_the_autograd_path():
    """VERDICT r3 item 1b / ADVICE r3: the word a fused-GroupNorm workgroup sets when its bounded wait runs out is read by the
    optimizer's clip kernel (no host in between): the update of that step and of every later step is skipped on the device until the
    host has looked, `metrics()` and `check_device_health()` (what a checkpoint calls first) raise, and the autograd path — which
    has no optimizer kernel of this library behind it — returns NaN gradients, which a GradScaler / clip_grad_norm_ sees.  The word
    is set by hand here: the wait itself never runs out (the bit-for-bit tests above would fail)."""
    import torch.nn.functional as F
    from oracle import vae_ref as R
    from lunaris_orion_amd._lib import LunarisHipError
    from lunaris_orion_amd.trainer import VAEStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    B, L = 2, 256
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    for pipelined in (False, True):
        m = LunarisCoreVAE(L); m.load_state_dict(R.closed_form_params(L)); m = m.to("cuda")
        st = VAEStepper(m, gradient_accumulation_steps=1, pipeline_optimizer=pipelined)
        st.step(x, 0, R.closed_form_eps(B, L, salt=0).cuda())
        st.synchronize_parameters(); torch.cuda.synchronize()
        st.metrics()                                         # healthy
        p1 = m.flat_parameters().clone()
        eng = m._engine(B)
        eng.sync_fail.fill_(1)                               # "a launch of this step was lost"
        for s in (1, 2):
            st.step(x, s, R.closed_form_eps(B, L, salt=s).cuda())
        st.synchronize_parameters(); torch.cuda.synchronize()
        assert torch.equal(m.flat_parameters(), p1), "an update was applied after the failure word was set"
        with pytest.raises(LunarisHipError, match="gave up waiting"):
            st.check_device_health()
        with pytest.raises(LunarisHipError, match="gave up waiting"):
            st.metrics()
    # the autograd path: NaN gradients
    m = LunarisCoreVAE(L); m.load_state_dict(R.closed_form_params(L)); m = m.to("cuda")
    recon, mu, logvar = m(x, R.closed_form_eps(B, L, salt=0).cuda())
    m._engine(B).sync_fail.fill_(1)
    (F.mse_loss(recon, x) + 0.1 * mu.pow(2).mean()).backward()
    torch.cuda.synchronize()
    assert all(torch.isnan(p.grad).all() for p in m.parameters())
