"""The three Linear layers' weight gradients kept as rank-B factors (csrc/lo_lowrank.hip; lunar_generate.py:124-125, 150-152, 165,
207-208 are the layers, train_hybrid.py:913, 921 the clip and AdamW that consume the gradients): the fused step never writes the
two [2L, 32768] / [32768, L] gradient matrices -- clip_grad_norm_'s norm comes from Gram matrices of the factors and the AdamW pass
forms each gradient tile with MFMA.  Checked against the materialised path of the same library (`LO_LINEAR_FACTORED=0`, the round-3
path, itself pinned to the reference's golden gradients) and against the oracle:
  * the gradients written out from the factors == lo_wgrad_tn's, every other gradient bit for bit;
  * the norm == the norm over the materialised buffer;
  * parameters, Adam moments and fp16 operand copies after optimizer steps == the materialised path's (Adam's first step is
    lr * g / |g|: sign-like on near-zero elements, so the parameter comparison is bounded by lr and tight on average);
  * batch sizes that need zero padding of the factor's batch axis (1, 33), the full batch 64 / latent 512, batch 96 (three MFMA K steps);
  * the update is skipped on a non-finite norm (GradScaler semantics) with factors exactly like with gradients.
"""
import numpy as np
import pytest
import torch

from oracle import vae_ref as R

pytestmark = pytest.mark.gpu


def _stepper(L, factored, **kw):
    from lunaris_orion_amd.trainer import VAEStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    m = LunarisCoreVAE(L)
    m.load_state_dict(R.closed_form_params(L))
    m = m.to("cuda")
    st = VAEStepper(m, **kw)
    st.linear_factored = factored
    return m, st


def _lin_names(m):
    return ("encoder.fc_mu.weight", "encoder.fc_logvar.weight", "decoder.fc.weight")


@pytest.mark.parametrize("B,L", [(1, 64), (33, 256), (64, 512), (96, 256)])
def test_factored_gradients_and_norm_equal_the_materialised_path(B, L):
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    eps = R.closed_form_eps(B, L, salt=0).cuda()
    out = {}
    for fac in (False, True):
        m, st = _stepper(L, fac, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)
        st.step(x, 0, eps)
        met = st.metrics()
        eng = m._engine(B)
        from lunaris_orion_amd import _lib
        assert bool(_lib.lib.lo_vae_linear_factored(eng.handle)) == fac
        if fac:
            raw = st.grads.clone()                      # before anything is written out: the Linear ranges were never touched
            o, n, _ = [lay for lay, (k, _) in zip(m._layout, m.named_parameters()) if k == "decoder.fc.weight"][0]
            assert raw[o:o + n].abs().max().item() == 0.0
        out[fac] = ({k: g.clone() for (k, _), g in zip(m.named_parameters(), st.parameter_grads())}, met["grad_norm"])
    gm, gf = out[False][0], out[True][0]
    for k in gm:
        if k in _lin_names(None):
            err = (gf[k] - gm[k]).norm().item() / (gm[k].norm().item() + 1e-30)
            assert err <= 2e-6, (k, err)                # same fp16 products, fp32 accumulation in a different order
        else:
            assert torch.equal(gf[k], gm[k]), k         # the rest of the backward is the same launches
    assert abs(out[True][1] - out[False][1]) <= 2e-5 * out[False][1], (out[True][1], out[False][1])


def test_factored_gradients_match_the_oracle():
    """Not only self-consistent: the written-out Linear gradients against autograd of the CPU oracle (fp16-operand tolerance of
    tests/test_vae_gpu.py), and the norm against the oracle's clip_grad_norm_."""
    L, B = 256, 2
    x, eps = R.normalise_sprites(R.closed_form_sprites(B)), R.closed_form_eps(B, L, salt=0)
    m, st = _stepper(L, True, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)
    st.step(x.cuda(), 0, eps.cuda())
    o = R.OracleTrainer(R.closed_form_params(L)).step(x, eps, 0.0, do_update=False)
    for (k, _), g in zip(m.named_parameters(), st.parameter_grads()):
        if k in _lin_names(None):
            ref = o["grads"][k]
            assert (g.cpu() - ref).norm().item() <= 3e-2 * ref.norm().item(), k
    assert abs(st.metrics()["grad_norm"] - o["grad_norm"]) <= 2e-3 * o["grad_norm"]


@pytest.mark.parametrize("B,L,pipelined", [(4, 256, False), (64, 512, True)], ids=["b4_l256_serial", "b64_l512_pipelined"])
def test_factored_optimizer_steps_equal_the_materialised_path(B, L, pipelined):
    lr = 1e-4
    xs = [R.normalise_sprites(R.closed_form_sprites(B, salt=s)).cuda() for s in range(2)]
    res = {}
    for fac in (False, True):
        m, st = _stepper(L, fac, lr=lr, max_grad_norm=1.0, pipeline_optimizer=pipelined)
        trace = []
        for s in range(3):
            st.step(xs[s % 2], s, R.closed_form_eps(B, L, salt=s).cuda())
            met = st.metrics()
            trace.append((met["recon_loss"], met["kl_loss"], met["grad_norm"], met["clip_coef"]))
        st.synchronize_parameters()
        torch.cuda.synchronize()
        eng = m._engine(B)
        with torch.no_grad():
            recon, mu, _ = m(xs[0], R.closed_form_eps(B, L, salt=9).cuda())       # reads the fp16 operand copies the update left
        res[fac] = (trace, m.flat_parameters().clone(), st.exp_avg.clone(), st.exp_avg_sq.clone(), recon.clone(), mu.clone())
    ta, tb = res[False][0], res[True][0]
    for a, b in zip(ta, tb):
        assert abs(a[0] - b[0]) <= 2e-5 * abs(a[0]) and abs(a[1] - b[1]) <= 2e-4 * abs(a[1]) and abs(a[2] - b[2]) <= 1e-4 * a[2] and abs(a[3] - b[3]) <= 1e-4
    dp = (res[True][1] - res[False][1]).abs()
    assert dp.max().item() <= 2.5 * 3 * lr and dp.mean().item() <= 0.02 * lr, (dp.max().item(), dp.mean().item())
    dm = (res[True][2] - res[False][2]).norm().item() / res[False][2].norm().item()
    dv = (res[True][3] - res[False][3]).norm().item() / res[False][3].norm().item()
    assert dm <= 1e-4 and dv <= 1e-4, (dm, dv)
    assert (res[True][4] - res[False][4]).abs().max().item() <= 5e-3 and (res[True][5] - res[False][5]).abs().max().item() <= 5e-3


def test_factored_update_is_skipped_on_a_non_finite_norm():
    """A non-finite factor (an overflowing fp16 activation gradient) makes the Gram matrices, hence the norm, non-finite: the update
    is skipped on the device and counted, exactly like with materialised gradients (train_hybrid.py:917-923, GradScaler)."""
    L, B = 256, 2
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    m, st = _stepper(L, True, lr=1e-3)
    m.loss_scale = 2.0 ** 40                            # overflows the fp16 activation gradients
    p0 = m.flat_parameters().clone()
    st.step(x, 0, R.closed_form_eps(B, L, salt=0).cuda())
    st.synchronize_parameters()
    torch.cuda.synchronize()
    met = st.metrics()
    assert met["grads_finite"] == 0.0 and met["skipped_steps"] >= 1.0
    assert torch.equal(m.flat_parameters(), p0)


def test_gathered_factors_give_the_average_of_the_ranks_gradients():
    """Data parallel (mode 2): two "ranks" run the product's phased backward on their shards one after the other on this GPU; a stand-in
    for FlatGradSync collects the factor blocks they hand over (what RCCL's all-gather would deliver, rank-major), and the library
    forms the averaged Linear gradients from the gathered buffer.  Result == the mean of the two ranks' lo_wgrad_tn gradients to fp32
    rounding, == the global-batch gradient to the fp16-operand tolerance of the DP identity test, and the Linear weight ranges of
    the flat buffer were never written by the backward itself."""
    import ctypes as C
    from lunaris_orion_amd import _lib
    L, Bper, world = 256, 2, 2
    x = R.normalise_sprites(R.closed_form_sprites(Bper * world)).cuda()
    eps = R.closed_form_eps(Bper * world, L, salt=0).cuda()

    class Collector:
        world, supports_then = 2, False

        def __init__(self):
            self.blocks, self.mat = [], None

        def begin(self, g, **kw):
            return False

        def begin_factored(self, pieces, factors, materialize, then=None, pre=None):
            self.blocks.append(factors.clone())
            self.mat = materialize
            return False

        def finish(self):
            pass

        def average_small(self, t):
            pass

    col = Collector()
    per_rank = []
    for r in range(world):
        xs, es = x[r * Bper:(r + 1) * Bper].contiguous(), eps[r * Bper:(r + 1) * Bper].contiguous()
        m0, s0 = _stepper(L, False, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)      # this rank's gradients, written out by lo_wgrad_tn
        s0.step(xs, 0, es)
        per_rank.append(s0.flat_grads().clone())
        mr, sr = _stepper(L, False, lr=0.0, weight_decay=0.0, max_grad_norm=1e9, grad_sync=col)
        assert sr.dp_factored
        sr.grads.fill_(float("nan"))
        sr.step(xs, 0, es)
        torch.cuda.synchronize()
    lin = [(o, n, k) for (o, n, _), (k, _) in zip(mr._layout, mr.named_parameters()) if k in _lin_names(None)]
    for o, n, k in lin:
        assert torch.isnan(sr.grads[o:o + n]).all(), k          # the phased backward left the factors, not the gradients
    gathered = torch.cat(col.blocks)
    col.mat(gathered, world)                                     # rank 1's engine: fac_scale of its own backward, same loss scale on both
    torch.cuda.synchronize()
    mean = (per_rank[0] + per_rank[1]) * 0.5
    mg, sg = _stepper(L, False, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)
    sg.step(x, 0, eps)
    g_global = sg.flat_grads()
    for o, n, k in lin:
        a, b, c = sr.grads[o:o + n].double(), mean[o:o + n].double(), g_global[o:o + n].double()
        assert (a - b).norm().item() <= 2e-6 * b.norm().item(), (k, (a - b).norm().item() / b.norm().item())
        assert (a - c).norm().item() <= 2e-3 * c.norm().item(), k
