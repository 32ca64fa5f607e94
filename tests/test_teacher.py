"""LunarMoETeacher drop-in: CPU tests of the module surface, GPU parity of the native forward against the oracle and
the golden fixture generated from the reference class (tests/golden/teacher_B2.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import teacher_ref as T
from oracle import vae_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_module_state_dict_matches_reference_layout():
    from lunaris_orion_amd.teacher import LunarMoETeacher
    m = LunarMoETeacher(num_experts=4, feature_dim=128, embedding_dim=64)
    sd = m.state_dict()
    shapes = T.teacher_param_shapes()
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert len(list(m.parameters())) == 252 and len(list(m.buffers())) == 99
    assert sum(p.numel() for p in m.parameters()) == 4_514_005          # SURVEY §6 (CLI defaults)
    # reference initialisation (lunar_evaluator.py:399-406, 136-137, 258)
    assert float(m.experts[0][0].layer_scale.mean()) == pytest.approx(0.1)
    assert float(m.gate[2].bias.abs().max()) == 0.0 and float(m.feature_extractor.conv1[2].weight.min()) == 1.0
    with pytest.raises(NotImplementedError):
        LunarMoETeacher(feature_dim=384)
    # README High-End recipe (feature_dim 512, embedding_dim 256): 62 936 405 parameters, first blocks get a shortcut
    big = LunarMoETeacher(num_experts=4, feature_dim=512, embedding_dim=256)
    shp = T.teacher_param_shapes(feature_dim=512, embedding_dim=256)
    assert list(big.state_dict().keys()) == list(shp.keys())
    assert all(tuple(v.shape) == tuple(shp[k]) for k, v in big.state_dict().items())
    assert sum(p.numel() for p in big.parameters()) == 62_936_405 and "experts.0.0.shortcut.0.weight" in shp


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
def test_teacher_forward_matches_oracle_and_golden(training):
    from lunaris_orion_amd.teacher import LunarMoETeacher
    g = np.load(os.path.join(GOLD, "teacher_B2.npz"))
    B = int(g["meta"][0])
    S = T.closed_form_teacher_state()
    m = LunarMoETeacher(num_experts=4, feature_dim=128, embedding_dim=64, dropout_rate=0.0)
    m.load_state_dict(S)
    m = m.to("cuda")
    m.train(training)
    x = R.normalise_sprites(R.closed_form_sprites(B))
    out = m(x.cuda())
    torch.cuda.synchronize()
    with torch.no_grad():
        ref, new_stats = T.teacher_forward(x, S, training=training)
    tag = "train" if training else "eval"
    # tolerances: fp16 activations through 24 full-resolution convs, then pooled to [B,128] vectors
    tol = {"quality_scores": 2e-3, "expert_weights": 2e-3, "style_embedding": 2e-2, "prompt_embedding": 2e-2, "semantic_score": 2e-3}
    for k, t in tol.items():
        got = out[k].detach().cpu()   # the outputs carry a graph since round 3 (like the reference module's)
        d = (got - ref[k]).abs().max().item()
        print(tag, k, d)
        assert d <= t, (k, d)
        assert np.abs(got.numpy() - g[f"{tag}/{k}"]).max() <= t, k
    assert out["feature_maps"] is None
    if training:
        sd = m.state_dict()
        for k in ("feature_extractor.fusion.2.running_mean", "experts.3.2.conv2.2.running_var", "feature_extractor.conv1.2.running_var"):
            d = (sd[k].cpu() - new_stats[k]).abs().max().item()
            assert d <= 2e-3 * max(1.0, new_stats[k].abs().max().item()), (k, d)
        assert int(sd["experts.0.0.conv1.2.num_batches_tracked"]) == 1


@pytest.mark.gpu
def test_teacher_head_gradients_match_autograd_of_oracle():
    """A13: d teacher_loss / d (gate, quality_heads) vs autograd through the oracle restatement."""
    import ctypes as C

    from lunaris_orion_amd import _lib
    from lunaris_orion_amd.teacher import LunarMoETeacher
    B = 2
    S = T.closed_form_teacher_state()
    m = LunarMoETeacher(dropout_rate=0.0)
    m.load_state_dict(S)
    m = m.to("cuda").train()
    x = R.normalise_sprites(R.closed_form_sprites(B))
    out = m(x.cuda())
    h, ws, _ = m._engine(B)
    b, e = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.lib.lo_teacher_grad_range(h, C.byref(b), C.byref(e)))
    n = e.value - b.value
    rows = torch.empty(B * n, device="cuda")
    grads = torch.zeros_like(m._flat)
    qw = 0.5
    _lib.check(_lib.lib.lo_teacher_heads_backward(h, m._flat.data_ptr(), ws.data_ptr(), out["expert_weights"].data_ptr(), qw,
                                                  rows.data_ptr(), grads.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    # oracle: autograd through the restatement with the original (pre-forward) state
    S2 = {k: (v.clone().requires_grad_(True) if (k.startswith("gate.") or k.startswith("quality_heads.")) else v) for k, v in S.items()}
    o, _ = T.teacher_forward(x, S2, training=True)
    loss = qw * (-o["quality_scores"].mean())
    loss.backward()
    for i, (k, t) in enumerate(m._named_state()):
        if not (k.startswith("gate.") or k.startswith("quality_heads.")):
            continue
        off = t.data_ptr() - m._flat.data_ptr()
        g = grads[off // 4: off // 4 + t.numel()].view(t.shape).cpu()
        ref = S2[k].grad
        err = (g - ref).norm().item() / (ref.norm().item() + 1e-12)
        assert err <= 2e-2, (k, err)
    # nothing outside the range is written
    assert float(grads[: b.value].abs().max()) == 0.0 and float(grads[e.value:].abs().max()) == 0.0


@pytest.mark.gpu
def test_hybrid_step_runs_and_matches_reference_semantics():
    """One full hybrid step (teacher on): metrics follow train_hybrid.py:870-892 (pg_loss is exactly -0 on the first
    step because the baseline is initialised to the batch mean), VAE losses equal the VAE-only step's."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from lunaris_orion_amd.trainer import HybridStepper, VAEStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    L, B = 256, 2
    P = R.closed_form_params(L)
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    eps0, eps1 = R.closed_form_eps(B, L, 0).cuda(), R.closed_form_eps(B, L, 1).cuda()
    vae = LunarisCoreVAE(L); vae.load_state_dict(P); vae = vae.to("cuda")
    t = LunarMoETeacher(dropout_rate=0.0); t.load_state_dict(T.closed_form_teacher_state()); t = t.to("cuda").train()
    gate_before = t.gate[2].weight.detach().clone()
    conv_before = t.experts[0][0].conv1[0].weight.detach().clone()
    hs = HybridStepper(vae, t, gradient_accumulation_steps=1)
    hs.step(x, 0, eps0)
    m0 = hs.metrics()
    vae2 = LunarisCoreVAE(L); vae2.load_state_dict(P); vae2 = vae2.to("cuda")
    vs = VAEStepper(vae2, gradient_accumulation_steps=1)
    vs.step(x, 0, eps0)
    v0 = vs.metrics()
    assert set(m0) >= {"recon_loss", "kl_loss", "quality_loss", "pg_loss", "semantic_reward", "quality_reward", "baseline", "advantage",
                       "vae_loss", "teacher_loss", "total_loss", "quality_scores"}
    assert m0["recon_loss"] == v0["recon_loss"] and m0["kl_loss"] == v0["kl_loss"]
    assert m0["advantage"] == 0.0 and m0["pg_loss"] == 0.0                      # first step: baseline == batch mean
    assert abs(m0["baseline"] - (m0["quality_reward"] + 0.5 * m0["semantic_reward"])) <= 1e-6
    assert abs(m0["teacher_loss"] - 0.5 * m0["quality_loss"]) <= 1e-7
    assert 0.3 < m0["quality_scores"] < 0.7
    hs.step(x, 1, eps1)
    m1 = hs.metrics()
    assert np.isfinite(list(m1.values())).all() and m1["grads_finite"] == 1.0
    # only gate / quality heads of the teacher move
    assert not torch.equal(t.gate[2].weight, gate_before)
    assert torch.equal(t.experts[0][0].conv1[0].weight, conv_before)
    assert int(t.feature_extractor.conv1[2].num_batches_tracked) == 4           # two teacher calls per step


@pytest.mark.gpu
def test_hybrid_two_steps_match_reference_trace():
    """Two full `_process_batch` steps against the trace the REFERENCE's own modules + torch.optim.AdamW +
    CosineAnnealingWarmRestarts produced on CPU (oracle/make_golden.py::run_hybrid, teacher dropout 0).
    Tolerances: VAE losses 2e-4 relative (fp16 operands, fp32 accumulate), teacher-derived scalars 3e-3 absolute
    (three fp16 expert stacks + train-mode BatchNorm), advantage/pg 5e-4 absolute (they are reward_scale-scaled)."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from lunaris_orion_amd.trainer import HybridStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hybrid_L256_B2.npz"))
    cols, trace = [str(c) for c in g["cols"]], g["trace"]
    L, B = 256, 2
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    vae = LunarisCoreVAE(L); vae.load_state_dict(R.closed_form_params(L)); vae = vae.to("cuda")
    t = LunarMoETeacher(dropout_rate=0.0); t.load_state_dict(T.closed_form_teacher_state()); t = t.to("cuda").train()
    hs = HybridStepper(vae, t, gradient_accumulation_steps=1)
    tol = {"recon_loss": ("rel", 2e-4), "kl_loss": ("rel", 2e-4), "vae_loss": ("rel", 1e-3), "pg_loss": ("abs", 5e-4), "advantage": ("abs", 5e-4)}
    for s in range(trace.shape[0]):
        hs.step(x, s, R.closed_form_eps(B, L, salt=s).cuda())
        m = hs.metrics()
        for j, c in enumerate(cols):
            kind, lim = tol.get(c, ("abs", 3e-3))
            err = abs(m[c] - trace[s, j]) / (abs(trace[s, j]) if kind == "rel" else 1.0)
            assert err <= lim, (s, c, m[c], trace[s, j])
    assert int(g["teacher_params_with_grad"]) == 28
    np.testing.assert_allclose(t.gate[2].weight.detach().cpu().numpy()[:4, :8], g["gate_w_after"], atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
def test_sparse_expert_path_equals_dense_path(training, monkeypatch):
    """The default expert path computes conv2 / proj only on the 6 image rows the attention quirk leaves non-constant and
    k|v instead of q|k|v; LO_T_DENSE=1 runs every convolution in full.  Same outputs, same BatchNorm running statistics."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    S = T.closed_form_teacher_state()
    x = R.normalise_sprites(R.closed_form_sprites(3)).cuda()
    outs, stats = [], []
    for dense in ("0", "1"):
        monkeypatch.setenv("LO_T_DENSE", dense)
        t = LunarMoETeacher(dropout_rate=0.0); t.load_state_dict(S); t = t.to("cuda")
        t.train(training)
        o = t(x)
        outs.append({k: v.detach().cpu() for k, v in o.items() if v is not None})
        stats.append({k: v.detach().cpu().clone() for k, v in t.state_dict().items() if "running_" in k})
    for k in outs[0]:
        assert (outs[0][k] - outs[1][k]).abs().max().item() <= 5e-4, k
    for k in stats[0]:
        ref = stats[1][k]
        assert (stats[0][k] - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), k


@pytest.mark.gpu
def test_teacher_forward_with_fused_tap_conv_kernel():
    """The 8-wave fused-tap 3x3 kernel (lo_conv3x3_pp; selected by default only on long grids, e.g. batch 64) forced on for
    the small parity batch: same tolerances as the default path.  LO_HALO is read once per process -> subprocess."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from oracle import teacher_ref as T, vae_ref as R
from lunaris_orion_amd.teacher import LunarMoETeacher
S = T.closed_form_teacher_state()
m = LunarMoETeacher(dropout_rate=0.0); m.load_state_dict(S); m = m.to("cuda").train()
x = R.normalise_sprites(R.closed_form_sprites(2))
out = m(x.cuda()); torch.cuda.synchronize()
with torch.no_grad():
    ref, stats = T.teacher_forward(x, S, training=True)
for k, t in {"quality_scores": 2e-3, "expert_weights": 2e-3, "style_embedding": 2e-2, "prompt_embedding": 2e-2, "semantic_score": 2e-3}.items():
    d = (out[k].cpu() - ref[k]).abs().max().item()
    assert d <= t, (k, d)
k = "experts.2.1.conv1.2.running_var"
assert (m.state_dict()[k].cpu() - stats[k]).abs().max().item() <= 2e-3 * max(1.0, stats[k].abs().max().item())
print("FUSED_TAP_OK")
""" % root
    env = dict(os.environ, LO_HALO="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert "FUSED_TAP_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_statistics_only_call_has_the_side_effects_of_forward():
    """`_process_batch`'s first teacher call (train_hybrid.py:853-855) only matters through the BatchNorm running
    statistics: `update_statistics_only` must leave the same state as a full train-mode forward."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    S = T.closed_form_teacher_state()
    x = R.normalise_sprites(R.closed_form_sprites(2)).cuda()
    states = []
    for stats_only in (False, True):
        t = LunarMoETeacher(dropout_rate=0.0); t.load_state_dict(S); t = t.to("cuda").train()
        t.update_statistics_only(x) if stats_only else t(x)
        torch.cuda.synchronize()
        states.append({k: v.detach().cpu().clone() for k, v in t.state_dict().items()})
    assert int(states[1]["experts.0.0.conv1.2.num_batches_tracked"]) == 1
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3])
def test_teacher_odd_batch_sizes(B):
    """A single sprite (BatchNorm statistics over one sample) and an odd batch, train mode, against the oracle."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    S = T.closed_form_teacher_state()
    m = LunarMoETeacher(num_experts=4, feature_dim=128, embedding_dim=64, dropout_rate=0.0)
    m.load_state_dict(S)
    m = m.to("cuda").train()
    x = R.normalise_sprites(R.closed_form_sprites(B))
    out = m(x.cuda())
    torch.cuda.synchronize()
    with torch.no_grad():
        ref, _ = T.teacher_forward(x, S, training=True)
    for k, t in {"quality_scores": 2e-3, "expert_weights": 2e-3, "semantic_score": 2e-3, "style_embedding": 2e-2}.items():
        assert (out[k].cpu() - ref[k]).abs().max().item() <= t, k


# ---- dropout (the reference's default: dropout_rate 0.1, train mode) ------------------------------------------------------
def test_dropout_reference_fixture_matches_oracle_with_the_same_masks():
    """CPU: tests/golden/teacher_drop_B2.npz was produced by the REFERENCE teacher (dropout 0.1, train mode) with its own
    nn.Dropout modules hooked to apply the masks of oracle/dropout_ref for DROP_SEED; the oracle on the same masks agrees."""
    from oracle import dropout_ref as D
    g = np.load(os.path.join(GOLD, "teacher_drop_B2.npz"))
    B, seed, p = int(g["meta"][0]), int(g["drop_seed"]), float(g["drop_p"])
    S = T.closed_form_teacher_state()
    x = R.normalise_sprites(R.closed_form_sprites(B))
    with torch.no_grad():
        o, stats = T.teacher_forward(x, S, training=True, masks=D.TeacherMasks(seed, p, B))
    for k in ("quality_scores", "expert_weights", "style_embedding", "prompt_embedding", "semantic_score"):
        assert np.abs(o[k].numpy() - g["train/" + k]).max() <= 2e-5, k
        assert float(g["nodrop_delta/" + k]) > 1e-3, k          # the masks matter: the no-dropout forward is far away
    for k in ("feature_extractor.fusion.2.running_mean", "experts.0.0.conv2.2.running_var", "experts.3.2.conv2.2.running_var"):
        assert np.abs(stats[k].numpy() - g["train/" + k]).max() <= 1e-5, k
    assert abs(float(g["train/attn_zero_fraction"]) - p) < 2e-3    # proj_drop zeroes ~p of the attention output


def test_dropout_mask_generator_statistics():
    """CPU restatement of the counter RNG: keep rate, independence of sites / calls, threshold."""
    from oracle import dropout_ref as D
    n = 1 << 20
    a = D.keep_flat(1, D.ds_block(0, 0, 2), n, 0.1)
    b = D.keep_flat(1, D.ds_block(0, 1, 2), n, 0.1)
    c = D.keep_flat(2, D.ds_block(0, 0, 2), n, 0.1)
    for k in (a, b, c):
        assert abs(k.mean() - 0.9) < 1.5e-3
    assert abs(np.corrcoef(a, b)[0, 1]) < 5e-3 and abs(np.corrcoef(a, c)[0, 1]) < 5e-3
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 5e-3
    assert D.threshold(0.1) == 6554 and D.keep_flat(1, 3, 1000, 0.5).mean() == pytest.approx(0.5, abs=0.06)
    m = D.TeacherMasks(5, 0.1, 2)
    ch = m.channelwise(D.ds_block(1, 2, 0), 128)
    assert ch.shape == (2, 128, 1, 1) and all(min(abs(v), abs(v - 1 / 0.9)) < 1e-6 for v in np.unique(ch.numpy()).tolist())


@pytest.mark.gpu
def test_device_dropout_masks_equal_the_cpu_restatement():
    """The device RNG (lo_dropout_mask = the decisions lo_teacher_forward applies) equals oracle/dropout_ref bit for bit."""
    from lunaris_orion_amd import _lib
    from oracle import dropout_ref as D
    for seed, site, p, n in ((0x5EEDD209C0FFEE11, D.ds_block(2, 1, 2), 0.1, 1 << 20), (7, D.DS_GATE, 0.1, 511), (2**63 + 5, D.ds_quality(3), 0.25, 4097)):
        keep = torch.empty(n, dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib.lo_dropout_mask(seed, site, p, n, keep.data_ptr(), _lib.stream_ptr()), "lo_dropout_mask")
        torch.cuda.synchronize()
        ref = D.keep_flat(seed, site, n, p)
        assert np.array_equal(keep.cpu().numpy().astype(bool), ref), (seed, site)
        assert abs(ref.mean() - (1 - p)) < 0.05


@pytest.mark.gpu
def test_teacher_dropout_forward_matches_reference_fixture_and_oracle():
    """Train mode, dropout 0.1 (the reference's default): the native forward with call seed DROP_SEED against (a) the fixture the
    REFERENCE produced on the same masks and (b) the CPU oracle on the same masks; BatchNorm running statistics included.
    Same tolerances as the dropout-free parity test."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from oracle import dropout_ref as D
    g = np.load(os.path.join(GOLD, "teacher_drop_B2.npz"))
    B, seed, p = int(g["meta"][0]), int(g["drop_seed"]), float(g["drop_p"])
    S = T.closed_form_teacher_state()
    m = LunarMoETeacher(num_experts=4, feature_dim=128, embedding_dim=64)      # dropout_rate: the default 0.1
    assert m.dropout_rate == p
    m.load_state_dict(S)
    m = m.to("cuda").train()
    m.set_dropout_stream(seed, exact_next=True)
    x = R.normalise_sprites(R.closed_form_sprites(B))
    out = m(x.cuda())
    torch.cuda.synchronize()
    assert m.last_drop_seed == seed and m.last_path(B) == 2
    with torch.no_grad():
        ref, new_stats = T.teacher_forward(x, S, training=True, masks=D.TeacherMasks(seed, p, B))
    tol = {"quality_scores": 2e-3, "expert_weights": 2e-3, "style_embedding": 2e-2, "prompt_embedding": 2e-2, "semantic_score": 2e-3}
    for k, t in tol.items():
        got = out[k].detach().cpu()   # the outputs carry a graph since round 3 (like the reference module's)
        d = (got - ref[k]).abs().max().item()
        print("dropout", k, d, "(masks move it by", float(g["nodrop_delta/" + k]), ")")
        assert d <= t, (k, d)
        assert np.abs(got.numpy() - g["train/" + k]).max() <= t, k
    sd = m.state_dict()
    for k in ("feature_extractor.fusion.2.running_mean", "experts.0.0.conv2.2.running_var", "experts.3.2.conv2.2.running_var"):
        d = np.abs(sd[k].cpu().numpy() - g["train/" + k]).max()
        assert d <= 2e-3 * max(1.0, np.abs(g["train/" + k]).max()), (k, d)


@pytest.mark.gpu
def test_teacher_dropout_modes_and_streams():
    """eval mode = identity (and the shortcut path); dropout_rate 0 in train mode = shortcut path; with p > 0 the shortcut path
    is not taken; consecutive calls and different torch seeds draw different masks; the same stream repeats bitwise."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    S = T.closed_form_teacher_state()
    x = R.normalise_sprites(R.closed_form_sprites(2)).cuda()

    def fresh(p):
        t = LunarMoETeacher(dropout_rate=p); t.load_state_dict(S)
        return t.to("cuda")
    t0, t1 = fresh(0.0).eval(), fresh(0.1).eval()
    a, b = t0(x), t1(x)
    assert torch.equal(a["quality_scores"], b["quality_scores"]) and torch.equal(a["style_embedding"], b["style_embedding"])
    assert t1.last_path(2) == 0 and t1.last_drop_seed is None
    tz = fresh(0.0).train(); tz(x)
    assert tz.last_path(2) == 0                      # p = 0: constant-field shortcuts
    t1.train()
    o1 = t1(x)["style_embedding"].clone(); s1 = t1.last_drop_seed
    o2 = t1(x)["style_embedding"].clone(); s2 = t1.last_drop_seed
    assert t1.last_path(2) == 2 and s1 != s2         # p > 0: every conv in full, a new call seed per forward
    assert (o1 - o2).abs().max().item() > 1e-3       # different masks (BatchNorm running stats do not enter a train-mode output)
    t2 = fresh(0.1).train(); t2.set_dropout_stream(s1, exact_next=True)
    assert torch.equal(t2(x)["style_embedding"], o1)                                   # same stream -> same bits
    torch.manual_seed(1234)
    t3 = fresh(0.1).train(); t3(x)
    torch.manual_seed(4321)
    t4 = fresh(0.1).train(); t4(x)
    assert t3.last_drop_seed != t4.last_drop_seed    # keyed by the torch seed (train_hybrid.py:1138-1141)


@pytest.mark.gpu
def test_teacher_head_gradients_with_dropout_match_autograd_of_oracle():
    """A13 with the reference's default dropout: the backward replays the gate / quality-head masks of its forward."""
    import ctypes as C

    from lunaris_orion_amd import _lib
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from oracle import dropout_ref as D
    B, seed, p = 2, 0xABCDEF0123456789, 0.1
    S = T.closed_form_teacher_state()
    m = LunarMoETeacher(); m.load_state_dict(S); m = m.to("cuda").train()
    m.set_dropout_stream(seed, exact_next=True)
    x = R.normalise_sprites(R.closed_form_sprites(B))
    out = m(x.cuda())
    h, ws, _ = m._engine(B)
    b, e = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.lib.lo_teacher_grad_range(h, C.byref(b), C.byref(e)))
    rows = torch.empty(B * (e.value - b.value), device="cuda")
    grads = torch.zeros_like(m._flat)
    _lib.check(_lib.lib.lo_teacher_heads_backward(h, m._flat.data_ptr(), ws.data_ptr(), out["expert_weights"].data_ptr(), 0.5,
                                                  rows.data_ptr(), grads.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    S2 = {k: (v.clone().requires_grad_(True) if (k.startswith("gate.") or k.startswith("quality_heads.")) else v) for k, v in S.items()}
    o, _ = T.teacher_forward(x, S2, training=True, masks=D.TeacherMasks(seed, p, B))
    (0.5 * (-o["quality_scores"].mean())).backward()
    for k, t in m._named_state():
        if not (k.startswith("gate.") or k.startswith("quality_heads.")):
            continue
        off = (t.data_ptr() - m._flat.data_ptr()) // 4
        gg = grads[off: off + t.numel()].view(t.shape).cpu()
        ref = S2[k].grad
        assert (gg - ref).norm().item() / (ref.norm().item() + 1e-12) <= 2e-2, k


@pytest.mark.gpu
def test_hybrid_steps_match_the_references_own_process_batch_with_dropout():
    """Two full hybrid steps at the reference's DEFAULTS (teacher dropout 0.1, train mode) against the 12-metric trace the
    reference's own `TrainingManager._process_batch` produced (oracle/make_golden.py::run_reference_loop: `train_hybrid.main()`
    unmodified under the tensorboard / DataLoader-timeout shims, its dropout modules hooked to the masks of the call seeds
    recorded in the fixture).  The native stepper draws the same masks from the same seed stream."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from lunaris_orion_amd.trainer import HybridStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    g = np.load(os.path.join(GOLD, "hybrid_loop_L256_B2.npz"))
    cols, trace, seeds = [str(c) for c in g["cols"]], g["trace"], [int(v) for v in g["call_seeds"]]
    L, B, steps = (int(v) for v in g["meta"])
    assert len(seeds) == 2 * steps and float(g["drop_p"]) == 0.1
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    vae = LunarisCoreVAE(L); vae.load_state_dict(R.closed_form_params(L)); vae = vae.to("cuda")
    t = LunarMoETeacher(); t.load_state_dict(T.closed_form_teacher_state()); t = t.to("cuda").train()     # dropout_rate 0.1
    t.set_dropout_stream(seeds[0], exact_next=True)
    hs = HybridStepper(vae, t, gradient_accumulation_steps=1)
    tol = {"recon_loss": ("rel", 2e-4), "kl_loss": ("rel", 2e-4), "vae_loss": ("rel", 1e-3), "pg_loss": ("abs", 5e-4), "advantage": ("abs", 5e-4)}
    for s in range(steps):
        hs.step(x, s, R.closed_form_eps(B, L, salt=s).cuda())
        assert t.last_drop_seed == seeds[2 * s + 1] and t.last_path(B) == 2       # same stream as the fixture, dropout path
        m = hs.metrics()
        for j, c in enumerate(cols):
            kind, lim = tol.get(c, ("abs", 3e-3))
            err = abs(m[c] - trace[s, j]) / (abs(trace[s, j]) if kind == "rel" else 1.0)
            assert err <= lim, (s, c, m[c], trace[s, j])
    assert int(g["teacher_params_with_grad"]) == 28
    np.testing.assert_allclose(t.gate[2].weight.detach().cpu().numpy()[:4, :8], g["gate_w_after"], atol=2e-4)
    np.testing.assert_allclose(vae.encoder.fc_mu.bias.detach().cpu().numpy()[:16], g["vae_fc_mu_b_after"], atol=2e-4)
    assert abs(hs.lr - float(g["lr_after"])) <= 1e-12


# ---- feature_dim 256 / 512 (README High-End recipe) -------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("F,B", [(256, 2), (512, 1)])
def test_wide_teacher_forward_matches_reference_fixture_and_oracle(F, B):
    """LunarMoETeacher(feature_dim=F, embedding_dim=256): eval mode and train mode with the default dropout (injected masks)
    against the fixture the REFERENCE produced (tests/golden/teacher_F{F}_B{B}.npz) and the oracle's functions (on device tensors:
    the CPU evaluation of these two shapes took 90 s of the suite; the fixture is the pin); BatchNorm running statistics of the
    shortcut / conv layers included.  Same tolerances as feature_dim 128."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from oracle import dropout_ref as D
    g = np.load(os.path.join(GOLD, f"teacher_F{F}_B{B}.npz"))
    assert [int(v) for v in g["meta"]] == [B, 4, F, 256]
    seed, p = int(g["drop_seed"]), float(g["drop_p"])
    S = T.closed_form_teacher_state(feature_dim=F, embedding_dim=256)
    x = R.normalise_sprites(R.closed_form_sprites(B))
    tol = {"quality_scores": 2e-3, "expert_weights": 2e-3, "style_embedding": 2e-2, "prompt_embedding": 2e-2, "semantic_score": 2e-3}
    for training in (False, True):
        m = LunarMoETeacher(num_experts=4, feature_dim=F, embedding_dim=256)
        m.load_state_dict(S)
        m = m.to("cuda")
        m.train(training)
        m.set_dropout_stream(seed, exact_next=True)
        out = m(x.cuda())
        torch.cuda.synchronize()
        tag = "train" if training else "eval"
        assert m.last_path(B) == (2 if training else 1)
        from tests.hip_helpers import oracle_teacher_on_device
        ref, new_stats = oracle_teacher_on_device(x, S, training, seed, p if training else 0.0)
        for k, t in tol.items():
            got = out[k].detach().cpu()   # the outputs carry a graph since round 3 (like the reference module's)
            d = (got - ref[k]).abs().max().item()
            print(f"F={F} {tag}", k, d)
            assert d <= t, (tag, k, d)
            assert np.abs(got.numpy() - g[f"{tag}/{k}"]).max() <= t, (tag, k)
        if training:
            sd = m.state_dict()
            for k in ("experts.0.0.shortcut.1.running_var", "experts.3.2.conv2.2.running_var", "experts.1.1.conv1.2.running_mean"):
                ref_s = g["train/" + k]
                assert np.abs(sd[k].cpu().numpy() - ref_s).max() <= 2e-3 * max(1.0, np.abs(ref_s).max()), k
            assert int(sd["experts.2.0.shortcut.1.num_batches_tracked"]) == 1


@pytest.mark.gpu
def test_wide_teacher_head_gradients_and_hybrid_step():
    """feature_dim 256: gate / quality-head gradients (LayerNorm over 256 features) vs autograd of the oracle with the same
    dropout masks, then one full hybrid step through HybridStepper."""
    import ctypes as C

    from lunaris_orion_amd import _lib
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from lunaris_orion_amd.trainer import HybridStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    from oracle import dropout_ref as D
    F, B, seed, p = 256, 2, 0x0F0F0F0F12345678, 0.1
    S = T.closed_form_teacher_state(feature_dim=F, embedding_dim=256)
    m = LunarMoETeacher(feature_dim=F, embedding_dim=256); m.load_state_dict(S); m = m.to("cuda").train()
    m.set_dropout_stream(seed, exact_next=True)
    x = R.normalise_sprites(R.closed_form_sprites(B))
    out = m(x.cuda())
    h, ws, _ = m._engine(B)
    b, e = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.lib.lo_teacher_grad_range(h, C.byref(b), C.byref(e)))
    rows = torch.empty(B * (e.value - b.value), device="cuda")
    grads = torch.zeros_like(m._flat)
    _lib.check(_lib.lib.lo_teacher_heads_backward(h, m._flat.data_ptr(), ws.data_ptr(), out["expert_weights"].data_ptr(), 0.5,
                                                  rows.data_ptr(), grads.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    S2 = {k: (v.clone().requires_grad_(True) if (k.startswith("gate.") or k.startswith("quality_heads.")) else v) for k, v in S.items()}
    o, _ = T.teacher_forward(x, S2, training=True, masks=D.TeacherMasks(seed, p, B))
    (0.5 * (-o["quality_scores"].mean())).backward()
    for k, t in m._named_state():
        if not (k.startswith("gate.") or k.startswith("quality_heads.")):
            continue
        off = (t.data_ptr() - m._flat.data_ptr()) // 4
        gg = grads[off: off + t.numel()].view(t.shape).cpu()
        ref = S2[k].grad
        assert (gg - ref).norm().item() / (ref.norm().item() + 1e-12) <= 2e-2, k
    L = 256
    vae = LunarisCoreVAE(L); vae.load_state_dict(R.closed_form_params(L)); vae = vae.to("cuda")
    hs = HybridStepper(vae, m, gradient_accumulation_steps=1)
    hs.step(x.cuda(), 0, R.closed_form_eps(B, L, 0).cuda())
    met = hs.metrics()
    assert np.isfinite(list(met.values())).all() and met["grads_finite"] == 1.0 and 0.3 < met["quality_scores"] < 0.7


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3])
def test_teacher_dropout_odd_batch_sizes(B):
    """Train mode with the default dropout at a single sprite and an odd batch (mask indexing, per-sample tables) vs the oracle."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    from oracle import dropout_ref as D
    seed = 0x00C0FFEE00C0FFEE + B
    S = T.closed_form_teacher_state()
    m = LunarMoETeacher(); m.load_state_dict(S); m = m.to("cuda").train()
    m.set_dropout_stream(seed, exact_next=True)
    x = R.normalise_sprites(R.closed_form_sprites(B))
    out = m(x.cuda())
    torch.cuda.synchronize()
    with torch.no_grad():
        ref, _ = T.teacher_forward(x, S, training=True, masks=D.TeacherMasks(seed, 0.1, B))
    for k, t in {"quality_scores": 2e-3, "expert_weights": 2e-3, "semantic_score": 2e-3, "style_embedding": 2e-2}.items():
        assert (out[k].cpu() - ref[k]).abs().max().item() <= t, k


@pytest.mark.gpu
def test_statistics_only_call_with_dropout_has_the_side_effects_of_forward():
    """`_process_batch`'s dead first teacher call under the reference's default dropout: same mask stream -> the BatchNorm running
    statistics a statistics-only call leaves are bitwise those of the full forward (it skips only the last tails, pooling, heads)."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    S = T.closed_form_teacher_state()
    x = R.normalise_sprites(R.closed_form_sprites(2)).cuda()
    states = []
    for stats_only in (False, True):
        t = LunarMoETeacher(); t.load_state_dict(S); t = t.to("cuda").train()
        t.set_dropout_stream(77, exact_next=True)
        t.update_statistics_only(x) if stats_only else t(x)
        torch.cuda.synchronize()
        assert t.last_path(2) == 2
        states.append({k: v.detach().cpu().clone() for k, v in t.state_dict().items()})
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k


def test_oracle_autograd_reproduces_the_reference_full_backward_fixture():
    """SURVEY §8 row F2, second half.  tests/golden/teacher_fullgrad_drop_B2.npz holds the gradients of the REFERENCE teacher for its own
    teacher loss with `torch.utils.checkpoint.checkpoint` made non-reentrant (oracle/make_golden.py run_teacher_fullgrad: the reference's
    modules, its dropout modules on injected masks).  Autograd through the oracle's plain restatement must give the same numbers:
    this pins the oracle that `lo_teacher_full_backward` is tested against on the GPU (tests/test_teacher_fullgrad_gpu.py)."""
    import os
    import numpy as np
    from oracle import dropout_ref as D
    from oracle import vae_ref as R
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "teacher_fullgrad_drop_B2.npz"))
    B = int(z["meta"][0])
    assert int(z["meta"][4]) == 1 and int(z["n_with_grad"]) == 234 and int(z["n_params"]) == 252
    seed, p = 0x5EEDD209C0FFEE11, 0.1
    x = R.normalise_sprites(R.closed_form_sprites(B))
    S = T.closed_form_teacher_state()
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "last_spatial" not in k else v) for k, v in S.items()}
    out, _ = T.teacher_forward(x, P, training=True, masks=D.TeacherMasks(seed, p, B))
    loss = float(z["quality_weight"]) * -torch.mean(out["quality_scores"])
    assert abs(loss.item() - float(z["loss"])) <= 1e-6
    names = [k for k, v in P.items() if v.requires_grad]
    grads = dict(zip(names, torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)))
    n = 0
    for k, g in grads.items():
        tag = f"tgrad/{k}"
        if tag + "/l2" not in z:
            # no gradient in the reference either (the heads the loss does not read), or the softmax-invariant relative-position tables
            assert g is None, k
            continue
        n += 1
        l2 = float(z[tag + "/l2"])
        assert abs(g.double().norm().item() - l2) <= 2e-3 * l2 + 1e-12, k
        idx = ((R.closed_form_uniform("sample." + tag, min(2048, g.numel())) + 1.0) * 0.5 * g.numel()).long().clamp_(0, g.numel() - 1)
        ref = torch.from_numpy(z[tag + "/samples"])
        assert (g.flatten()[idx] - ref).norm().item() <= 2e-3 * ref.norm().item() + 1e-9, k
    assert n == 234 - 24
