"""SURVEY §8 row F2, second half: `lo_teacher_full_backward` -- gradients for every teacher parameter on the path of the teacher
loss, experts and feature extractor included (lunar_evaluator.py:194-197, 266-275, 411-414 with non-reentrant checkpoints;
train_hybrid.py:891-904).  Checked against
  * the reference itself: tests/golden/teacher_fullgrad_{,drop_}B2.npz (oracle/make_golden.py run_teacher_fullgrad: the reference's
    modules, `checkpoint` made non-reentrant in-process, its own dropout modules on injected masks) -- sums, norms and 2048 sampled
    entries of all 234 gradients;
  * autograd of the CPU oracle (oracle/teacher_ref.py) on the same inputs, every tensor in full.
Tolerance: 4e-2 of the tensor's norm, and cosine >= 0.999 with the norm within 1 % for every tensor that carries a visible share of
the total.  What the tolerance covers was measured (tools/README.md): the deviation is unbiased noise (cosine 0.9997-1.0000,
norm ratio 1 +- 0.005; independent of the gradient scale 2^17 ... 2^25), 0.1-0.5 % in the feature extractor, BatchNorm and layer-scale
gradients and 1-2.6 % in the conv / attention weights of the expert blocks.  Its source is LeakyReLU's kink under fp16 activations:
about 1 in 1000 conv outputs lies close enough to zero that its fp16 value has the other sign than the fp32 oracle's, and each such
element carries a slope of 1 instead of 0.2 (or the reverse) -- sqrt(1e-3) = 3 % in L2 on the element gradients.  The reference under
its own --mixed_precision has the same property.  Tensors whose gradient is mathematically zero are compared on an absolute scale.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dropout_ref as D
from oracle import teacher_ref as T
from oracle import vae_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DROP_SEED, DROP_P, QW = 0x5EEDD209C0FFEE11, 0.1, 0.5


def _teacher(drop, F=128, emb=64):
    from lunaris_orion_amd.teacher import LunarMoETeacher
    kw = {} if F == 128 else {"feature_dim": F, "embedding_dim": emb}
    m = LunarMoETeacher(num_experts=4, feature_dim=F, embedding_dim=emb, dropout_rate=DROP_P if drop else 0.0)
    m.load_state_dict(T.closed_form_teacher_state(**kw))
    m = m.to("cuda")
    m.train()
    return m


_ORACLE_CACHE = {}


def _oracle_grads(x, drop, device="cpu", **kw):
    key = (tuple(x.shape), bool(drop), device, tuple(sorted(kw.items())))
    if key not in _ORACLE_CACHE:              # inputs are closed-form: the same (shape, dropout) always means the same numbers
        _ORACLE_CACHE[key] = _oracle_grads_uncached(x, drop, device, **kw)
    return _ORACLE_CACHE[key]


def _oracle_grads_uncached(x, drop, device="cpu", **kw):
    S = T.closed_form_teacher_state(**kw)
    P = {k: (v.to(device).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "last_spatial" not in k else v.to(device))
         for k, v in S.items()}
    masks = D.TeacherMasks(DROP_SEED, DROP_P, x.shape[0], device=device) if drop else None
    out, _ = T.teacher_forward(x.to(device), P, training=True, masks=masks)
    loss = QW * -torch.mean(out["quality_scores"])
    names = [k for k, v in P.items() if v.requires_grad]
    g = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
    return {k: (None if t is None else t.detach().cpu()) for k, t in zip(names, g)}, loss.item()


def _native_grads(m, x, drop, keep=False):
    if drop:
        m.set_dropout_stream(DROP_SEED, exact_next=True)
    with torch.no_grad():
        out = m._native_forward(x, keep=True)[0] if keep else m(x)       # keep: lo_teacher_forward_keep, no first pass in the backward
    flat = m.full_backward(x, out["expert_weights"], QW)
    torch.cuda.synchronize()
    return {k: v.detach().cpu() for k, v in m.parameter_grad_views(flat).items()}, out


def _zero_by_construction(k):
    # a conv bias in front of a train-mode BatchNorm shifts the batch mean and nothing else -- but only where no LeakyReLU sits
    # between them: in this model every such conv is followed by LeakyReLU, except the shortcut conv (feature_dim != 128)
    return k.endswith("shortcut.0.bias")


@pytest.mark.parametrize("drop,keep", [(False, False), (True, False), (True, True)], ids=["no_dropout", "dropout_0.1", "dropout_0.1_kept_forward"])
def test_full_backward_matches_the_reference_fixture_and_the_oracle(drop, keep):
    """keep: the step's forward is lo_teacher_forward_keep (plain expert path, block outputs left in the backward's scratch: what
    HybridStepper(teacher_full_backward=True) runs); otherwise the production forward, and the backward recomputes them itself."""
    B = 2
    x = R.normalise_sprites(R.closed_form_sprites(B))
    m = _teacher(drop)
    got, out = _native_grads(m, x.cuda(), drop, keep)
    if keep:
        # the kept forward is the same function as the production one: outputs against the oracle's at the forward tolerances
        oo, _ = T.teacher_forward(x, T.closed_form_teacher_state(), training=True, masks=D.TeacherMasks(DROP_SEED, DROP_P, B))
        assert (out["quality_scores"].cpu() - oo["quality_scores"]).abs().max().item() <= 2e-3
        assert (out["style_embedding"].cpu() - oo["style_embedding"]).abs().max().item() <= 2e-2
    ora, _ = _oracle_grads(x, drop)
    z = np.load(os.path.join(GOLD, f"teacher_fullgrad_{'drop_' if drop else ''}B2.npz"))
    assert int(z["n_with_grad"]) == 234
    worst = {}
    n_checked = 0
    for k, g in got.items():
        o = ora.get(k)
        if o is None:
            # off the loss path (style / prompt / semantic heads) or the softmax-invariant relative-position tables: exactly zero here
            assert g.abs().max().item() == 0.0, k
            continue
        n_checked += 1
        on, d = o.norm().item(), (g - o).norm().item()
        scale = max(on, 1e-3 * float(z["total_norm"]))
        worst[k] = d / scale
        assert d <= 4e-2 * scale, (k, d, on)
        if on >= 1e-3 * float(z["total_norm"]):
            cos = torch.nn.functional.cosine_similarity(g.flatten().double(), o.flatten().double(), dim=0).item()
            assert cos >= 0.999 and abs(g.norm().item() / on - 1.0) <= 1e-2, (k, cos, g.norm().item() / on)
        # the reference's own numbers (sampled entries)
        tag = f"tgrad/{k}"
        idx = ((R.closed_form_uniform("sample." + tag, min(2048, g.numel())) + 1.0) * 0.5 * g.numel()).long().clamp_(0, g.numel() - 1)
        ref = torch.from_numpy(z[tag + "/samples"])
        ds = (g.flatten()[idx] - ref).norm().item()
        assert ds <= 4e-2 * max(ref.norm().item(), 1e-3 * float(z["total_norm"]) * (len(idx) / g.numel()) ** 0.5) + 1e-12, (k, ds, ref.norm().item())
    assert n_checked == 234 - 24          # 24 relative-position tables are off the oracle's graph
    tot = torch.sqrt(sum((g.double() ** 2).sum() for g in got.values())).item()
    assert abs(tot - float(z["total_norm"])) <= 1e-2 * float(z["total_norm"]), (tot, float(z["total_norm"]))
    print("worst relative errors:", sorted(worst.items(), key=lambda kv: -kv[1])[:8])


@pytest.mark.parametrize("F", [256, 512])
def test_full_backward_wide_teacher_shortcut_branch_matches_the_oracle(F):
    """feature_dim != 128 (512 = the README High-End recipe): the first block of every expert has the Conv1x1 + BatchNorm shortcut
    (lunar_evaluator.py:254-257), 32- / 64-wide heads; the oracle's autograd runs on the device (same functions, ATen fp32 kernels)."""
    B, emb = 1, 256
    x = R.normalise_sprites(R.closed_form_sprites(B))
    m = _teacher(True, F, emb)
    got, _ = _native_grads(m, x.cuda(), True)
    prev = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    try:
        ora, _ = _oracle_grads(x, True, device="cuda", feature_dim=F, embedding_dim=emb)
    finally:
        torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = prev
    tot = torch.sqrt(sum((o.double() ** 2).sum() for o in ora.values() if o is not None)).item()
    n = 0
    for k, g in got.items():
        o = ora.get(k)
        if o is None:
            assert g.abs().max().item() == 0.0, k
            continue
        n += 1
        d, on = (g - o).norm().item(), o.norm().item()
        if _zero_by_construction(k):
            assert on <= 1e-4 * tot and g.norm().item() <= 2e-3 * tot, (k, on, g.norm().item())
            continue
        # one sample instead of two and twice the channels: the kink noise of the module docstring is 1.5x larger here (measured
        # worst 4.3 %); a missing or mis-scaled term would show in the cosine / the norm ratio, which stay tight
        assert d <= 6e-2 * max(on, 1e-3 * tot), (k, d, on)
        if on >= 1e-3 * tot:
            cos = torch.nn.functional.cosine_similarity(g.flatten().double(), o.flatten().double(), dim=0).item()
            assert cos >= 0.998 and abs(g.norm().item() / on - 1.0) <= 1e-2, (k, cos, g.norm().item() / on)
    assert n == 234 + 4 * 4 - 24          # + shortcut conv weight / bias and BatchNorm weight / bias per expert


def test_full_parameter_update_touches_exactly_what_the_reference_optimizer_would():
    """lo_teacher_clip_adamw_full against torch.optim.AdamW fed with the SAME gradients (clip_grad_norm_ first), two steps: every tensor
    with a gradient moves as AdamW says (the relative-position tables by weight decay alone), BatchNorm buffers and the three heads
    the loss does not read stay bit-identical, and the forward after the update sees the new weights (operand re-pack)."""
    import ctypes as C
    from lunaris_orion_amd import _lib
    B, lr, wd, max_norm = 2, 1e-3, 0.01, 0.05
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    m = _teacher(True)
    names = [k for k, _ in m.named_parameters()]
    off_path = lambda k: k.split(".")[0] in ("semantic_head", "style_net", "prompt_net")
    ref_p = {k: p.detach().clone().requires_grad_(True) for k, p in m.named_parameters() if not off_path(k)}
    opt = torch.optim.AdamW(list(ref_p.values()), lr=lr, weight_decay=wd, betas=(0.9, 0.999), eps=1e-8)
    eng = m._engine(B)
    nt, ne = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.lib.lo_teacher_full_param_count(eng.handle, C.byref(nt), C.byref(ne)), "count")
    assert nt.value == len(ref_p) == 234 and ne.value == sum(p.numel() for p in ref_p.values())
    mm, vv = torch.zeros_like(m._flat), torch.zeros_like(m._flat)
    scratch = torch.zeros(1028, device="cuda")
    state0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    q_before = None
    for step in (1, 2):
        m.set_dropout_stream(DROP_SEED + step, exact_next=True)
        with torch.no_grad():
            out = m(x)
        if q_before is None:
            q_before = out["quality_scores"].clone()
        stats = {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k}
        flat = m.full_backward(x, out["expert_weights"], QW)
        gv = m.parameter_grad_views(flat)
        for k, p in ref_p.items():
            p.grad = gv[k].detach().clone()
        torch.nn.utils.clip_grad_norm_(list(ref_p.values()), max_norm)
        opt.step()
        _lib.check(_lib.lib.lo_teacher_clip_adamw_full(eng.handle, m._flat.data_ptr(), flat.data_ptr(), mm.data_ptr(), vv.data_ptr(), max_norm, lr,
                                                       0.9, 0.999, 1e-8, wd, step, scratch.data_ptr(), _lib.stream_ptr()), "clip_adamw_full")
        m.mark_weights_changed()
        torch.cuda.synchronize()
        for k, v in m.state_dict().items():
            if "running" in k:
                assert torch.equal(v, stats[k]), k                                   # buffers are not parameters
    sd = m.state_dict()
    for k in names:
        if off_path(k):
            assert torch.equal(sd[k], state0[k]), k
            continue
        a, b = sd[k], ref_p[k].detach()
        assert (a - b).abs().max().item() <= 2e-6 + 1e-5 * b.abs().max().item(), (k, (a - b).abs().max().item())
        assert not torch.equal(a, state0[k]), k                                       # every live tensor moved
    with torch.no_grad():
        m.set_dropout_stream(DROP_SEED + 1, exact_next=True)
        q_after = m(x)["quality_scores"]
    assert torch.isfinite(q_after).all() and not torch.equal(q_after, q_before)
    assert q_after.mean().item() > q_before.mean().item()                             # the loss is -mean(quality_scores): two steps up


def test_hybrid_stepper_with_teacher_full_backward_runs_the_whole_step():
    """`--teacher_full_backward` end to end: HybridStepper(teacher_full_backward=True), three steps of the full hybrid _process_batch;
    the expert and feature-extractor weights move, the metrics stay finite, and the default (flag off) leaves them alone."""
    from lunaris_orion_amd.trainer import HybridStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    B, L = 2, 256
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    moved = {}
    for full in (False, True):
        vae = LunarisCoreVAE(L)
        vae.load_state_dict(R.closed_form_params(L))
        vae = vae.to("cuda")
        t = _teacher(True)
        t.set_dropout_stream(DROP_SEED)
        w0 = {k: v.detach().clone() for k, v in t.state_dict().items()}
        st = HybridStepper(vae, t, teacher_lr=1e-4, teacher_full_backward=full, lr=1e-4)
        for s in range(3):
            st.step(x, s, R.closed_form_eps(B, L, salt=s).cuda())
        torch.cuda.synchronize()
        met = st.metrics()
        assert all(np.isfinite(v) for v in met.values()), met
        sd = t.state_dict()
        moved[full] = {k: not torch.equal(sd[k], w0[k]) for k in w0 if "running" not in k and "num_batches" not in k and "last_spatial" not in k}
    assert moved[False]["gate.2.weight"] and not moved[False]["experts.0.0.conv1.0.weight"] and not moved[False]["feature_extractor.conv1.0.weight"]
    for k, mv in moved[True].items():
        assert mv == (k.split(".")[0] not in ("semantic_head", "style_net", "prompt_net")), k


def test_module_with_full_backward_fills_every_grad_through_autograd():
    """LunarMoETeacher(full_backward=True) in a foreign training loop: `teacher_loss.backward()` (train_hybrid.py:891-904) leaves a .grad on
    every parameter on the path, equal to the reference's (fixture); multiplied by a GradScaler-style 65 536 the gradients come out
    multiplied by exactly 65 536 (the upstream normalisation is a power of two picked on the device)."""
    from lunaris_orion_amd.teacher import LunarMoETeacher
    B = 2
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    z = np.load(os.path.join(GOLD, "teacher_fullgrad_drop_B2.npz"))
    grads = {}
    for scale in (1.0, 65536.0):
        m = LunarMoETeacher(num_experts=4, feature_dim=128, embedding_dim=64, dropout_rate=DROP_P, full_backward=True)
        m.load_state_dict(T.closed_form_teacher_state())
        m = m.to("cuda").train()
        m.set_dropout_stream(DROP_SEED, exact_next=True)
        out = m(x)
        loss = QW * -torch.mean(out["quality_scores"])
        (loss * scale).backward()
        torch.cuda.synchronize()
        grads[scale] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in m.named_parameters()}
    n = 0
    for k, g in grads[1.0].items():
        if k.split(".")[0] in ("semantic_head", "style_net", "prompt_net"):
            assert g is None, k
            continue
        assert g is not None, k
        assert torch.equal(grads[65536.0][k], g * 65536.0), k
        tag = f"tgrad/{k}"
        if tag + "/samples" not in z:
            assert g.abs().max().item() == 0.0, k         # relative-position tables
            continue
        n += 1
        g = g.cpu()
        idx = ((R.closed_form_uniform("sample." + tag, min(2048, g.numel())) + 1.0) * 0.5 * g.numel()).long().clamp_(0, g.numel() - 1)
        ref = torch.from_numpy(z[tag + "/samples"])
        ds = (g.flatten()[idx] - ref).norm().item()
        assert ds <= 4e-2 * max(ref.norm().item(), 1e-3 * float(z["total_norm"]) * (len(idx) / g.numel()) ** 0.5) + 1e-12, (k, ds, ref.norm().item())
    assert n == 210


def test_full_backward_is_bitwise_reproducible():
    """Every reduction of the full backward has a fixed order (per-block partial rows + ordered sums, one writer per k / v row in the
    attention, no float atomics): two fresh engines give bit-identical gradients, with the kept forward and with the recomputed one."""
    B = 2
    x = R.normalise_sprites(R.closed_form_sprites(B)).cuda()
    for keep in (False, True):
        runs = []
        for _ in range(2):
            got, _ = _native_grads(_teacher(True), x, True, keep)
            runs.append(got)
        for k in runs[0]:
            assert torch.equal(runs[0][k], runs[1][k]), (keep, k)


def test_full_backward_at_batch64_matches_the_oracle_on_the_device():
    """BASELINE's full batch: every gradient of the teacher's full backward at batch 64 (the shape bench.py times) against autograd of the
    oracle evaluated on the device (the same oracle/teacher_ref.py functions on ATen fp32 kernels; ~110 GB of saved activations: sized
    for the 288 GB of this GPU), default dropout on the library's masks, the step's kept forward."""
    B = 64
    x = R.normalise_sprites(R.closed_form_sprites(B))
    got, out = _native_grads(_teacher(True), x.cuda(), True, keep=True)
    torch.cuda.empty_cache()
    prev = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    try:
        ora, _ = _oracle_grads_uncached(x, True, device="cuda")
    finally:
        torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = prev
    torch.cuda.empty_cache()
    tot = torch.sqrt(sum((o.double() ** 2).sum() for o in ora.values() if o is not None)).item()
    worst, n = {}, 0
    for k, g in got.items():
        o = ora.get(k)
        if o is None:
            assert g.abs().max().item() == 0.0, k
            continue
        n += 1
        d, on = (g - o).norm().item(), o.norm().item()
        worst[k] = d / max(on, 1e-3 * tot)
        assert d <= 4e-2 * max(on, 1e-3 * tot), (k, d, on)
        if on >= 1e-3 * tot:
            cos = torch.nn.functional.cosine_similarity(g.flatten().double(), o.flatten().double(), dim=0).item()
            assert cos >= 0.999 and abs(g.norm().item() / on - 1.0) <= 1e-2, (k, cos, g.norm().item() / on)
    assert n == 210
    print("batch 64 worst relative errors:", sorted(worst.items(), key=lambda kv: -kv[1])[:5])
