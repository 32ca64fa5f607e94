"""SURVEY §8 row F2, second half: `lo_teacher_full_backward` -- gradients for every teacher parameter on the path of the teacher
loss, experts and feature extractor included (lunar_evaluator.py:194-197, 266-275, 411-414 with non-reentrant checkpoints;
train_hybrid.py:891-904).  Checked against
  * the reference itself: tests/golden/teacher_fullgrad_{,drop_}B2.npz (oracle/make_golden.py run_teacher_fullgrad: the reference's
    modules, `checkpoint` made non-reentrant in-process, its own dropout modules on injected masks) -- sums, norms and 2048 sampled
    entries of all 234 gradients;
  * autograd of the CPU oracle (oracle/teacher_ref.py) on the same inputs, every tensor in full.
Tolerance: fp16 operands / fp16 activation gradients through 3 blocks of conv + BatchNorm: 3e-2 of the tensor's norm for the big
tensors; tiny tensors (biases in front of a BatchNorm have a mathematically ZERO gradient: what is left is rounding noise) are
compared on an absolute scale tied to the largest gradient norm of the layer kind.
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


def _oracle_grads(x, drop, device="cpu", **kw):
    S = T.closed_form_teacher_state(**kw)
    P = {k: (v.to(device).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "last_spatial" not in k else v.to(device))
         for k, v in S.items()}
    masks = D.TeacherMasks(DROP_SEED, DROP_P, x.shape[0], device=device) if drop else None
    out, _ = T.teacher_forward(x.to(device), P, training=True, masks=masks)
    loss = QW * -torch.mean(out["quality_scores"])
    names = [k for k, v in P.items() if v.requires_grad]
    g = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
    return {k: (None if t is None else t.detach().cpu()) for k, t in zip(names, g)}, loss.item()


def _native_grads(m, x, drop):
    if drop:
        m.set_dropout_stream(DROP_SEED, exact_next=True)
    with torch.no_grad():
        out = m(x)
    flat = m.full_backward(x, out["expert_weights"], QW)
    torch.cuda.synchronize()
    return {k: v.detach().cpu() for k, v in m.parameter_grad_views(flat).items()}, out


def _zero_by_construction(k):
    # a conv bias in front of a train-mode BatchNorm shifts the batch mean and nothing else -- but only where no LeakyReLU sits
    # between them: in this model every such conv is followed by LeakyReLU, except the shortcut conv (feature_dim != 128)
    return k.endswith("shortcut.0.bias")


@pytest.mark.parametrize("drop", [False, True], ids=["no_dropout", "dropout_0.1"])
def test_full_backward_matches_the_reference_fixture_and_the_oracle(drop):
    B = 2
    x = R.normalise_sprites(R.closed_form_sprites(B))
    m = _teacher(drop)
    got, out = _native_grads(m, x.cuda(), drop)
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
        assert d <= 3e-2 * scale, (k, d, on)
        # the reference's own numbers (sampled entries)
        tag = f"tgrad/{k}"
        idx = ((R.closed_form_uniform("sample." + tag, min(2048, g.numel())) + 1.0) * 0.5 * g.numel()).long().clamp_(0, g.numel() - 1)
        ref = torch.from_numpy(z[tag + "/samples"])
        ds = (g.flatten()[idx] - ref).norm().item()
        assert ds <= 3e-2 * max(ref.norm().item(), 1e-3 * float(z["total_norm"]) * (len(idx) / g.numel()) ** 0.5) + 1e-12, (k, ds, ref.norm().item())
    assert n_checked == 234 - 24          # 24 relative-position tables are off the oracle's graph
    tot = torch.sqrt(sum((g.double() ** 2).sum() for g in got.values())).item()
    assert abs(tot - float(z["total_norm"])) <= 1e-2 * float(z["total_norm"]), (tot, float(z["total_norm"]))
    print("worst relative errors:", sorted(worst.items(), key=lambda kv: -kv[1])[:8])
