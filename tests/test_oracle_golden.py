"""CPU tests (no GPU): the oracle restatement reproduces the golden vectors generated from the reference's own
classes (oracle/make_golden.py), so parity of the oracle stays pinned on every run, here and on the GPU box."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sample(t, tag):
    n = t.numel()
    u = R.closed_form_uniform("sample." + tag, min(2048, n))
    idx = ((u + 1.0) * 0.5 * n).long().clamp_(0, n - 1)
    return t.detach().flatten()[idx].numpy()


@pytest.mark.parametrize("L", [256, 512])
def test_oracle_forward_backward_step_match_golden(L):
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = np.load(os.path.join(GOLD, f"vae_L{L}_B2.npz"))
    Lm, B, steps = [int(v) for v in g["meta"]]
    assert Lm == L
    P = R.closed_form_params(L)
    x = R.normalise_sprites(R.closed_form_sprites(B))
    tr = R.OracleTrainer(P)
    for s in range(steps):
        eps = R.closed_form_eps(B, L, salt=s)
        acts = {}
        if s == 0:
            R.vae_forward(x, eps, tr.P, acts)
        o = tr.step(x, eps, 0.0)
        ref = g["trace"][s]
        assert abs(o["recon_loss"] - ref[0]) <= 2e-7
        assert abs(o["kl_loss"] - ref[1]) <= 2e-7
        assert abs(o["grad_norm"] - ref[3]) <= 1e-5 * ref[3]
        assert abs(o["lr"] - ref[4]) <= 1e-12
        if s == 0:
            assert np.abs(o["mu"].numpy() - g["mu"]).max() <= 2e-6
            assert np.abs(o["logvar"].numpy() - g["logvar"]).max() <= 2e-6
            assert np.abs(_sample(o["recon"], "recon") - g["recon/samples"]).max() <= 2e-6
            for k, gr in o["grads"].items():
                ref_s = g[f"grad/{k}/samples"]
                assert np.abs(_sample(gr, "grad/" + k) - ref_s).max() <= 1e-6 + 1e-4 * np.abs(ref_s).max(), k
            # per-layer activations of the reference (hooked module outputs) vs the restatement's intermediates
            name_map = {"encoder.down1.0": "encoder.down1.0.conv", "encoder.down1.1": "encoder.down1.0.gn",
                        "encoder.down1.3": "encoder.down1.3.out", "encoder.down4.3.conv2.0": "encoder.down4.3.conv2.conv",
                        "decoder.up1.0": "decoder.up1.0.conv", "decoder.up4.1": "decoder.up4.0.gn",
                        "decoder.final_conv": "decoder.final_conv"}
            for ref_name, my_name in name_map.items():
                ref_s = g[f"act/{ref_name}/samples"]
                assert np.abs(_sample(acts[my_name], "act/" + ref_name) - ref_s).max() <= 1e-5 * max(1.0, np.abs(ref_s).max()), ref_name
    for k, p in tr.P.items():
        ref_s = g[f"param_after{steps}/{k}/samples"]
        assert np.abs(_sample(p, f"param_after{steps}/{k}") - ref_s).max() <= 1e-6, k


def test_selfattention2d_oracle_matches_golden():
    g = np.load(os.path.join(GOLD, "selfattn2d.npz"))
    B, C, H, W = [int(v) for v in g["meta"]]
    sd = {}
    for k, shape in (("query_conv.weight", (C // 8, C, 1, 1)), ("query_conv.bias", (C // 8,)), ("key_conv.weight", (C // 8, C, 1, 1)),
                     ("key_conv.bias", (C // 8,)), ("value_conv.weight", (C, C, 1, 1)), ("value_conv.bias", (C,))):
        t = R.closed_form_tensor("attn." + k, shape)
        sd[k] = t if len(shape) > 1 else t * 0 + 0.05
    x = R.closed_form_tensor("attn.x", (B, C, H, W)) * 8.0
    y = R.self_attention_2d(x, sd["query_conv.weight"], sd["query_conv.bias"], sd["key_conv.weight"], sd["key_conv.bias"],
                            sd["value_conv.weight"], sd["value_conv.bias"], torch.tensor([0.7]))
    assert np.abs(y.numpy() - g["y"]).max() <= 1e-5


def test_closed_form_inputs_are_stable():
    """The closed-form generators are part of the fixture contract: pin a few values."""
    u = R.closed_form_uniform("pin", 4)
    assert u.dtype == torch.float64 and (u >= -1).all() and (u < 1).all()
    s = R.closed_form_sprites(2)
    assert s.dtype == torch.uint8 and tuple(s.shape) == (2, 128, 128, 3)
    x = R.normalise_sprites(s)
    assert tuple(x.shape) == (2, 3, 128, 128) and x.min() >= -1.0 and x.max() <= 1.0
    assert int(s.sum()) == int(R.closed_form_sprites(2).sum())
    e = R.closed_form_eps(64, 512)
    assert abs(e.mean().item()) < 0.02 and abs(e.std().item() - 1.0) < 0.02


def test_scheduler_restatement_matches_torch():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=1e-6)
    for k in range(75):
        assert abs(opt.param_groups[0]["lr"] - R.cosine_warm_restarts_lr(1e-4, 1e-6, 10, 2, k)) <= 1e-15
        opt.step()
        sch.step()


def test_teacher_oracle_matches_golden():
    """LunarMoETeacher restatement (oracle/teacher_ref.py) vs the fixture generated from the reference class."""
    from oracle import teacher_ref as T
    g = np.load(os.path.join(GOLD, "teacher_B2.npz"))
    B = int(g["meta"][0])
    S = T.closed_form_teacher_state()
    assert len(S) == 351 and sum(1 for k in S if "running" in k or "tracked" in k or "last_spatial" in k) == 99
    x = R.normalise_sprites(R.closed_form_sprites(B))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():
        out, new_stats = T.teacher_forward(x, S, training=True)
    for k in ("quality_scores", "expert_weights", "style_embedding", "prompt_embedding", "semantic_score"):
        assert np.abs(out[k].numpy() - g["train/" + k]).max() <= 2e-5, k
    assert np.abs(new_stats["feature_extractor.fusion.2.running_mean"].numpy() - g["train/feature_extractor.fusion.2.running_mean"]).max() <= 1e-5
    assert int(g["train/attn_nonbias_positions"]) == 543        # the chunk-index write quirk
