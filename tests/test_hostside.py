"""Host-side pieces around the step: early stopping, PNG outputs, torch-format optimizer state."""
from collections import OrderedDict

import numpy as np
import torch

from lunaris_orion_amd import hostside
from lunaris_orion_amd.trainer import cosine_warm_restarts_lr


def test_early_stopping_matches_reference_semantics():
    """train_hybrid.py:206-225: the counter resets on any non-worse loss and trips after `patience` worse epochs."""
    es = hostside.EarlyStopping(patience=2)
    for v, stop in [(1.0, False), (1.1, False), (0.9, False), (0.95, False), (0.96, True)]:
        es(v)
        assert es.early_stop == stop, v
    assert es.best_loss == 0.9


def test_adamw_and_scheduler_state_dicts_are_torch_compatible():
    ps = OrderedDict(a=torch.nn.Parameter(torch.randn(3, 4)), b=torch.nn.Parameter(torch.randn(5)), c=torch.nn.Parameter(torch.randn(2)))
    opt = torch.optim.AdamW(ps.values(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=1e-6)
    g = torch.Generator().manual_seed(0)
    for _ in range(13):
        for k, p in ps.items():
            p.grad = None if k == "c" else torch.randn(p.shape, generator=g)       # "c" never receives a gradient
        opt.step(); sch.step()
    ref = opt.state_dict()
    off = {"a": 0, "b": 64, "c": 128}
    m, v = torch.zeros(192), torch.zeros(192)
    assert hostside.load_adamw_state_dict(ref, ps, m, v, off) == 13
    mine = hostside.adamw_state_dict(ps, m, v, off, 13, ref["param_groups"][0]["lr"], 1e-4, (0.9, 0.999), 1e-8, 0.01, only=["a", "b"])
    assert set(mine["param_groups"][0]) == set(ref["param_groups"][0]) and set(mine["state"]) == set(ref["state"]) == {0, 1}
    for i in ref["state"]:
        for k in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(mine["state"][i][k], ref["state"][i][k]), (i, k)
    opt2 = torch.optim.AdamW(ps.values(), lr=1e-4, weight_decay=0.01)
    sch2 = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt2, T_0=10, T_mult=2, eta_min=1e-6)
    opt2.load_state_dict(mine)
    sch2.load_state_dict(hostside.scheduler_state_dict(10, 2, 1e-6, 1e-4, 13, sch.get_last_lr()[0]))
    sch.step(); sch2.step()
    assert abs(sch.get_last_lr()[0] - sch2.get_last_lr()[0]) < 1e-15
    assert abs(sch.get_last_lr()[0] - cosine_warm_restarts_lr(1e-4, 1e-6, 10, 2, 14)) < 1e-12


def test_png_outputs(tmp_path):
    x = torch.rand(5, 3, 128, 128) * 2 - 1
    u8 = hostside.to_uint8_hwc(x)
    assert u8.shape == (5, 128, 128, 3) and u8.dtype == np.uint8
    assert np.array_equal(u8, np.transpose(((x + 1) * 127.5).clamp(0, 255).numpy().astype(np.uint8), (0, 2, 3, 1)))
    p = hostside.save_comparison(tmp_path / "eval" / "c.png", x, -x, torch.rand(5, 4), torch.rand(5, 1))
    from PIL import Image
    im = Image.open(p)
    assert im.size == (2 * 128 + 10, 4 * 128 + 3 * 10 + 30)                      # train_hybrid.py:736-738
    got = np.asarray(im)[:128, :128]
    assert np.array_equal(got, u8[0])
    paths = list(hostside.save_samples(tmp_path / "s", x[:2], 7))
    assert len(paths) == 2 and all(q.exists() and q.name.startswith("sample_7_") for q in paths)


def test_loss_scale_policy_counts_one_overflow_once_under_a_two_step_lag():
    """observations lag the device by one or two optimizer steps.  One genuine overflow at step n also skips the steps
    already enqueued at the old scale; the policy halves once for it (GradScaler.update() semantics), halves again only for a
    skip that ran at the new scale, and regrows after `growth_interval` clean steps."""
    from lunaris_orion_amd.trainer import LossScalePolicy
    pol, scale, lag = LossScalePolicy(init=65536.0, growth_interval=50), 65536.0, 2
    overflow_above = 40000.0                     # a step skips iff it was ENQUEUED at a scale above this
    scale_at, skipped, history = {}, {}, []
    total = 0
    for k in range(1, 131):                      # step k is enqueued at the current scale, then the observation of step k - lag lands
        scale_at[k] = scale
        total += 1 if (scale_at[k] > overflow_above and k >= 10) else 0
        skipped[k] = total
        if k - lag >= 1:
            new_scale = pol.observe(scale, skipped[k - lag], k - lag, k)
            if new_scale != scale:
                history.append((k, new_scale))
            scale = new_scale
    assert history[0] == (12, 32768.0)           # the overflow of step 10 is seen two steps later: ONE halving ...
    assert [h for h in history if h[0] < 60] == [(12, 32768.0)]      # ... although steps 11 and 12 skipped too (old scale)
    assert history[1] == (62, 65536.0)           # 50 clean steps later the scale grows back,
    assert history[2][1] == 32768.0 and history[2][0] == 65                # overflows again at the first step enqueued there, halves once
    # a skip at the NEW scale is a new overflow: halved again
    pol2, s2 = LossScalePolicy(), 65536.0
    s2 = pol2.observe(s2, 1, 5, 7)               # step 5 skipped, seen at 7 -> 32768; steps 6, 7 ran at the old scale
    assert s2 == 32768.0
    s2 = pol2.observe(s2, 3, 7, 9)               # steps 6, 7 skipped (old scale): ignored
    assert s2 == 32768.0
    s2 = pol2.observe(s2, 4, 8, 10)              # step 8 ran at 32768 and skipped: new overflow
    assert s2 == 16384.0


def test_periodic_checkpoint_retention(tmp_path):
    """`--save_every` / `--keep_n_checkpoints` (train_hybrid.py:1113-1115): the newest N step_<N>.pt files stay, latest / best do too."""
    for n in (100, 200, 1000, 300, 50):
        (tmp_path / f"step_{n}.pt").write_bytes(b"x")
    for n in ("latest", "best", "step_final"):
        (tmp_path / f"{n}.pt").write_bytes(b"x")
    removed = hostside.prune_periodic_checkpoints(tmp_path, 3)
    assert sorted(p.name for p in removed) == ["step_100.pt", "step_50.pt"]
    assert sorted(p.name for p in tmp_path.iterdir()) == ["best.pt", "latest.pt", "step_1000.pt", "step_200.pt", "step_300.pt", "step_final.pt"]
    assert hostside.prune_periodic_checkpoints(tmp_path, 5) == []


def test_default_teacher_for_vae_only_checkpoints_leaves_the_rng_alone_and_fails_softly():
    """building the default teacher entries of a VAE-only checkpoint must not reseed the run's generators, and a
    `--feature_dim` the teacher does not implement must not kill the run at its first save."""
    import pytest
    torch.manual_seed(1234)
    a = torch.rand(3)
    torch.manual_seed(1234)
    hostside._DEFAULT_TEACHER.clear()
    hostside._default_teacher({"seed": 7, "feature_dim": 128})
    assert torch.equal(torch.rand(3), a)
    with pytest.raises(NotImplementedError):
        hostside._default_teacher({"feature_dim": 64})

    class _V(torch.nn.Module):                    # the smallest object checkpoint_dict() needs: parameters that are views of one buffer
        def __init__(self):
            super().__init__()
            self._flat = torch.zeros(64)
            self.w = torch.nn.Parameter(torch.zeros(4, 4))
            self.w.data = self._flat[:16].view(4, 4)
            self.loss_scale, self.noise_calls = 65536.0, 17

        def flat_parameters(self):
            return self._flat

    class _S:
        exp_avg, exp_avg_sq = torch.zeros(64), torch.zeros(64)
        opt_steps, lr, base_lr, betas, eps, weight_decay, t0, min_lr = 0, 1e-4, 1e-4, (0.9, 0.999), 1e-8, 0.01, 10, 1e-6
    with pytest.warns(UserWarning, match="without teacher entries"):
        ck = hostside.checkpoint_dict(_S(), _V(), None, 5, 1.0, {"feature_dim": 64, "teacher_lr": 1e-4})
    assert ck["teacher_state_dict"] == {} and ck["lunaris_amd_extra"]["noise_calls"] == 17
    assert "teacher_entries_omitted" in ck["lunaris_amd_extra"] and set(ck["vae_state_dict"]) == {"w"}


def test_stream_position_fast_forward():
    from lunaris_orion_amd.vae import lcg_advance
    s = x = 0xDEADBEEFCAFEF00D
    for _ in range(777):
        x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    assert lcg_advance(s, 777) == x and lcg_advance(s, 0) == s
