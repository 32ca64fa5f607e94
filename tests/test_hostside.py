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
