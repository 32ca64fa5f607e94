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
        LunarMoETeacher(feature_dim=512)


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
        got = out[k].cpu()
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
