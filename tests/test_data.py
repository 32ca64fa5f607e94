"""Sprite input path: shard index, epoch batching, the asynchronous feeder."""
import os

import numpy as np
import pytest
import torch

from lunaris_orion_amd.data import SpriteShards, epoch_batches, split_indices


def _write(tmp, sizes, seed=0):
    rng = np.random.default_rng(seed)
    allv = []
    for i, n in enumerate(sizes):
        a = rng.integers(0, 256, (n, 128, 128, 3), dtype=np.uint8)
        np.save(os.path.join(tmp, f"sprites_{i:03d}.npy"), a)
        with open(os.path.join(tmp, f"labels_{i:03d}.csv"), "w") as f:
            f.write("prompt,style,palette,theme,category,size,seed,quality\n")
            for _ in range(n):
                f.write("a,b,c,d,e,128,0,1\n")
        allv.append(a)
    return np.concatenate(allv)


def test_shards_global_index_and_checks(tmp_path):
    allv = _write(str(tmp_path), [5, 3, 4])
    s = SpriteShards(str(tmp_path))
    assert len(s) == 12
    idx = np.array([11, 0, 5, 4, 7, 8, 3])
    got = s.batch_u8(idx).numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, allv[idx])
    with open(os.path.join(str(tmp_path), "labels_002.csv"), "a") as f:
        f.write("a,b,c,d,e,128,0,1\n")
    with pytest.raises(AssertionError, match="Mismatch between total sprites"):
        SpriteShards(str(tmp_path))
    with pytest.raises(ValueError, match="No sprites or labels files found"):
        SpriteShards(os.path.join(str(tmp_path), "nope"))


def test_wrong_sprite_shape_is_refused(tmp_path):
    np.save(os.path.join(str(tmp_path), "sprites_000.npy"), np.zeros((2, 64, 64, 3), dtype=np.uint8))
    open(os.path.join(str(tmp_path), "labels_000.csv"), "w").write("h\n1\n2\n")
    with pytest.raises(ValueError, match="Expected 128x128x3"):
        SpriteShards(str(tmp_path))


def test_split_and_epoch_batches_follow_the_reference_loader():
    tr, va = split_indices(103, 0.9, torch.Generator().manual_seed(1))
    assert len(tr) == 92 and len(va) == 11 and len(set(tr) | set(va)) == 103          # int(0.9 n), the rest
    seen = []
    for r in range(2):
        bs = list(epoch_batches(tr, 8, rank=r, world=2, rng=np.random.default_rng(3)))
        assert len(bs) == (92 // 2) // 8 and all(len(b) == 8 for b in bs)              # drop_last per rank
        seen += [i for b in bs for i in b]
    assert len(seen) == len(set(seen)) and set(seen) <= set(tr)                         # ranks see disjoint samples
    # sizes that are NOT a multiple of world * batch: every rank must still get the same number of batches (a rank with one
    # batch more would run one more gradient exchange than the others and hang the job)
    from lunaris_orion_amd.data import steps_per_epoch
    for n, batch, world in ((63, 16, 2), (127, 16, 2), (2 * 16 * 3 + 31, 16, 2), (1000, 7, 8), (8 * 7 - 1, 7, 8)):
        counts = [len(list(epoch_batches(np.arange(n), batch, rank=r, world=world, rng=np.random.default_rng(5)))) for r in range(world)]
        assert counts == [steps_per_epoch(n, batch, world)] * world, (n, batch, world, counts)
    fixed = list(epoch_batches(np.arange(20), 6, shuffle=False))
    assert [list(b) for b in fixed] == [list(range(0, 6)), list(range(6, 12)), list(range(12, 18))]


@pytest.mark.gpu
def test_feeder_batches_equal_the_dataset_arithmetic(tmp_path):
    """uint8 batches arrive in order through pinned buffers + the copy stream; decoded on the GPU they equal
    `x/127.5 - 1`, HWC->CHW (train_hybrid.py:181-182) to 1 ulp.  More batches than ring slots: slots are reused."""
    from lunaris_orion_amd.data import SpriteFeeder
    from lunaris_orion_amd.trainer import VAEStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    allv = _write(str(tmp_path), [9, 14])
    s = SpriteShards(str(tmp_path))
    batches = [np.sort(np.random.default_rng(k).choice(23, 4, replace=False)) for k in range(11)]
    st = VAEStepper(LunarisCoreVAE(256).to("cuda"))
    n = 0
    for k, u8 in enumerate(SpriteFeeder(s, batches, 4, depth=3)):
        x = st.decode_sprites(u8)
        ref = torch.from_numpy(allv[batches[k]]).float().div(127.5).sub(1.0).permute(0, 3, 1, 2)
        assert (x.cpu() - ref).abs().max().item() <= 1.2e-7, k
        n += 1
    assert n == len(batches)
    with pytest.raises(ValueError, match="feeder built for"):
        list(SpriteFeeder(s, [np.arange(3)], 4))
