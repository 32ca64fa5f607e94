"""World-size-2 gloo tests (CPU) of the data-parallel path: averaging the flat per-rank gradients with
FlatGradSync reproduces the single-process gradient at the global batch (checked with the oracle), and the
host-side step logic (LR schedule, accumulation rule) agrees between ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vae_ref as R
    from lunaris_orion_amd.parallel import FlatGradSync, shard_batch
    from lunaris_orion_amd.trainer import cosine_warm_restarts_lr
    L, Bg = 64, 2 * world
    P = R.closed_form_params(L)
    x = R.normalise_sprites(R.closed_form_sprites(Bg))
    eps = R.closed_form_eps(Bg, L)
    # this rank's shard
    xs, es = shard_batch(x, rank, world), shard_batch(eps, rank, world)
    o = R.OracleTrainer(P).step(xs, es, 0.0, do_update=False)
    flat = torch.cat([g.flatten() for g in o["grads"].values()])
    flat_direct = flat.clone()[:-1]          # odd length: the direct form's remainder path
    sync = FlatGradSync()
    sync(flat)
    direct = FlatGradSync(mode="direct")
    direct.begin(flat_direct)
    direct.finish()
    assert direct.bytes_per_phase() == [4 * flat_direct.numel()]
    # plain Python objects only (tensors through a spawn-context Queue need the producer to stay alive)
    res = {"sum": flat.double().sum().item(), "l2": flat.double().norm().item(),
           "direct_vs_allreduce": (flat_direct - flat[:-1]).abs().max().item(),
           "lr": [cosine_warm_restarts_lr(1e-4, 1e-6, 10, 2, k) for k in range(40)]}
    if rank == 0:
        full = R.OracleTrainer(P).step(x, eps, 0.0, do_update=False)
        fg = torch.cat([g.flatten() for g in full["grads"].values()])
        res["rel"] = ((flat - fg).norm() / fg.norm()).item()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_sync_equals_global_batch_gradient():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out[0]["sum"] == out[1]["sum"] and out[0]["l2"] == out[1]["l2"]   # both ranks hold the same averaged gradient
    assert out[0]["rel"] <= 1e-5, out[0]["rel"]               # == single-process gradient at the global batch
    assert out[0]["lr"] == out[1]["lr"]
    # the all-to-all reduce-scatter + all-gather form gives the same average (sum order differs: fp32 rounding only)
    assert max(out[0]["direct_vs_allreduce"], out[1]["direct_vs_allreduce"]) <= 1e-7


def test_shard_batch_partitions_without_overlap():
    sys.path.insert(0, ROOT)
    from lunaris_orion_amd.parallel import shard_batch
    x = torch.arange(10).view(10, 1)
    parts = [shard_batch(x, r, 4) for r in range(4)]
    assert [p.shape[0] for p in parts] == [2, 2, 2, 2]
    assert torch.cat(parts).flatten().tolist() == list(range(8))
