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
    direct.direct_min_elems = 0              # the product sends ranges below 4 M elements through one all-reduce (latency-bound)
    direct.begin(flat_direct)
    direct.finish()
    assert direct.bytes_per_phase() == [4 * flat_direct.numel()] and direct.modes_per_phase() == ["direct"]
    small = FlatGradSync(mode="direct")      # ... which is what this one does: same numbers, reported as what it ran as
    fs = flat.clone()
    fs.copy_(torch.cat([g.flatten() for g in R.OracleTrainer(P).step(xs, es, 0.0, do_update=False)["grads"].values()]))
    small.begin(fs[:1000]); small.finish()
    assert small.modes_per_phase() == ["allreduce"] and torch.equal(fs[:1000], flat[:1000])
    # the product's hand-over pattern: three ranges handed over one after the other (tail of the buffer first, like the phased
    # backward does), fp16 wire, then one finish(); every range complete and equal to the all-reduce average to fp16 rounding
    o2 = R.OracleTrainer(P).step(xs, es, 0.0, do_update=False)
    flat16 = torch.cat([g.flatten() for g in o2["grads"].values()])
    n = flat16.numel()
    b1, b2 = (n // 10) * 3 + 1, (n // 10) * 1 + 3                # odd boundaries: remainder paths in every range
    d16 = FlatGradSync(mode="direct", compress_fp16=True)
    d16.compress_min_elems = d16.direct_min_elems = 0   # the fp16 wire / direct form on every range (the product keeps ranges below 16 MB on the fp32 wire and on one all-reduce)
    d16.begin(flat16[b1:]); d16.begin(flat16[b2:b1]); d16.begin(flat16[:b2])
    d16.finish()
    assert d16.bytes_per_phase() == [2 * (n - b1), 2 * (b1 - b2), 2 * b2]
    # the product's rule: a range below `compress_min_elems` stays on the fp32 wire (the last, exposed range of a step)
    mixed = FlatGradSync(mode="direct", compress_fp16=True)
    mixed.direct_min_elems = 0
    mixed.compress_min_elems = b1 - b2 + 1
    fm = torch.cat([g.flatten() for g in R.OracleTrainer(P).step(xs, es, 0.0, do_update=False)["grads"].values()])
    mixed.begin(fm[b1:]); mixed.begin(fm[b2:b1]); mixed.begin(fm[:b2])
    mixed.finish()
    assert mixed.bytes_per_phase() == [2 * (n - b1), 4 * (b1 - b2), 4 * b2]
    assert (fm[:b1] - flat[:b1]).abs().max().item() <= 1e-6 * flat.abs().max().item()      # fp32 wire: the all-reduce average
    # the Linear layers' gradients as factors (FlatGradSync.begin_factored): every rank all-gathers the factor blocks and forms
    # dW = (1 / N) sum_r Y_r^T X_r from the gathered buffer itself -- the all-reduce average of the per-rank matrices to fp32 rounding
    gen = torch.Generator().manual_seed(100 + rank)
    Nf, Kf, Bp = 48, 80, 32
    yt = (torch.randn(Nf, Bp, generator=gen) * 0.5).half()           # dY^T [N][Bp] of this rank
    xt = (torch.randn(Kf, Bp, generator=gen) * 0.5).half()           # X^T  [K][Bp]
    block = torch.cat([yt.flatten(), xt.flatten()])
    dw_local = yt.float() @ xt.float().t()                           # this rank's materialised gradient
    dw_allreduce = dw_local.clone()
    FlatGradSync()(dw_allreduce.view(-1))                            # the exchange the factors replace
    dw_fact = torch.empty(Nf, Kf)

    def materialize(gathered, w):
        acc = torch.zeros(Nf, Kf)
        per = block.numel()
        for r_ in range(w):                                          # rank-major blocks, rank order: the same bits on every rank
            blk = gathered[r_ * per:(r_ + 1) * per]
            acc += blk[:Nf * Bp].view(Nf, Bp).float() @ blk[Nf * Bp:].view(Kf, Bp).float().t()
        dw_fact.copy_(acc / w)
    fsync = FlatGradSync()
    piece = torch.full((7,), float(rank + 1))
    fsync.begin_factored([piece], block, materialize)
    fsync.finish()
    assert fsync.modes_per_phase()[0] == "factors" and fsync.bytes_per_phase() == [2 * block.numel() + 4 * 7]
    assert torch.equal(piece, torch.full((7,), 1.5))                 # the rest of the range: averaged the usual way
    # plain Python objects only (tensors through a spawn-context Queue need the producer to stay alive)
    res = {"sum": flat.double().sum().item(), "l2": flat.double().norm().item(),
           "direct_vs_allreduce": (flat_direct - flat[:-1]).abs().max().item(),
           "direct16_rel": ((flat16 - flat).norm() / flat.norm()).item(), "direct16_sum": flat16.double().sum().item(),
           "lr": [cosine_warm_restarts_lr(1e-4, 1e-6, 10, 2, k) for k in range(40)],
           "factored_vs_allreduce": ((dw_fact - dw_allreduce).abs().max() / dw_allreduce.abs().max()).item(),
           "factored_sum": dw_fact.double().sum().item()}
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
    # three hand-over ranges on the fp16 wire: identical on both ranks, equal to the fp32 average to fp16 rounding (2^-11 relative
    # per element before the fp32 share sum)
    assert out[0]["direct16_sum"] == out[1]["direct16_sum"]
    assert max(out[0]["direct16_rel"], out[1]["direct16_rel"]) <= 1e-3, (out[0]["direct16_rel"], out[1]["direct16_rel"])
    # factor exchange == all-reduce of the materialised gradient (fp32 rounding), identical on both ranks
    assert max(out[0]["factored_vs_allreduce"], out[1]["factored_vs_allreduce"]) <= 1e-6
    assert out[0]["factored_sum"] == out[1]["factored_sum"]


def test_shard_batch_partitions_without_overlap():
    sys.path.insert(0, ROOT)
    from lunaris_orion_amd.parallel import shard_batch
    x = torch.arange(10).view(10, 1)
    parts = [shard_batch(x, r, 4) for r in range(4)]
    assert [p.shape[0] for p in parts] == [2, 2, 2, 2]
    assert torch.cat(parts).flatten().tolist() == list(range(8))
