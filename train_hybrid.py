#!/usr/bin/env python
"""`train_hybrid.py` for the MI355X-native build: the reference CLI (same 35 flags, same defaults:
/root/reference/train_hybrid.py:1076-1133) driving the native VAE step (lunaris_orion_amd.trainer.VAEStepper).

What is the same: flags/defaults (the dead flags are accepted; `--save_every` / `--keep_n_checkpoints`, which the reference parses
and never reads, are honoured: periodic `checkpoints/step_<N>.pt`, the newest N kept), seeds
(:1138-1141), the dataset contract (`sprites*.npy` uint8 (N,128,128,3) + `labels*.csv`, :100-201), 90/10 split with
drop_last (:551-569), the step semantics of `_process_batch` (:838-954) including its accumulation rule, the 12 metric
names (:929-942), the checkpoint dictionary keys (:596-606), SIGINT -> checkpoint (:587-592).

What differs, deliberately: with the teacher on (the default flags) `HybridStepper` runs the full `_process_batch`
as the reference executes it, teacher dropout 0.1 included (`--teacher_dropout`, a builder flag, defaults to the reference's
constructor default; 0 selects the dropout-free fast path); `--feature_dim` 128 (default), 256 or 512 (README High-End recipe); with
`--reward_scale 0 --quality_weight 0` (or `--vae_only`) the teacher is not run at all (its result cannot influence the
VAE then) and the teacher-only metrics are reported as 0.  The reference's
defects are not inherited: no tensorboard hard dependency, no DataLoader timeout assertion, per-epoch average
loss is a real number, checkpoints are written.  Launch one process per GPU with torch.distributed.run for DP.
"""
from __future__ import annotations

import argparse
import logging
import os
import signal
import sys
import time
from pathlib import Path

# multi-process GPU work on this stack needs dmabuf IPC (hipIpcGetMemHandle fails in the legacy mode); harmless for one process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Hybrid Training for Lunaris: Generator and Evaluator (MI355X-native VAE path)")
    p.add_argument("--data_dir", type=str, required=True)
    p.add_argument("--output_dir", type=str, default="output")
    p.add_argument("--resume_from", type=str)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--gradient_accumulation_steps", type=int, default=2)
    p.add_argument("--chunk_size", type=int, default=32)                 # dead in the reference
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("--num_workers", type=int, default=4)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--compile", action="store_true")                     # no tracing compiler here: accepted, ignored
    p.add_argument("--mixed_precision", action="store_true")             # the native path always runs fp16 MFMA / fp32 accumulate
    p.add_argument("--latent_dim", type=int, default=256)
    p.add_argument("--embedding_dim", type=int, default=64)
    p.add_argument("--feature_dim", type=int, default=128)
    p.add_argument("--num_experts", type=int, default=4)
    p.add_argument("--vae_lr", type=float, default=1e-4)
    p.add_argument("--teacher_lr", type=float, default=1e-4)
    p.add_argument("--min_lr", type=float, default=1e-6)
    p.add_argument("--weight_decay", type=float, default=0.01)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--scheduler_t0", type=int, default=10)
    p.add_argument("--recon_weight", type=float, default=1.0)
    p.add_argument("--kl_weight", type=float, default=0.1)
    p.add_argument("--quality_weight", type=float, default=0.5)
    p.add_argument("--log_every", type=int, default=100)
    p.add_argument("--save_every", type=int, default=1000)               # dead in the reference; honoured here: checkpoints/step_<N>.pt every N steps
    p.add_argument("--sample_every", type=int, default=500)              # dead in the reference
    p.add_argument("--keep_n_checkpoints", type=int, default=5)          # dead in the reference; honoured here: the newest N step_<N>.pt files are kept
    p.add_argument("--early_stopping_patience", type=int, default=7)
    p.add_argument("--eval_save_freq", type=int, default=500)
    p.add_argument("--reward_scale", type=float, default=0.1)
    p.add_argument("--semantic_weight", type=float, default=0.5)
    p.add_argument("--baseline_momentum", type=float, default=0.9)
    p.add_argument("--force_cpu", action="store_true")
    p.add_argument("--memory_efficient", action="store_true")            # dead in the reference
    # additions of this build (defaults keep the reference behaviour)
    p.add_argument("--vae_only", action="store_true", help="shorthand for --reward_scale 0 --quality_weight 0")
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many micro-batches (0 = no limit)")
    p.add_argument("--generate_samples", type=int, default=0, help="decode this many prior samples to PNG when training ends (lunar_generate.py:278-291)")
    p.add_argument("--mfma_precision", choices=["fp16", "fp8"], default="fp16",
                   help="operand format of the VAE's forward convolutions: fp16 (parity-tested default) or fp8 = OCP e4m3 in the layers with Cin %% 128 == 0 (10 of 16 at batch 64; the 128->64 transposed conv stays on its patch-resident fp16 kernel), fp16 backward")
    p.add_argument("--teacher_dropout", type=float, default=0.1,
                   help="dropout_rate of the teacher (the reference constructs it with its default 0.1, train_hybrid.py:400-404); 0 = dropout-free fast path")
    p.add_argument("--teacher_full_backward", action="store_true",
                   help="train the teacher 'as documented' (SURVEY §8 F2): gradients and AdamW updates for the experts and the feature extractor too, "
                        "i.e. the reference with non-reentrant checkpoints (lunar_evaluator.py:194-197, 266-275, 411-414); off = the reference as it "
                        "executes (only the gate and the quality heads learn)")
    return p


from lunaris_orion_amd.data import SpriteFeeder, SpriteShards, epoch_batches, steps_per_epoch as _steps_per_epoch   # noqa: E402


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.vae_only:
        args.reward_scale, args.quality_weight = 0.0, 0.0
    if args.force_cpu:
        raise SystemExit("--force_cpu: this build has no CPU path (the CPU oracle under oracle/ is test infrastructure only)")
    teacher_on = args.reward_scale != 0.0 or args.quality_weight != 0.0
    if teacher_on and args.feature_dim not in (128, 256, 512):
        raise SystemExit(f"--feature_dim {args.feature_dim} is not built: 128 (default), 256 and 512 (README High-End recipe) are")
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)        # one rank per GPU on a full node; the modulo only matters in rehearsals
    torch.cuda.set_device(local)
    grad_sync = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl = RCCL over xGMI; LO_DIST_BACKEND=gloo rehearses the multi-rank control flow with several ranks on one GPU
        dist.init_process_group(os.environ.get("LO_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
        from lunaris_orion_amd.parallel import FlatGradSync
        # default: one RCCL all-reduce per hand-over range on the fp32 wire (DP == single process to fp32 rounding, the documented
        # contract).  LO_DP_EXCHANGE=direct opts into the xGMI-shaped exchange (all-to-all reduce-scatter + all-gather, fp16 wire):
        # it stays opt-in until an N > 1 RCCL run has checked its parity and timing on hardware
        dp_mode = os.environ.get("LO_DP_EXCHANGE", "allreduce")
        grad_sync = FlatGradSync(mode=dp_mode, compress_fp16=dp_mode == "direct")

    out_dir = Path(args.output_dir)
    (out_dir / "checkpoints").mkdir(parents=True, exist_ok=True)
    (out_dir / "eval_samples").mkdir(exist_ok=True)
    log = logging.getLogger("TrainHybrid")
    log.setLevel(logging.DEBUG)
    if rank == 0 and not log.handlers:
        fh = logging.FileHandler(out_dir / "training.log")
        fh.setLevel(logging.DEBUG)
        ch = logging.StreamHandler()
        ch.setLevel(logging.INFO)
        for h in (fh, ch):
            h.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
            log.addHandler(h)
    writer = None
    try:
        from torch.utils.tensorboard import SummaryWriter
        writer = SummaryWriter(log_dir=str(out_dir / "tensorboard")) if rank == 0 else None
    except Exception:
        log.info("tensorboard not installed: scalar logging goes to training.log only")

    from lunaris_orion_amd.trainer import HybridStepper, VAEStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE
    vae = LunarisCoreVAE(latent_dim=args.latent_dim, mfma_precision=args.mfma_precision).to("cuda")
    common = dict(lr=args.vae_lr, min_lr=args.min_lr, scheduler_t0=args.scheduler_t0, weight_decay=args.weight_decay,
                  max_grad_norm=args.max_grad_norm, recon_weight=args.recon_weight, kl_weight=args.kl_weight,
                  gradient_accumulation_steps=args.gradient_accumulation_steps, grad_sync=grad_sync,
                  pipeline_optimizer=True)     # Linear / decoder update beside the next encoder forward; checkpoints synchronise first
    teacher = None
    if teacher_on:
        from lunaris_orion_amd.teacher import LunarMoETeacher
        teacher = LunarMoETeacher(num_experts=args.num_experts, feature_dim=args.feature_dim, embedding_dim=args.embedding_dim,
                                  dropout_rate=args.teacher_dropout).to("cuda").train()
        stepper = HybridStepper(vae, teacher, teacher_lr=args.teacher_lr, quality_weight=args.quality_weight, reward_scale=args.reward_scale,
                                semantic_weight=args.semantic_weight, baseline_momentum=args.baseline_momentum,
                                teacher_full_backward=args.teacher_full_backward, **common)
        if args.teacher_full_backward:
            log.info("teacher full backward on: every teacher parameter on the loss path is trained (non-reentrant checkpoint semantics)")
        log.info(f"teacher on: LunarMoETeacher forward as executed by the reference, dropout_rate {args.teacher_dropout} in train mode"
                 + ("" if args.teacher_dropout > 0 else " (dropout-free fast path: constant-field shortcuts)"))
    else:
        stepper = VAEStepper(vae, **common)
    log.info(f"VAE Parameters - Total: {sum(p.numel() for p in vae.parameters()):,}")
    global_step, best_loss = 0, float("inf")

    from lunaris_orion_amd import hostside

    def save_checkpoint(tag="latest"):
        if rank != 0:
            return
        stepper.synchronize_parameters()          # the pipelined optimizer step may still be updating the decoder's parameters
        torch.cuda.synchronize()
        stepper.check_device_health()             # raises instead of writing a checkpoint after a lost launch (rendezvous-failure word)
        torch.save(hostside.checkpoint_dict(stepper, vae, teacher, global_step, best_loss, vars(args)), out_dir / "checkpoints" / f"{tag}.pt")
        log.info(f"Checkpoint saved at step {global_step}")

    if args.resume_from:
        ck = torch.load(args.resume_from, map_location="cpu", weights_only=True)
        global_step, best_loss = hostside.restore_checkpoint(ck, stepper, vae, teacher)
        log.info(f"Successfully loaded checkpoint from step {global_step}")

    data = SpriteShards(args.data_dir)
    n_train = int(0.9 * len(data))
    perm = torch.randperm(len(data)).numpy()                 # random_split(:555)
    train_idx = perm[:n_train]
    per_rank = args.batch_size
    steps_per_epoch = _steps_per_epoch(n_train, per_rank, world)    # drop_last (:569) at the global batch: identical on every rank
    log.info(f"Dataset initialized with {len(data)} samples; {steps_per_epoch} batches/epoch/rank")

    early = hostside.EarlyStopping(patience=args.early_stopping_patience)
    interrupted = {"flag": False}
    signal.signal(signal.SIGINT, lambda *_: interrupted.__setitem__("flag", True))
    done = False
    for epoch in range(args.num_epochs):
        t0 = time.time()
        epoch_losses = []
        feeder = SpriteFeeder(data, epoch_batches(train_idx, per_rank, rank, world), per_rank, device=f"cuda:{local}")
        for b, batch_u8 in enumerate(feeder):
            images = stepper.decode_sprites(batch_u8)
            stepper.step(images, batch_idx=b)
            global_step += 1
            if global_step % args.log_every == 0 or b == steps_per_epoch - 1:
                m = stepper.metrics()
                keys = ("recon_loss", "kl_loss", "quality_loss", "pg_loss", "semantic_reward", "quality_reward", "baseline", "advantage",
                        "vae_loss", "teacher_loss", "total_loss", "quality_scores")
                metrics = {k: m.get(k, m["vae_loss"] if k == "total_loss" else 0.0) for k in keys}
                epoch_losses.append(metrics["total_loss"])
                if writer is not None:
                    for k, v in metrics.items():
                        writer.add_scalar(k, v, global_step)
                log.info(f"step {global_step} loss {metrics['total_loss']:.4f} recon {metrics['recon_loss']:.4f} "
                         f"kl {metrics['kl_loss']:.4f} lr {m['lr']:.2e} grad_norm {m['grad_norm']:.3f}")
            if args.save_every > 0 and global_step % args.save_every == 0:                 # train_hybrid.py:1113,1115 (dead flags there)
                save_checkpoint(f"step_{global_step}")
                if rank == 0:
                    for gone in hostside.prune_periodic_checkpoints(out_dir / "checkpoints", args.keep_n_checkpoints):
                        log.debug(f"removed old periodic checkpoint {gone.name}")
            if rank == 0 and global_step % args.eval_save_freq == 0:                       # train_hybrid.py:951-952
                recon = stepper.last[0]
                tout = getattr(stepper, "last_teacher_out", None)
                pth = hostside.save_comparison(out_dir / "eval_samples" / f"comparison_{global_step}_{time.strftime('%Y%m%d_%H%M%S')}.png", images, recon,
                                               tout["quality_scores"] if tout else None, tout["semantic_score"] if tout else None)
                log.info(f"Saved comparison image {pth.name}")
            stop = interrupted["flag"]
            if grad_sync is not None:
                # data parallel: SIGINT may reach one rank only, or the ranks at different micro-batches; a rank that leaves the loop
                # alone strands the others in the next gradient exchange.  Agree on the flag (MAX over ranks) at a fixed point
                # every rank reaches: each log step
                stop = False
                if global_step % args.log_every == 0 or b == steps_per_epoch - 1:
                    t_stop = torch.tensor([1.0 if interrupted["flag"] else 0.0], dtype=torch.float32, device="cuda")
                    grad_sync.max_small(t_stop)
                    stop = bool(t_stop.item() > 0)
            if stop or (args.max_steps and global_step >= args.max_steps):
                interrupted["flag"] = interrupted["flag"] or stop
                done = True
                feeder.close()
                break
        avg = float(np.mean(epoch_losses)) if epoch_losses else float("nan")
        if grad_sync is not None:
            # data parallel: every rank must take the same early-stopping / best-checkpoint decision (a rank that stops alone
            # leaves the others waiting in the next gradient exchange): decide on the mean over ranks
            t_avg = torch.tensor([avg], dtype=torch.float32, device="cuda")
            grad_sync.average_small(t_avg)
            avg = float(t_avg.item())
        log.info(f"Epoch {epoch + 1} Summary: Time {(time.time() - t0) / 60:.2f} min, Average Loss {avg:.4f}, Best Loss {best_loss:.4f}")
        early(avg)                                                                         # train_hybrid.py:1049-1052
        if early.early_stop:
            log.info("Early stopping triggered")
            done = True
        if avg < best_loss:
            best_loss = avg
            save_checkpoint("best")
        save_checkpoint("latest")
        if done:
            break
    if rank == 0 and args.generate_samples > 0:
        paths = hostside.save_samples(out_dir, vae.sample(args.generate_samples), global_step)
        log.info(f"Generated and saved {len(list(paths))} samples")
    if writer is not None:
        writer.close()
    log.info("Training completed.")


if __name__ == "__main__":
    main()
