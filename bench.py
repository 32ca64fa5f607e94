#!/usr/bin/env python
"""Benchmark of the hot path: 128x128 sprites/sec through the full VAE training step on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A step = TrainingManager._process_batch semantics for the VAE (reference train_hybrid.py:838-954) with
--gradient_accumulation_steps 1 and the teacher's scalar at 0 (--reward_scale 0 --quality_weight 0): forward,
MSE + KL, backward, global-norm clip, AdamW, LR schedule — all through liblunaris_hip.so.  Inputs are synthetic
sprites already resident in HBM (fp32 NCHW, normalised like train_hybrid.py:181-182).  Per-GPU batch is fixed
(weak scaling); with N>1 the flat gradient buffer is averaged with one RCCL all-reduce per step.

Prints ONE JSON line (rank 0) with the throughput, a `roofline` object for the dominant kernel (HIP events on the
launch stream, collected by the library's per-launch profiler in extra steps after the timed region) and a
`cpu_baseline` object (the CPU oracle timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# multi-process GPU work on this stack needs dmabuf IPC (hipIpcGetMemHandle fails in the legacy mode); harmless for one process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0           # HBM3E spec (same table)


def synth_sprites(n: int, seed: int) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    u8 = rng.integers(0, 256, (n, 128, 128, 3), dtype=np.uint8)
    return (torch.from_numpy(u8).float() / 127.5 - 1.0).permute(0, 3, 1, 2).contiguous()


def collect_profile(lib):
    rows = {}
    name = C.create_string_buffer(128)
    ms, fl, by = C.c_double(), C.c_double(), C.c_double()
    for i in range(lib.lo_prof_count()):
        lib.lo_prof_get(i, name, 128, C.byref(ms), C.byref(fl), C.byref(by))
        r = rows.setdefault(name.value.decode(), [0.0, 0, 0.0, 0.0])
        if ms.value >= 0:
            r[0] += ms.value
            r[1] += 1
            r[2] += fl.value
            r[3] += by.value
    return rows


def _host_cores() -> int:
    # the GPU box gives one GPU's share of the host (16 cores); more threads than that only oversubscribes
    return min(16, len(os.sched_getaffinity(0)))


def _oracle_trainer(latent: int):
    from oracle import vae_ref as R
    torch.manual_seed(0)
    P = {k: torch.randn(shp) * (0.02 if len(shp) > 1 else 0.0) + (1.0 if k.endswith(".1.weight") else 0.0)
         for k, shp in R.param_shapes(latent).items()}
    return R.OracleTrainer(P)


def cpu_baseline_hybrid(latent: int, batch: int = 4, warm: int = 1, steps: int = 2, dropout: float = 0.1):
    """Full hybrid steps of the CPU oracle (VAE step + two train-mode teacher forwards with dropout masks), bounded sample:
    `warm` untimed + `steps` timed steps, median."""
    from oracle import dropout_ref as D
    from oracle import teacher_ref as T
    cores = _host_cores()
    torch.set_num_threads(cores)
    tr = _oracle_trainer(latent)
    S = T.closed_form_teacher_state(embedding_dim=256)
    x = synth_sprites(batch, 321)
    eps = torch.randn(batch, latent)
    ts = []
    for i in range(warm + steps):
        t0 = time.perf_counter()
        with torch.no_grad():
            T.teacher_forward(x, S, training=True, masks=D.TeacherMasks(2 * i + 1, dropout, batch) if dropout > 0 else None)
        recon = tr.step(x, eps, 0.0, True)["recon"]
        with torch.no_grad():
            T.teacher_forward(recon, S, training=True, masks=D.TeacherMasks(2 * i + 2, dropout, batch) if dropout > 0 else None)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts[warm:])[len(ts[warm:]) // 2]
    return {"value": batch / med, "unit": "sprites/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle, full hybrid step (VAE step + 2 train-mode teacher forwards, dropout {dropout}, fp32, {cores} threads), "
                      f"batch {batch} latent {latent}, {warm} warm-up + {steps} timed steps, median"}


def cpu_baseline(batch: int, latent: int, steps: int, warm: int = 3):
    """The oracle (CPU restatement, fp32, the box's host cores) on a bounded sample of the same workload (BASELINE.md §4:
    3 warm-up + >= 10 timed steps, median)."""
    cores = _host_cores()
    torch.set_num_threads(cores)
    tr = _oracle_trainer(latent)
    x = synth_sprites(batch, 123)
    eps = torch.randn(batch, latent)
    for _ in range(warm):
        tr.step(x, eps, 0.0)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        tr.step(x, eps, 0.0)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    return {"value": batch / med, "unit": "sprites/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (oracle/vae_ref.py OracleTrainer, fp32, {cores} threads), batch {batch} latent {latent}, "
                      f"{warm} warm-up + {steps} timed full training steps, median"}


MIN_WARMUP_STEPS = 300


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--latent", type=int, default=512)
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--cpu-batch", type=int, default=0, help="batch of the main CPU-baseline leg (0 = the GPU leg's batch)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed steps of each CPU-baseline leg (after --cpu-warmup warm-up steps; median)")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="warm-up steps of the VAE-only CPU-baseline legs")
    ap.add_argument("--cpu-hybrid-steps", type=int, default=2, help="timed steps of the CPU baseline of the full hybrid step (batch 4; 0 = skip that leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-warmup", type=int, default=MIN_WARMUP_STEPS, help="warm-up steps are topped up to this count (tests pass 0)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank control flow on one GPU)")
    ap.add_argument("--dp-fp16", action="store_true", help="fp16 wire format for the gradient exchange (the default with --dp-exchange direct)")
    ap.add_argument("--dp-fp32", action="store_true", help="fp32 wire format (the default with --dp-exchange allreduce)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--precision", choices=["fp16", "fp8"], default="fp16",
                    help="operand format of the forward convs in the timed leg (fp8 = BASELINE config 5 mode; the headline number is fp16)")
    ap.add_argument("--fp8-steps", type=int, default=100, help="extra leg (N=1): VAE-only steps in the fp8 operand mode, BASELINE config 5 (0 = skip)")
    ap.add_argument("--hybrid-steps", type=int, default=20, help="extra leg (N=1): full hybrid VAE+teacher steps, BASELINE config 3 (0 = skip)")
    ap.add_argument("--fullbwd-steps", type=int, default=3, help="extra leg (N=1): full hybrid steps with --teacher_full_backward, SURVEY F2 (0 = skip)")
    ap.add_argument("--highend-steps", type=int, default=3, help="extra leg (N=1): full hybrid steps of the README High-End recipe, --feature_dim 512 (0 = skip)")
    ap.add_argument("--config2-steps", type=int, default=100, help="extra leg (N=1): VAE-only steps at batch 32 / latent 256, BASELINE config 2 (0 = skip)")
    ap.add_argument("--dp-exchange", choices=["allreduce", "direct"], default=os.environ.get("LO_DP_EXCHANGE", "direct"),
                    help="N > 1: gradient exchange = the direct all-to-all reduce-scatter + all-gather over all xGMI links (default; fp16 wire) "
                         "or one RCCL all-reduce per hand-over range (fp32 wire); every range is exchanged on a communication stream behind the backward")
    args = ap.parse_args()
    args.dp_fp16 = (args.dp_fp16 or args.dp_exchange == "direct") and not args.dp_fp32

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    local_dev = local_rank % max(torch.cuda.device_count(), 1)     # one rank per GPU on a full node; the modulo only matters in rehearsals
    torch.cuda.set_device(local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from lunaris_orion_amd import _lib
    from lunaris_orion_amd.trainer import VAEStepper
    from lunaris_orion_amd.vae import LunarisCoreVAE

    torch.manual_seed(42)                      # identical initial weights on every rank (train_hybrid.py:1088,1138)
    model = LunarisCoreVAE(latent_dim=args.latent, mfma_precision=args.precision).to("cuda")
    grad_sync = None
    if world > 1:
        from lunaris_orion_amd.parallel import FlatGradSync
        grad_sync = FlatGradSync(compress_fp16=args.dp_fp16, mode=args.dp_exchange, time_exposed=True)
    pipeline = os.environ.get("LO_PIPELINE_OPT", "1") != "0"      # A/B knob; the pipelined optimizer step is the default
    st = VAEStepper(model, lr=1e-4, min_lr=1e-6, scheduler_t0=10, weight_decay=0.01, max_grad_norm=1.0, recon_weight=1.0,
                    kl_weight=0.1, gradient_accumulation_steps=1, grad_sync=grad_sync, pipeline_optimizer=pipeline)
    B = args.batch
    pool = [synth_sprites(B, 1000 * rank + i).cuda() for i in range(4)]

    def run(n):
        t = time.perf_counter()
        for i in range(n):
            st.step(pool[i % len(pool)], batch_idx=i)
        return time.perf_counter() - t        # host time to ENQUEUE n steps (the queue is drained by the caller)

    # W untimed warm-up steps as asked, topped up to MIN_WARMUP_STEPS (same fixed count on every rank): the first ~0.5 s after start-up
    # (clock ramp, first-touch of the 1.5 GB workspace, RCCL channel set-up) run 15-20 % slower and made 50-step timings scatter
    # between 15 k and 19.5 k sprites/s on the same box; with them out of the way the same 50 steps repeat within 0.5 %
    warmup_run = max(args.warmup, args.min_warmup)
    run(warmup_run)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if grad_sync is not None:
        grad_sync.reset_timing()
    t0 = time.perf_counter()
    host_enqueue_s = run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    met = st.metrics()
    assert met["grads_finite"] == 1.0 and np.isfinite(met["recon_loss"]), met
    # the host's own cost of enqueuing a step: a burst of 5 steps into an EMPTY queue, outside the timed region (host_enqueue_ms
    # above is taken with ~20 steps already queued, where the runtime's back-pressure, not the host, sets the pace)
    torch.cuda.synchronize()
    host_idle_s = run(5) / 5
    torch.cuda.synchronize()
    dp_info = None
    if dist is not None:
        # what actually ran: ranks of the process group and the device each one drove, bytes handed to the exchange per backward
        # phase, and the time the optimizer's stream had to wait for the exchange (HIP events around FlatGradSync.finish())
        ids = [None] * world
        dist.all_gather_object(ids, {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(),
                                     "exposed_ms": grad_sync.exposed_ms_per_step()})
        dp_info = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "devices": [d["device"] for d in ids],
                   "device_names": sorted(set(d["name"] for d in ids)), "exchange": args.dp_exchange, "exchange_per_phase": grad_sync.modes_per_phase(),
                   "wire": "fp16" if args.dp_fp16 else "fp32", "bytes_per_phase": grad_sync.bytes_per_phase(),
                   "exchange_exposed_ms": max((d["exposed_ms"] for d in ids if d["exposed_ms"] is not None), default=None)}

    out = None
    if rank == 0:
        # ---- roofline leg: per-launch HIP events on the launch stream, extra steps after the timed region
        rows = {}
        if args.prof_steps > 0:
            saved_sync, st.grad_sync = st.grad_sync, None     # rank-local leg: the other ranks are not stepping, so no collective
            _lib.lib.lo_prof_enable(1)
            run(args.prof_steps)
            torch.cuda.synchronize()
            rows = collect_profile(_lib.lib)
            _lib.lib.lo_prof_enable(0)
            st.grad_sync = saved_sync
        if not rows:
            rows = {"(not profiled)": [1.0, 1, 0.0, 0.0]}
        total_ms = sum(r[0] for r in rows.values())
        # the tile instantiations of one template are one kernel: lo_igemm_nt<128,64,64>, <64,64,64>, ... -> lo_igemm_nt
        fam = {}
        for k, r in rows.items():
            f = fam.setdefault(k.split("<")[0].split(" ")[0], [0.0, 0, 0.0, 0.0, 0])
            for i in range(4):
                f[i] += r[i]
            f[4] += 1
        dom_name, dom = max(fam.items(), key=lambda kv: kv[1][0])
        n_inst = dom[4]
        dom = dom[:4]
        d_ms, d_n, d_fl, d_by = dom
        if d_fl > 0:
            roof = {"bound": "mfma", "achieved": d_fl / (d_ms * 1e-3) / 1e12, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": d_by / (d_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        # HBM bytes per launch from the PMC passes committed under profiles/ (tools/rocpd_extract.py traffic); null when absent
        roof["traffic"] = None
        try:
            tpath = next(pp for pp in (os.path.join(ROOT, "profiles", f) for f in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")) if os.path.exists(pp))
            tr = json.load(open(tpath))["kernels"]
            roof["traffic_source"] = f"profiles/{os.path.basename(tpath)} (committed rocprofv3 PMC passes of this workload; not re-measured in this run)"
            key = next((k for k in tr if k in dom_name or dom_name.split("<")[0] in k), None)
            if key:
                roof["traffic"] = tr[key]["hbm_bytes_per_launch"]
                roof["traffic_unit"] = "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
                roof["algorithmic_bytes_per_launch"] = d_by / max(d_n, 1)
        except Exception:
            pass
        roof["kernel"] = dom_name + (f" ({n_inst} tile instantiations)" if n_inst > 1 else "")
        roof["launches_per_step"] = d_n / max(args.prof_steps, 1)
        roof["avg_launch_ms"] = d_ms / max(d_n, 1)
        roof["share_of_kernel_time"] = d_ms / total_ms
        mfma = {k: r for k, r in rows.items() if r[2] > 0 and any(t in k for t in ("igemm", "wgrad", "conv3x3_pp", "convt4"))}
        mf_ms = sum(r[0] for r in mfma.values())
        mf_fl = sum(r[2] for r in mfma.values())
        roof["all_mfma_kernels_tflops"] = mf_fl / (mf_ms * 1e-3) / 1e12 if mf_ms > 0 else None
        # the same figure for every kernel family that takes >= 2 % of the kernel time (same HIP-event rows; MFMA kernels against the
        # dense fp16 peak, the others against the HBM peak with their algorithmic bytes)
        others = []
        for name, f in sorted(fam.items(), key=lambda kv: -kv[1][0]):
            ms, n, fl, by = f[:4]
            if ms < 0.02 * total_ms or ms <= 0:
                continue
            if fl > 0:
                ach, peak, unit, bound = fl / (ms * 1e-3) / 1e12, PEAK_MFMA_F16_TFLOPS, "TFLOP/s", "mfma"
            elif by > 0:
                ach, peak, unit, bound = by / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
            else:
                continue
            others.append({"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                           "ms_per_step": ms / max(args.prof_steps, 1), "launches_per_step": n / max(args.prof_steps, 1)})
        roof["by_kernel"] = others
        # ---- the graded region (BASELINE north_star / SURVEY 8d): the encoder conv stack = its 12 convolutions x {forward, data
        # gradient, weight gradient (+ slab reduce)}, 8 649.18 MFLOP per image and step.  Same HIP-event method, second pass with
        # per-layer record names (lo_prof_enable(2)); kernels run one after the other on the launch stream while profiling, so this
        # is the sum of their own durations.  Encoder layers are kind 0 (3x3 s1) / kind 1 (3x3 s2); the first conv has its own kernels.
        if args.prof_steps > 0:
            import re
            saved_sync, st.grad_sync = st.grad_sync, None
            _lib.lib.lo_prof_enable(2)
            run(args.prof_steps)
            torch.cuda.synchronize()
            lrows = collect_profile(_lib.lib)
            _lib.lib.lo_prof_enable(0)
            st.grad_sync = saved_sync
            pat = re.compile(r"^(fwd|dgrad L\d+|wgrad L\d+) kind[01] ")
            parts = {"forward": 0.0, "data_gradient": 0.0, "weight_gradient": 0.0}
            n_launch = 0
            for k, r in lrows.items():
                which = None
                if pat.match(k):
                    which = "forward" if k.startswith("fwd") else ("data_gradient" if k.startswith("dgrad") else "weight_gradient")
                elif k.startswith("lo_first_conv_fwd"):
                    which = "forward"
                elif k.startswith("lo_first_conv_wgrad"):
                    which = "weight_gradient"
                if which:
                    parts[which] += r[0] / args.prof_steps
                    n_launch += r[1]
            enc_ms = sum(parts.values())
            enc_flops = 8649.18e6 * B
            if enc_ms > 0:
                roof["encoder_conv_stack"] = {"flops": enc_flops, "ms": enc_ms, "tflops": enc_flops / (enc_ms * 1e-3) / 1e12,
                                              "frac": enc_flops / (enc_ms * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS, "peak": PEAK_MFMA_F16_TFLOPS,
                                              "ms_by_pass": parts, "launches_per_step": n_launch / args.prof_steps,
                                              "note": "12 encoder convs x {fwd, dgrad, wgrad + slab reduce}; HIP events per launch, serial on the launch stream; "
                                                      "GroupNorm / Mish passes not included (fused epilogues are)"}
            if args.breakdown:
                for k, r in sorted(lrows.items(), key=lambda kv: -kv[1][0]):
                    floor_ms = 1e3 * max(r[2] / (PEAK_MFMA_F16_TFLOPS * 1e12), r[3] / (PEAK_HBM_GBS * 1e9))
                    sys.stderr.write(f"[layer] {k:52s} {r[0] / args.prof_steps:9.4f} ms/step  n={r[1] // args.prof_steps:3d}  "
                                     f"{(r[2] / (r[0] * 1e-3) / 1e12 if r[0] > 0 else 0):8.1f} TFLOP/s  "
                                     f"{(r[3] / (r[0] * 1e-3) / 1e9 if r[0] > 0 else 0):8.1f} GB/s  "
                                     f"floor {floor_ms / args.prof_steps:7.4f} ms ({(floor_ms / r[0] if r[0] > 0 else 0):4.2f} of floor)\n")
                sys.stderr.write(f"[layer] sum of kernel time {sum(r[0] for r in lrows.values()) / args.prof_steps:.4f} ms/step\n")
        if args.breakdown:
            # floor = the larger of flops / MFMA peak and algorithmic bytes / HBM peak for the launches of the row; "of floor" is how
            # close the measured time comes to it (LO_PROF_LAYERS=1 splits the conv rows per layer geometry)
            for k, r in sorted(rows.items(), key=lambda kv: -kv[1][0]):
                floor_ms = 1e3 * max(r[2] / (PEAK_MFMA_F16_TFLOPS * 1e12), r[3] / (PEAK_HBM_GBS * 1e9))
                sys.stderr.write(f"{k:52s} {r[0] / max(args.prof_steps, 1):9.4f} ms/step  n={r[1] // max(args.prof_steps, 1):3d}  "
                                 f"{(r[2] / (r[0] * 1e-3) / 1e12 if r[0] > 0 else 0):8.1f} TFLOP/s  "
                                 f"{(r[3] / (r[0] * 1e-3) / 1e9 if r[0] > 0 else 0):8.1f} GB/s  "
                                 f"floor {floor_ms / max(args.prof_steps, 1):7.4f} ms ({(floor_ms / r[0] if r[0] > 0 else 0):4.2f} of floor)\n")
            sys.stderr.write(f"sum of kernel time {total_ms / max(args.prof_steps, 1):.4f} ms/step\n")
        out = {
            "metric": "128x128 sprites/sec, VAE training step (train_hybrid.py _process_batch), latent_dim=512",
            "value": world * B * args.steps / elapsed,
            "unit": "sprites/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_steps_run": warmup_run,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "fp8 (e4m3 forward operands) / f16 backward",
            "data": "synthetic",
            "config": {"workload": f"VAE-only training step (fwd+MSE/KL+bwd+clip+AdamW), per-GPU batch {B}, latent_dim {args.latent}, "
                                   "teacher scalar 0 (--reward_scale 0 --quality_weight 0; the full hybrid step of BASELINE config 3 is the config3_full_hybrid object at N=1), "
                                   "gradient_accumulation_steps 1, " + ("fp16 MFMA operands" if args.precision == "fp16" else
                                   "e4m3 MFMA operands in the forward convs with Cin % 128 == 0, fp16 elsewhere") + " / fp32 accumulate, fp32 master weights",
                       "global_batch": world * B, "latent_dim": args.latent,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       # how the three Linear layers' weight gradients (82 % of the parameters) exist in this run: never written
                       # (Gram-matrix norm + AdamW forming the tiles, csrc/lo_lowrank.hip), exchanged as factors, or written out
                       "linear_weight_gradients": ("factored (never materialised)" if getattr(st, "linear_factored", False) and world == 1
                                                   else ("factors all-gathered, averaged matrices formed per rank" if getattr(st, "dp_factored", False) and world > 1
                                                         else "materialised"))},
            "roofline": roof,
            "host_enqueue_ms": 1e3 * host_enqueue_s / args.steps,     # host time per step inside the timed region (includes queue back-pressure)
            "host_enqueue_ms_idle_queue": 1e3 * host_idle_s,          # the host's own cost: 5 steps enqueued into an empty queue (launch-bound only when THIS nears ms_per_step)
            "final_metrics": {k: met[k] for k in ("recon_loss", "kl_loss", "grad_norm")},
        }
        if world > 1:
            out["data_parallel"] = dp_info
        if world == 1 and args.fp8_steps > 0 and args.precision == "fp16":
            # BASELINE config 5: the same VAE-only step with e4m3 operands in the forward convs; loss parity against the fp16 mode
            # from identical weights, inputs and noise (first step), then throughput
            def fresh(prec):
                torch.manual_seed(42)
                m = LunarisCoreVAE(latent_dim=args.latent, mfma_precision=prec).to("cuda")
                return m, VAEStepper(m, lr=1e-4, min_lr=1e-6, scheduler_t0=10, weight_decay=0.01, max_grad_norm=1.0, recon_weight=1.0,
                                     kl_weight=0.1, gradient_accumulation_steps=1, pipeline_optimizer=pipeline)
            first = {}
            for prec in ("fp16", "fp8"):
                m8, s8 = fresh(prec)
                s8.step(pool[0], batch_idx=0)
                first[prec] = s8.metrics()
                if prec == "fp16":
                    del m8, s8
            for i in range(1, 60):
                s8.step(pool[i % len(pool)], batch_idx=i)
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            for i in range(args.fp8_steps):
                s8.step(pool[i % len(pool)], batch_idx=i)
            torch.cuda.synchronize()
            dt8 = (time.perf_counter() - t8) / args.fp8_steps
            out["config5_fp8_forward"] = {
                "value": B / dt8, "unit": "sprites/s", "ms_per_step": 1e3 * dt8, "steps": args.fp8_steps,
                "workload": f"VAE-only step, batch {B}, latent {args.latent}: OCP e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4) in the "
                            "forward convs with Cin % 128 == 0 (all but the 128->64 transposed conv, which stays on its patch-resident fp16 kernel), fp16 backward",
                "fp8_forward_conv_layers": m8._engine(B).fp8_layers, "forward_conv_layers": 16,
                "loss_parity_vs_f16_first_step": {k: abs(first["fp8"][k] - first["fp16"][k]) for k in ("recon_loss", "kl_loss")},
                "f16_first_step": {k: first["fp16"][k] for k in ("recon_loss", "kl_loss", "grad_norm")},
                "fp8_first_step": {k: first["fp8"][k] for k in ("recon_loss", "kl_loss", "grad_norm")}}
            del m8, s8
            torch.cuda.empty_cache()
        if world == 1 and args.config2_steps > 0:
            # BASELINE config 2: batch 32, latent 256, VAE-only (teacher weight 0), one GPU
            torch.manual_seed(42)
            m2 = LunarisCoreVAE(latent_dim=256).to("cuda")
            s2 = VAEStepper(m2, lr=1e-4, min_lr=1e-6, scheduler_t0=10, weight_decay=0.01, max_grad_norm=1.0, recon_weight=1.0,
                            kl_weight=0.1, gradient_accumulation_steps=1, pipeline_optimizer=pipeline)
            pool2 = [p_[:32].contiguous() for p_ in pool]
            for i in range(100):
                s2.step(pool2[i % len(pool2)], batch_idx=i)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(args.config2_steps):
                s2.step(pool2[i % len(pool2)], batch_idx=i)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t2) / args.config2_steps
            out["config2_b32_l256"] = {"value": 32 / dt2, "unit": "sprites/s", "ms_per_step": 1e3 * dt2, "steps": args.config2_steps,
                                       "workload": "VAE-only step (fwd+MSE/KL+bwd+clip+AdamW), batch 32, latent 256, fp16 MFMA operands / fp32 accumulate",
                                       "recon_loss": s2.metrics()["recon_loss"]}
            del m2, s2, pool2
            torch.cuda.empty_cache()
        if world == 1 and args.hybrid_steps > 0:
            # BASELINE config 3: batch 64, latent 512, embedding_dim 256, feature_dim 128, teacher on (both teacher forwards of
            # _process_batch, reward/advantage, gate + quality-head update).  Headline of this leg: the reference's defaults,
            # i.e. teacher dropout 0.1 in train mode (every teacher conv in full); the dropout-free fast path is reported beside it.
            from lunaris_orion_amd.teacher import LunarMoETeacher
            from lunaris_orion_amd.trainer import HybridStepper
            del st
            torch.cuda.empty_cache()

            def hybrid_leg(p_drop, steps, prof_steps, precision="fp16", feature_dim=128, warm=6, full_bwd=False):
                torch.manual_seed(42)
                teacher = LunarMoETeacher(num_experts=4, feature_dim=feature_dim, embedding_dim=256, dropout_rate=p_drop, mfma_precision=precision).to("cuda").train()
                vae_m = LunarisCoreVAE(latent_dim=args.latent, mfma_precision=precision).to("cuda")    # same seed: same initial weights in every leg
                hs = HybridStepper(vae_m, teacher, gradient_accumulation_steps=1, pipeline_optimizer=pipeline, teacher_full_backward=full_bwd)
                hs.step(pool[0], batch_idx=0)
                first = hs.metrics()                       # first step from identical weights / sprites / mask stream: comparable across legs
                for i in range(1, warm):
                    hs.step(pool[i % len(pool)], batch_idx=i)
                torch.cuda.synchronize()
                th = time.perf_counter()
                for i in range(steps):
                    hs.step(pool[i % len(pool)], batch_idx=i)
                enq = time.perf_counter() - th
                torch.cuda.synchronize()
                dth = (time.perf_counter() - th) / steps
                hm = hs.metrics()
                leg = {"value": B / dth, "unit": "sprites/s", "ms_per_step": 1e3 * dth, "steps": steps, "host_enqueue_ms": 1e3 * enq / steps,
                       "teacher_dropout": p_drop, "teacher_path": {0: "sparse shortcuts", 1: "dense", 2: "dropout (every conv in full)"}[teacher.last_path(B)],
                       "quality_scores": hm["quality_scores"], "recon_loss": hm["recon_loss"]}
                if prof_steps > 0:
                    _lib.lib.lo_prof_enable(1)
                    for i in range(prof_steps):
                        hs.step(pool[i % len(pool)], batch_idx=i)
                    torch.cuda.synchronize()
                    rw = collect_profile(_lib.lib)
                    _lib.lib.lo_prof_enable(0)
                    tot = sum(r[0] for r in rw.values())
                    if full_bwd:
                        leg["kernel_ms_per_step"] = tot / prof_steps
                        leg["top_kernels_ms_per_step"] = {k: round(r[0] / prof_steps, 3) for k, r in sorted(rw.items(), key=lambda kv: -kv[1][0])[:12]}
                    # the teacher's full-resolution 3x3 convolutions (one kernel, lo_conv3x3_pp, under per-call-site profiler names)
                    conv = [r for k, r in rw.items() if k.startswith("t_conv1") or k.startswith("t_conv2 (dense") or k.startswith("t_conv2 (generic")]
                    c_ms, c_n, c_fl = sum(r[0] for r in conv), sum(r[1] for r in conv), sum(r[2] for r in conv)
                    if c_ms > 0:
                        ach = c_fl / (c_ms * 1e-3) / 1e12
                        leg["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F16_TFLOPS,
                                           "traffic": None, "kernel": "lo_conv3x3_pp (teacher 3x3 convs 128->128 at 128x128)",
                                           "launches_per_step": c_n / prof_steps, "avg_launch_ms": c_ms / max(c_n, 1),
                                           "share_of_kernel_time": c_ms / tot, "kernel_ms_per_step": tot / prof_steps}
                        # the teacher's conv stack as a region (the counterpart of roofline.encoder_conv_stack): every 3x3 convolution of
                        # both teacher forwards of the step, algorithmic FLOPs of the launches that ran
                        leg["roofline"]["teacher_conv_stack"] = {"flops": c_fl / prof_steps, "ms": c_ms / prof_steps, "tflops": ach,
                                                                 "frac": ach / PEAK_MFMA_F16_TFLOPS, "launches_per_step": c_n / prof_steps}
                leg["first_step"] = {k: first[k] for k in ("recon_loss", "kl_loss", "quality_scores", "teacher_loss", "baseline", "grad_norm")}
                del hs, teacher, vae_m
                torch.cuda.empty_cache()
                return leg
            main_leg = hybrid_leg(0.1, args.hybrid_steps, 2)
            fast_leg = hybrid_leg(0.0, args.hybrid_steps, 0)
            if args.fp8_steps > 0 and "config5_fp8_forward" in out:
                # BASELINE config 5 on the full hybrid step: e4m3 operands in the VAE's forward convs (Cin % 128 == 0) AND in the
                # teacher's 24 full-resolution 3x3 convs per forward (where the FLOPs of this step are); same seeds as the fp16 leg
                f8 = hybrid_leg(0.1, args.hybrid_steps, 2, precision="fp8")
                f8["workload"] = ("full hybrid step as config3_full_hybrid (teacher dropout 0.1), e4m3 operands in the teacher's 3x3 convolutions and in the "
                                  "VAE's forward convs with Cin % 128 == 0; fp16 backward, fp32 statistics")
                f8["loss_parity_vs_f16_first_step"] = {k: abs(f8["first_step"][k] - main_leg["first_step"][k]) for k in f8["first_step"]}
                if "roofline" in f8:
                    f8["roofline"]["peak"] = 5000.0           # dense fp8 MFMA peak (MI355X_MICROARCH.md)
                    f8["roofline"]["frac"] = f8["roofline"]["achieved"] / 5000.0
                    f8["roofline"]["kernel"] = "lo_conv3x3_pp<f8> (teacher 3x3 convs, e4m3 operands)"
                out["config5_fp8_forward"]["full_hybrid"] = f8
            main_leg["workload"] = (f"full hybrid _process_batch at the reference's defaults: VAE step + 2 train-mode teacher forwards (feature_dim 128, 4 experts, "
                                    f"embedding_dim 256, teacher dropout 0.1 applied at all six sites) + reward/advantage + gate/quality-head update, batch {B}, latent {args.latent}")
            main_leg["without_teacher_dropout"] = {k: fast_leg[k] for k in ("value", "ms_per_step", "teacher_dropout", "teacher_path", "host_enqueue_ms")}
            main_leg["without_teacher_dropout"]["note"] = "LunarMoETeacher(dropout_rate=0): constant-field shortcuts valid; NOT the reference's default step"
            out["config3_full_hybrid"] = main_leg
            if args.fullbwd_steps > 0:
                # the same step with the teacher trained "as documented" (--teacher_full_backward): trunk recomputed
                # block by block and differentiated, clip + AdamW over all 234 live teacher tensors
                fb = hybrid_leg(0.1, args.fullbwd_steps, 1, warm=2, full_bwd=True)
                fb["workload"] = ("config3_full_hybrid with --teacher_full_backward: + recomputation and backward of the feature extractor and the 12 ExpertBlocks "
                                  "(non-reentrant checkpoint semantics of lunar_evaluator.py:194-197, 266-275, 411-414) + AdamW over every live teacher parameter")
                out["config3_full_hybrid"]["teacher_full_backward"] = fb
            if args.highend_steps > 0:
                # README.md:102-118 "High-End" recipe: batch 64, latent 512, embedding 256, feature_dim 512 (2 TFLOP per image and teacher
                # forward: 262 TFLOP per step), teacher dropout 0.1; generic full-resolution teacher path
                he = hybrid_leg(0.1, args.highend_steps, 1, feature_dim=512, warm=2)
                he["workload"] = (f"README High-End recipe: full hybrid _process_batch, batch {B}, latent {args.latent}, embedding_dim 256, feature_dim 512, 4 experts, "
                                  "teacher dropout 0.1 (generic full-resolution teacher path, fp16 operands)")
                if "roofline" in he:
                    he["roofline"]["kernel"] = "lo_conv3x3_pp / lo_igemm_nt (teacher 3x3 convs 128->512 and 512->512 at 128x128)"
                out["readme_high_end_feature_dim512"] = he
            if not args.no_cpu_baseline and args.cpu_hybrid_steps > 0:
                out["config3_full_hybrid"]["cpu_baseline"] = cpu_baseline_hybrid(args.latent, steps=args.cpu_hybrid_steps)
        if world == 1 and not args.no_cpu_baseline:
            # the oracle at this leg's shape and at config 1's shape (batch 8, latent 256); 3 warm-up + N timed, median
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch or B, args.latent, args.cpu_steps, warm=args.cpu_warmup)
            out["cpu_baseline_config1"] = cpu_baseline(8, 256, args.cpu_steps, warm=args.cpu_warmup)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
