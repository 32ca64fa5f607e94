"""CLI contract: the 35 reference flags with the reference defaults (train_hybrid.py:1076-1133); a small end-to-end
run on synthetic sprites on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REFERENCE_DEFAULTS = {
    "output_dir": "output", "resume_from": None, "batch_size": 16, "gradient_accumulation_steps": 2, "chunk_size": 32,
    "num_epochs": 100, "num_workers": 4, "seed": 42, "compile": False, "mixed_precision": False, "latent_dim": 256,
    "embedding_dim": 64, "feature_dim": 128, "num_experts": 4, "vae_lr": 1e-4, "teacher_lr": 1e-4, "min_lr": 1e-6,
    "weight_decay": 0.01, "max_grad_norm": 1.0, "scheduler_t0": 10, "recon_weight": 1.0, "kl_weight": 0.1,
    "quality_weight": 0.5, "log_every": 100, "save_every": 1000, "sample_every": 500, "keep_n_checkpoints": 5,
    "early_stopping_patience": 7, "eval_save_freq": 500, "reward_scale": 0.1, "semantic_weight": 0.5,
    "baseline_momentum": 0.9, "force_cpu": False, "memory_efficient": False,
}


def test_cli_flags_and_defaults_match_reference():
    sys.path.insert(0, ROOT)
    import train_hybrid
    args = train_hybrid.build_parser().parse_args(["--data_dir", "x"])
    for k, v in REFERENCE_DEFAULTS.items():
        assert getattr(args, k) == v, k
    assert len(REFERENCE_DEFAULTS) + 1 == 35      # + data_dir
    assert args.generate_samples == 0 and args.max_steps == 0 and not args.vae_only      # additions default to off
    assert args.mfma_precision == "fp16"


def test_unsupported_feature_dim_is_refused_loudly(tmp_path):
    sys.path.insert(0, ROOT)
    import train_hybrid
    with pytest.raises(SystemExit) as e:
        train_hybrid.main(["--data_dir", str(tmp_path), "--feature_dim", "384"])
    assert "feature_dim" in str(e.value)


def _make_data(d, n=24):
    rng = np.random.default_rng(0)
    np.save(os.path.join(d, "sprites_000.npy"), rng.integers(0, 256, (n, 128, 128, 3), dtype=np.uint8))
    with open(os.path.join(d, "labels_000.csv"), "w") as f:
        f.write("filename,category,prompt,seed,pixel_size,guidance_scale,pag_scale,num_steps\n")
        for i in range(n):
            f.write(f"s{i}.png,cat,prompt {i},{i},4,5.0,2.0,20\n")


@pytest.mark.gpu
def test_cli_end_to_end_on_gpu(tmp_path):
    data = tmp_path / "data"
    data.mkdir()
    _make_data(str(data))
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_hybrid.py"), "--data_dir", str(data), "--output_dir", str(out),
                        "--vae_only", "--batch_size", "4", "--gradient_accumulation_steps", "1", "--num_epochs", "2",
                        "--log_every", "1", "--latent_dim", "256"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    log = (out / "training.log").read_text()
    assert "Average Loss" in log and "nan" not in log.split("Average Loss")[1][:12]
    assert (out / "checkpoints" / "latest.pt").exists()
    import torch
    ck = torch.load(out / "checkpoints" / "latest.pt", map_location="cpu", weights_only=False)
    assert set(ck.keys()) >= {"global_step", "vae_state_dict", "teacher_state_dict", "vae_optimizer", "teacher_optimizer",
                              "vae_scheduler", "teacher_scheduler", "best_loss", "args"}
    assert len(ck["vae_state_dict"]) == 72
    # the optimizer / scheduler entries are torch.optim state_dicts: they load into the objects the reference builds
    from lunaris_orion_amd.vae import LunarisCoreVAE
    m = LunarisCoreVAE(256)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=1e-6)
    opt.load_state_dict(ck["vae_optimizer"])
    sch.load_state_dict(ck["vae_scheduler"])
    steps = 2 * (int(0.9 * 24) // 4)
    assert int(float(opt.state_dict()["state"][0]["step"])) == steps and sch.last_epoch == steps
    assert float(opt.state_dict()["state"][5]["exp_avg_sq"].abs().sum()) > 0
    # a VAE-only checkpoint still carries a structurally valid (untrained) teacher, so that the reference's `_load_checkpoint`
    # (train_hybrid.py:798-822: teacher state, teacher_optimizer.load_state_dict, teacher_scheduler.load_state_dict) goes through
    from lunaris_orion_amd.teacher import LunarMoETeacher
    t = LunarMoETeacher(num_experts=4, feature_dim=128, embedding_dim=64)
    assert len(ck["teacher_state_dict"]) == 351 and not t.load_state_dict(ck["teacher_state_dict"], strict=True).missing_keys
    topt = torch.optim.AdamW(t.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999))
    tsch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(topt, T_0=10, T_mult=2, eta_min=1e-6)
    topt.load_state_dict(ck["teacher_optimizer"])
    tsch.load_state_dict(ck["teacher_scheduler"])
    assert len(topt.state_dict()["param_groups"][0]["params"]) == 252 and tsch.last_epoch == 0
    # resume: two more epochs continue from the saved step count and write a comparison image + decoded samples
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_hybrid.py"), "--data_dir", str(data), "--output_dir", str(out),
                        "--vae_only", "--batch_size", "4", "--gradient_accumulation_steps", "1", "--num_epochs", "1", "--log_every", "1",
                        "--latent_dim", "256", "--resume_from", str(out / "checkpoints" / "latest.pt"), "--eval_save_freq", "3",
                        "--generate_samples", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ck2 = torch.load(out / "checkpoints" / "latest.pt", map_location="cpu", weights_only=True)
    assert ck2["global_step"] == ck["global_step"] + steps // 2
    assert int(float(ck2["vae_optimizer"]["state"][0]["step"])) == steps + steps // 2
    assert list((out / "eval_samples").glob("comparison_*.png")) and len(list(out.glob("sample_*.png"))) == 2


@pytest.mark.gpu
def test_cli_hybrid_end_to_end_on_gpu(tmp_path):
    """Default flags (teacher on) on a tiny dataset: the 12 metrics are logged and the teacher state is checkpointed."""
    data = tmp_path / "data"
    data.mkdir()
    _make_data(str(data), n=12)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_hybrid.py"), "--data_dir", str(data), "--output_dir", str(out),
                        "--batch_size", "2", "--gradient_accumulation_steps", "1", "--num_epochs", "1", "--log_every", "1",
                        "--max_steps", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import torch
    ck = torch.load(out / "checkpoints" / "latest.pt", map_location="cpu", weights_only=False)
    assert len(ck["teacher_state_dict"]) == 351
    # teacher optimizer state exists exactly for the 28 tensors that receive gradients in the reference (gate, quality heads)
    assert len(ck["teacher_optimizer"]["state"]) == 28 and len(ck["teacher_optimizer"]["param_groups"][0]["params"]) == 252
    assert "reward_state" in ck["lunaris_amd_extra"]
    log = (out / "training.log").read_text()
    assert "dropout_rate 0.1 in train mode" in log          # the reference's default: teacher dropout applied


@pytest.mark.gpu
def test_cli_hybrid_readme_high_end_flags_on_gpu(tmp_path):
    """README High-End recipe flags at a tiny batch: --feature_dim 256 / --embedding_dim 256 (the 512 variant is the same code
    path, tested at module level), and --teacher_dropout 0 selecting the dropout-free teacher."""
    data = tmp_path / "data"
    data.mkdir()
    _make_data(str(data), n=12)
    for extra, n_state in ((["--feature_dim", "256", "--embedding_dim", "256"], 379), (["--teacher_dropout", "0"], 351)):
        out = tmp_path / ("out" + str(n_state))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "train_hybrid.py"), "--data_dir", str(data), "--output_dir", str(out),
                            "--batch_size", "2", "--gradient_accumulation_steps", "1", "--num_epochs", "1", "--log_every", "1",
                            "--max_steps", "2"] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        import torch
        ck = torch.load(out / "checkpoints" / "latest.pt", map_location="cpu", weights_only=False)
        assert len(ck["teacher_state_dict"]) == n_state     # 351 + 4 experts x 7 shortcut entries (Conv1x1 weight / bias, BatchNorm 5)
        log = (out / "training.log").read_text()
        assert ("dropout-free fast path" in log) == (extra[0] == "--teacher_dropout")


@pytest.mark.gpu
def test_cli_teacher_full_backward_flag_on_gpu(tmp_path):
    """--teacher_full_backward: every teacher parameter on the loss path is trained -- the checkpoint carries AdamW
    state for all of them -- and a run resumed from that checkpoint continues (optimizer state of the full range restored)."""
    data = tmp_path / "data"
    data.mkdir()
    _make_data(str(data), n=12)
    out = tmp_path / "out"
    base = [sys.executable, os.path.join(ROOT, "train_hybrid.py"), "--data_dir", str(data), "--batch_size", "2", "--gradient_accumulation_steps", "1",
            "--num_epochs", "1", "--log_every", "1", "--teacher_full_backward"]
    r = subprocess.run(base + ["--output_dir", str(out), "--max_steps", "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import torch
    ck = torch.load(out / "checkpoints" / "latest.pt", map_location="cpu", weights_only=False)
    st = ck["teacher_optimizer"]["state"]
    moving = [i for i, s_ in st.items() if float(s_["exp_avg_sq"].abs().sum()) > 0]
    # 234 tensors have a gradient in the reference under non-reentrant checkpoints; the 24 relative-position tables among them have a
    # zero gradient (softmax-invariant), so 210 second moments have moved
    assert len(moving) == 210, len(moving)
    assert "teacher full backward on" in (out / "training.log").read_text()
    out2 = tmp_path / "out2"
    r = subprocess.run(base + ["--output_dir", str(out2), "--max_steps", "1", "--resume_from", str(out / "checkpoints" / "latest.pt")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ck2 = torch.load(out2 / "checkpoints" / "latest.pt", map_location="cpu", weights_only=False)
    k = next(i for i in moving)
    assert not torch.equal(ck2["teacher_optimizer"]["state"][k]["exp_avg"], st[k]["exp_avg"])      # the restored moments kept moving


@pytest.mark.gpu
def test_bench_line_contract(tmp_path):
    """bench.py prints ONE JSON line with the driver's keys plus `roofline` and (when asked) `cpu_baseline`."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--min-warmup", "0", "--batch", "8", "--latent", "256",
                        "--hybrid-steps", "1", "--cpu-batch", "2", "--cpu-steps", "1", "--cpu-warmup", "1", "--cpu-hybrid-steps", "0",
                        "--fp8-steps", "0", "--config2-steps", "0", "--highend-steps", "0", "--fullbwd-steps", "0"], capture_output=True, text=True, timeout=900)      # the legs this test does not look at: off
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["config3_full_hybrid"]["value"] > 0


@pytest.mark.gpu
def test_bench_two_rank_control_flow_on_one_gpu():
    """The driver's N > 1 launch (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) rehearsed with two
    ranks sharing the one GPU of the test box (gloo carries the collectives; RCCL refuses two ranks on one device): barriers,
    max-over-ranks timing, the rank-0-only roofline leg (which must not issue collectives) and the single JSON line."""
    import json
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--min-warmup", "0",
                        "--batch", "4", "--latent", "256", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["value"] > 0 and "cpu_baseline" not in d and "config3_full_hybrid" not in d
    assert d["roofline"]["launches_per_step"] > 0
    dp = d["data_parallel"]                      # what ran: ranks, devices, bytes handed over per backward phase, exposed exchange time
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and len(dp["devices"]) == 2
    # what each hand-over range ran as: the first one travels as factors (all-gather + local contraction), the others fall back to
    # all-reduce here (gloo carries all-to-all for CPU tensors only; under RCCL the ranges above 4 M elements take the direct form)
    assert dp["exchange_per_phase"][0] == "factors" and set(dp["exchange_per_phase"][1:]) <= {"allreduce"}, dp["exchange_per_phase"]
    assert len(dp["bytes_per_phase"]) == 3 and dp["exchange_exposed_ms"] is not None and d["host_enqueue_ms"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["vae_only", "hybrid", "hybrid_full_backward"])
def test_cli_two_ranks_on_one_gpu(tmp_path, mode):
    """train_hybrid.py under `torch.distributed.run` with two ranks (rehearsal: both on the test box's one GPU, gloo carrying the
    collectives): disjoint shards per rank, the phased backward with the three overlapped exchanges, the rank-consistent
    early-stopping mean, rank 0 writing the log and the checkpoint, both ranks ending together; with the teacher on also the
    5-float reward-mean exchange and the gate / quality-head gradient exchange."""
    data = tmp_path / "data"
    data.mkdir()
    _make_data(str(data), n=40)
    out = tmp_path / "out"
    env = dict(os.environ, LO_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", {"vae_only": "29543", "hybrid": "29545", "hybrid_full_backward": "29547"}[mode], os.path.join(ROOT, "train_hybrid.py"),
                        "--data_dir", str(data), "--output_dir", str(out),
                        "--batch_size", "4", "--gradient_accumulation_steps", "1", "--num_epochs", "2", "--log_every", "1",
                        "--latent_dim", "256"] + {"vae_only": ["--vae_only"], "hybrid": [], "hybrid_full_backward": ["--teacher_full_backward"]}[mode],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    log = (out / "training.log").read_text()
    assert log.count("Average Loss") == 2 and "Training completed." in log
    assert f"{(int(0.9 * 40) // 2) // 4} batches/epoch/rank" in log
    import torch
    ck = torch.load(out / "checkpoints" / "latest.pt", map_location="cpu", weights_only=False)
    assert ck["global_step"] == 2 * ((int(0.9 * 40) // 2) // 4)
    if mode == "hybrid_full_backward":
        # --teacher_full_backward under data parallel: the whole flat teacher gradient is averaged across the ranks and every live
        # tensor is updated (second moments of the expert / feature-extractor weights have moved)
        st = ck["teacher_optimizer"]["state"]
        assert sum(1 for s_ in st.values() if float(s_["exp_avg_sq"].abs().sum()) > 0) == 210
