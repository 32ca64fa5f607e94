"""What is reachable on the encoder conv stack with an implementation that is NOT this repository's.

For each of the 12 encoder convolutions (lunar_generate.py:36,41,95,102,109,116) x {forward, data gradient, weight gradient} at
batch 64, fp16 operands:
  * `torch.mm` (hipBLASLt / rocBLAS) at the implicit-GEMM shape (M, N, K) of the leg -- the contraction alone, operands already in
    GEMM layout (no im2col, no bias, no GroupNorm): an upper bound on what any conv kernel of that shape can reach with the vendor's
    tiles;
  * `aten::convolution` / `aten::convolution_backward` on channels-last fp16 tensors at the real geometry (MIOpen).
300 warm-up calls, then 100 timed calls between two HIP events on the current stream.  One context line: the CPU oracle's VAE
training step (oracle/vae_ref.py, plain PyTorch modules) moved to the GPU under fp16 autocast with fused AdamW -- the reference's own
way of running this step on a GPU.

Tools only: nothing here is imported by the package or by bench.py.  `--mine FILE` merges the per-layer table of this repository
(`bench.py --breakdown` under LO_PROF_LAYERS, profiles/rNN_bench_b64_l512_per_layer.txt) and flags every leg where the better vendor
number beats ours by more than 15 %.

  python tools/vendor_ceiling.py --mine profiles/r04_bench_b64_l512_per_layer.txt > profiles/r04_vendor_ceiling.txt
"""
import argparse
import os
import re
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (name, Cin, Cout, H_in, stride): the encoder's conv layers in forward order (lunar_generate.py:95-119)
LAYERS = []
_ch = (3, 64, 128, 256, 512)
_h = 128
for _s in range(4):
    LAYERS.append((f"down{_s + 1}.0 conv k3 s2", _ch[_s], _ch[_s + 1], _h, 2))
    _h //= 2
    LAYERS.append((f"down{_s + 1}.res.conv1 k3 s1", _ch[_s + 1], _ch[_s + 1], _h, 1))
    LAYERS.append((f"down{_s + 1}.res.conv2 k3 s1", _ch[_s + 1], _ch[_s + 1], _h, 1))


def timed(fn, warm, iters):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # microseconds per call


def parse_mine(path):
    """per-op microseconds of this repository's launches, keyed by (leg, stride, Hout, Cin, Cout) of the FORWARD conv"""
    out = {}
    if not path or not os.path.exists(path):
        return out
    pat = re.compile(r"\[layer\] (fwd|dgrad L\d+|wgrad L\d+) kind([01]) (\d+)x\d+ (\d+)->(\d+)\s+([0-9.]+) ms/step\s+n=\s*(\d+)")
    for line in open(path):
        m = pat.search(line)
        if not m:
            continue
        leg = m.group(1).split()[0]
        kind, ho, a, b, ms, n = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), float(m.group(6)), int(m.group(7))
        stride = 2 if kind == 1 else 1
        cin, cout = (b, a) if leg == "dgrad" else (a, b)        # the data gradient's geometry is printed as Cout -> Cin
        layers = n if leg == "fwd" else 1                       # "fwd" lines pool the layers of one shape; wgrad n = GEMM + reduce
        out.setdefault((leg, stride, ho, cin, cout), []).append(ms * 1e3 / layers)
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--warm", type=int, default=300)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--mine", default=None)
    ap.add_argument("--skip-oracle", action="store_true")
    args = ap.parse_args()
    assert torch.cuda.is_available()
    dev, B = "cuda", args.batch
    mine = parse_mine(args.mine)
    torch.backends.cudnn.benchmark = True            # MIOpen: search for the fastest solver of each shape
    print(f"# vendor ceiling, {torch.cuda.get_device_name()}, torch {torch.__version__}, batch {B}, fp16, {args.warm} warm-up + {args.iters} timed calls")
    print("# GEMM = torch.mm at the implicit-GEMM shape (hipBLASLt); conv = aten convolution / convolution_backward, channels-last (MIOpen)")
    print(f"# {'layer':28s} {'leg':6s} {'M':>7s} {'N':>5s} {'K':>6s} {'GFLOP':>7s} | {'GEMM us':>8s} {'TF/s':>6s} | {'conv us':>8s} {'TF/s':>6s} | {'ours us':>8s} {'TF/s':>6s} | best vendor / ours")
    tot = {"gemm": 0.0, "conv": 0.0, "best": 0.0, "ours": 0.0, "flop": 0.0}
    targets = []
    for name, cin, cout, h, stride in LAYERS:
        ho = h // stride
        M, N, K = B * ho * ho, cout, cin * 9
        flop = 2.0 * M * N * K
        x = torch.randn(B, cin, h, h, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.float16) * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, cout, ho, ho, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        kp = (K + 7) // 8 * 8                       # K = 27 padded to 32 for the GEMM stand-in of the first conv
        a_f = torch.randn(M, kp, device=dev, dtype=torch.float16)
        b_f = torch.randn(kp, N, device=dev, dtype=torch.float16)
        a_d = torch.randn(M, N, device=dev, dtype=torch.float16)
        b_d = torch.randn(N, kp, device=dev, dtype=torch.float16)
        a_w = torch.randn(kp, M, device=dev, dtype=torch.float16)
        b_w = torch.randn(M, N, device=dev, dtype=torch.float16)
        legs = {
            "fwd": (lambda: torch.mm(a_f, b_f), lambda: torch.ops.aten.convolution(x, w, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1)),
            "dgrad": (lambda: torch.mm(a_d, b_d),
                      lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])),
            "wgrad": (lambda: torch.mm(a_w, b_w),
                      lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])),
        }
        for leg, (gemm, conv) in legs.items():
            if leg == "dgrad" and cin == 3:
                continue                             # the reference needs no gradient with respect to the input images
            tg = timed(gemm, args.warm, args.iters)
            try:
                tc = timed(conv, args.warm, args.iters)
            except Exception as e:                   # a shape MIOpen refuses
                tc = float("nan")
                print(f"# {name} {leg}: aten convolution failed: {e}")
            ours = mine.get((leg, stride, ho, cin, cout))
            best = min(tg, tc) if tc == tc else tg
            tf = lambda us: flop / us / 1e6
            flag = ""
            if ours is not None:
                ratio = best / ours
                flag = f"{ratio:5.2f}" + ("   <-- vendor > 15 % faster" if ratio < 1 / 1.15 else "")
                if ratio < 1 / 1.15:
                    targets.append((name, leg, ours, best))
                tot["ours"] += ours
                tot["best"] += best
            tot["gemm"] += tg
            tot["conv"] += tc if tc == tc else 0.0
            tot["flop"] += flop
            print(f"  {name:28s} {leg:6s} {M:7d} {N:5d} {K:6d} {flop / 1e9:7.2f} | {tg:8.1f} {tf(tg):6.0f} | {tc:8.1f} {tf(tc):6.0f} | "
                  + (f"{ours:8.1f} {tf(ours):6.0f} | {flag}" if ours is not None else f"{'-':>8s} {'-':>6s} |"))
    print(f"# sums over the legs above: {tot['flop'] / 1e9:.1f} GFLOP; GEMM {tot['gemm'] / 1e3:.3f} ms = {tot['flop'] / tot['gemm'] / 1e6:.0f} TFLOP/s "
          f"({tot['flop'] / tot['gemm'] / 1e6 / 2500:.3f} of 2.5 PFLOP/s); MIOpen conv {tot['conv'] / 1e3:.3f} ms = {tot['flop'] / max(tot['conv'], 1e-9) / 1e6:.0f} TFLOP/s")
    if tot["ours"] > 0:
        print(f"# legs with a number of ours: ours {tot['ours'] / 1e3:.3f} ms against best-vendor-per-leg {tot['best'] / 1e3:.3f} ms")
        print("# named targets (vendor kernel more than 15 % faster than ours): " + ("none" if not targets else ""))
        for name, leg, o, b in targets:
            print(f"#   {name} {leg}: ours {o:.1f} us, vendor {b:.1f} us")

    if not args.skip_oracle:
        # context: the oracle's modules on the GPU, fp16 autocast + GradScaler + fused AdamW (how the reference runs this step on a GPU)
        from oracle import vae_ref as R
        L = 512
        P = {k: v.to(dev).requires_grad_(True) for k, v in R.closed_form_params(L).items()}
        opt = torch.optim.AdamW(list(P.values()), lr=1e-4, weight_decay=0.01, fused=True)
        scaler = torch.amp.GradScaler("cuda")
        xb = torch.rand(B, 3, 128, 128, device=dev) * 2 - 1
        eps = torch.randn(B, L, device=dev)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                recon, mu, logvar = R.vae_forward(xb, eps, P)
                loss = F.mse_loss(recon, xb) + 0.1 * (-0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp()))
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(list(P.values()), 1.0)
            scaler.step(opt)
            scaler.update()
        us = timed(step, 30, 30)
        print(f"# context: oracle VAE step (plain PyTorch ops: MIOpen / hipBLASLt / ATen) on this GPU, fp16 autocast + GradScaler + fused AdamW, "
              f"batch {B}, latent {L}: {us / 1e3:.2f} ms per step = {B / us * 1e6:.0f} sprites/s")


if __name__ == "__main__":
    main()
