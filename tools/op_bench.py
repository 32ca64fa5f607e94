#!/usr/bin/env python
"""Micro-benchmark of one implicit-GEMM op through the C ABI (tuning aid; also the target of rocprofv3 --pmc runs).

  python tools/op_bench.py --op conv|wgrad --kind 0 --B 64 --H 32 --Cin 128 --Cout 128 --iters 20
  python tools/op_bench.py --op attn --B 64 --H 8 --Cin 512        (SelfAttention2d forward on a [B, Cin, H, H] map)
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lunaris_orion_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="conv")
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--Cin", type=int, default=128)
    ap.add_argument("--Cout", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--stats", type=int, default=1)
    a = ap.parse_args()
    lib = _lib.lib
    B, H, Cin, Cout, kind = a.B, a.H, a.Cin, a.Cout, a.kind
    Ho = H if kind in (0, 3, 6) else (H // 2 if kind in (1, 5) else 2 * H)
    st = _lib.stream_ptr()
    if a.op == "attn":
        # SelfAttention2d forward (lunar_generate.py:56-78): roofline line of the fused kernel.  FLOPs = QK^T + PV (the three
        # 1x1 projections are separate launches and are timed with it; their FLOPs are counted too)
        Cc, N, D = Cin, H * H, Cin // 8
        xa = torch.randn(B, Cc, N, device="cuda")
        ws = [torch.randn(D, Cc, device="cuda") * Cc ** -0.5, torch.zeros(D, device="cuda"), torch.randn(D, Cc, device="cuda") * Cc ** -0.5,
              torch.zeros(D, device="cuda"), torch.randn(Cc, Cc, device="cuda") * Cc ** -0.5, torch.zeros(Cc, device="cuda"), torch.full((1,), 0.7, device="cuda")]
        q, k = torch.empty(B, D, N, device="cuda"), torch.empty(B, D, N, device="cuda")
        v, out = torch.empty(B, Cc, N, device="cuda"), torch.empty(B, Cc, N, device="cuda")
        fl_core = 2.0 * B * N * N * (D + Cc)
        fl_proj = 2.0 * B * N * Cc * (2 * D + Cc)

        def run_attn():
            _lib.check(lib.lo_selfattn2d_forward(xa.data_ptr(), *[t.data_ptr() for t in ws], q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, Cc, N, st))
        for _ in range(3):
            run_attn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run_attn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        byts = 4.0 * B * N * (2.0 * D + 3.0 * Cc)
        print(f"attn B={B} C={Cc} N={N}: {ms * 1e3:.1f} us/call (one q|k|v projection launch + fused attention)  {(fl_core + fl_proj) / ms / 1e9:.2f} TFLOP/s "
              f"= {(fl_core + fl_proj) / ms / 1e9 / 2500.0:.4f} of the dense fp16 MFMA peak;  {byts / ms / 1e6:.1f} GB/s of q/k/v/x/out traffic = {byts / ms / 1e6 / 8000.0:.4f} of HBM peak")
        return
    x = (torch.randn(B, H, H, Cin, device="cuda") * 0.5).half()
    if a.op == "conv":
        n = lib.lo_packed_weight_elems_for(kind, B, H, H, Cin, Cout)
        wp = (torch.randn(n, device="cuda") * 0.05).half()
        bias = torch.zeros(Cout, device="cuda")
        out = torch.empty(B, Ho, Ho, Cout, dtype=torch.float16, device="cuda")
        part = torch.empty(B * 4096 * 16, device="cuda") if a.stats and kind in (0, 1, 2) else None
        mt = C.c_int(0)
        fl = 2.0 * out.numel() * (n / Cout)

        def run():
            _lib.check(lib.lo_conv_forward(kind, B, H, H, Cin, Cout, x.data_ptr(), wp.data_ptr(), bias.data_ptr(), None,
                                           out.data_ptr(), _lib.ptr(part), C.byref(mt), st))
    else:
        dy = (torch.randn(B, Ho, Ho, Cout, device="cuda") * 0.1).half()
        nb = lib.lo_wgrad_slab_bytes_for(kind, B, H, H, Cin, Cout)
        slab = torch.empty(nb // 4 + 1, device="cuda")
        n = lib.lo_packed_weight_elems_for(kind, B, H, H, Cin, Cout)
        grad = torch.empty(n, device="cuda")
        fl = 2.0 * dy.numel() * (n / Cout)

        def run():
            _lib.check(lib.lo_conv_wgrad(kind, B, H, H, Cin, Cout, x.data_ptr(), dy.data_ptr(), slab.data_ptr(), grad.data_ptr(), 1.0, st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{a.op} kind={kind} B={B} H={H} Cin={Cin} Cout={Cout}: {ms * 1e3:.1f} us/call  {fl / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
