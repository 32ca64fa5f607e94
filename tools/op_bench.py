#!/usr/bin/env python
"""Micro-benchmark of one implicit-GEMM op through the C ABI (tuning aid; also the target of rocprofv3 --pmc runs).

  python tools/op_bench.py --op conv|wgrad --kind 0 --B 64 --H 32 --Cin 128 --Cout 128 --iters 20
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
