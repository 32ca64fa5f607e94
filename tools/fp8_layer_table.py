#!/usr/bin/env python
"""Per-layer answer to "which of the 16 forward convolutions of the VAE should run on e4m3 operands at batch 64" (VERDICT r3 item 8;
BASELINE.json configs[4]).  For every forward conv / transposed conv of the VAE (lunar_generate.py:95-119, 169-187) at batch 64: the
fp16 launch the step uses (`lo_conv_forward`: the same launcher, which picks the fused-tap / patch-resident kernel where one applies)
against the e4m3 implicit GEMM (`lo_conv_forward_f8`, activation quantisation not included), 200 warm-up + 100 timed calls each between
two HIP events on the library's stream.  The e4m3 path exists for Cin % 128 == 0 and Cout % 64 == 0; the step's fp8 mode uses it on every such layer but the 128 -> 64
transposed conv, which the patch-resident fp16 kernel runs faster.

  python tools/fp8_layer_table.py > profiles/r04_fp8_per_layer.txt
"""
import ctypes as C
import os
import sys


import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lunaris_orion_amd import _lib  # noqa: E402

B = 64
lib = _lib.lib
st = _lib.stream_ptr()

# (name, kind, H_in, Cin, Cout): kind 0 = conv3 s1, 1 = conv3 s2, 2 = convT4 s2
LAYERS = []
ch, h = (3, 64, 128, 256, 512), 128
for s in range(4):
    if s > 0:
        LAYERS.append((f"enc.down{s + 1}.0 conv k3 s2", 1, h, ch[s], ch[s + 1]))
    h //= 2
    LAYERS.append((f"enc.down{s + 1}.res.conv1 k3 s1", 0, h, ch[s + 1], ch[s + 1]))
    LAYERS.append((f"enc.down{s + 1}.res.conv2 k3 s1", 0, h, ch[s + 1], ch[s + 1]))
dch = (512, 256, 128, 64, 32)
hh = 8
for s in range(4):
    LAYERS.append((f"dec.up{s + 1} convT k4 s2", 2, hh, dch[s], dch[s + 1]))
    hh *= 2


def timed(fn, warm=200, iters=100):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    print(f"# {torch.cuda.get_device_name()}, batch {B}: forward convolutions of the VAE, fp16 launch of the step vs e4m3 implicit GEMM (us per call)")
    print(f"# {'layer':34s} {'in':>9s} {'Cin':>4s} {'Cout':>4s} {'GFLOP':>7s} | {'fp16 us':>8s} {'TF/s':>6s} | {'e4m3 us':>8s} {'TF/s':>6s} | e4m3 / fp16")
    faster, applicable = 0, 0
    for name, kind, H, cin, cout in LAYERS:
        Ho = H if kind == 0 else (H // 2 if kind == 1 else 2 * H)
        taps = 16 if kind == 2 else 9
        flop = 2.0 * B * (H * H if kind == 2 else Ho * Ho) * cout * cin * taps
        x = (torch.randn(B, H, H, cin, device="cuda") * 0.5).half()
        wshape = (cin, cout, 4, 4) if kind == 2 else (cout, cin, 3, 3)
        w = torch.randn(*wshape, device="cuda") * (cin * taps) ** -0.5
        bias = torch.zeros(cout, device="cuda")
        n = lib.lo_packed_weight_elems_for(kind, B, H, H, cin, cout)
        wp = torch.empty(n, dtype=torch.float16, device="cuda")
        _lib.check(lib.lo_pack_weight_for(kind, B, H, H, cin, cout, w.data_ptr(), wp.data_ptr(), st), "pack")
        out = torch.empty(B, Ho, Ho, cout, dtype=torch.float16, device="cuda")
        part = torch.empty(B * 4096 * 16, dtype=torch.float32, device="cuda")
        mt = C.c_int()
        f16 = lambda: _lib.check(lib.lo_conv_forward(kind, B, H, H, cin, cout, x.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, out.data_ptr(),
                                                     part.data_ptr(), C.byref(mt), st), "conv")
        t16 = timed(f16)
        line = f"  {name:34s} {H:4d}x{H:<4d} {cin:4d} {cout:4d} {flop / 1e9:7.2f} | {t16:8.1f} {flop / t16 / 1e6:6.0f} | "
        if cin % 128 == 0 and cout % 64 == 0:
            applicable += 1
            x8 = torch.empty(x.numel(), dtype=torch.uint8, device="cuda")
            _lib.check(lib.lo_quantize_act_f8(x.data_ptr(), x8.data_ptr(), x.numel(), st), "quant")
            wp8 = torch.empty(n, dtype=torch.uint8, device="cuda")
            wsc = torch.empty(4 * cout, dtype=torch.float32, device="cuda")
            _lib.check(lib.lo_pack_weight_f8_for(kind, B, H, H, cin, cout, wp.data_ptr(), wp8.data_ptr(), wsc.data_ptr(), st), "pack8")
            f8 = lambda: _lib.check(lib.lo_conv_forward_f8(kind, B, H, H, cin, cout, x8.data_ptr(), wp8.data_ptr(), wsc.data_ptr(), bias.data_ptr(), None,
                                                           out.data_ptr(), part.data_ptr(), C.byref(mt), st), "conv8")
            t8 = timed(f8)
            faster += t8 < t16
            line += f"{t8:8.1f} {flop / t8 / 1e6:6.0f} | {t8 / t16:5.2f}" + ("   e4m3 faster" if t8 < t16 else "   fp16 kernel of the step is faster: stays fp16")
            if kind == 0 and cin == 128 and cout == 128:
                # the one shape the fused-tap kernel has an e4m3 form for (the teacher's 3x3 convs): both operand formats on the same kernel
                tt = {}
                for f8mode in (0, 1):
                    ft = lambda: _lib.check(lib.lo_conv3x3_fused_tap_forward(B, H, H, cin, cout, f8mode, (x8 if f8mode else x).data_ptr(), (wp8 if f8mode else wp).data_ptr(),
                                                                             wsc.data_ptr() if f8mode else None, bias.data_ptr(), 0, out.data_ptr(), None, st), "fused tap")
                    tt[f8mode] = timed(ft)
                line += f"\n  {'':34s} fused-tap kernel on this shape: fp16 {tt[0]:.1f} us, e4m3 {tt[1]:.1f} us"
        else:
            line += f"{'-':>8s} {'-':>6s} | n/a: " + ("Cin % 128 != 0 (a 128-wide K step of v_mfma_scale_f32_16x16x128_f8f6f4 does not fit the channel count)" if cin % 128 else "Cout % 64 != 0")
        print(line)
    print(f"# e4m3 applicable on {applicable} of {len(LAYERS)} layers, faster than the step's fp16 launch on {faster}")


if __name__ == "__main__":
    main()
