"""The teacher's 3x3 convolution (lunar_evaluator.py:242-251: Conv2d(128, 128, k3, p1) on 128 x 128 maps, batch 64: 309 GFLOP per launch,
48 launches per hybrid step) on the vendor's kernels: hipBLASLt at the implicit-GEMM shape and MIOpen's convolution forward / data
gradient / weight gradient, channels-last fp16.  Tools only (the reference point for `teacher_conv_stack.frac`):
  python tools/vendor_teacher_conv.py > profiles/r04_vendor_teacher_conv.txt
"""
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
B, C, H = 64, 128, 128
def timed(fn, warm=20, iters=20):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
x = torch.randn(B, C, H, H, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(C, C, 3, 3, device="cuda", dtype=torch.float16) * 0.03).contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, C, H, H, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
M, N, K = B * H * H, C, C * 9
flop = 2.0 * M * N * K
a = torch.randn(M, K, device="cuda", dtype=torch.float16); b = torch.randn(K, N, device="cuda", dtype=torch.float16)
t = timed(lambda: torch.mm(a, b)); print(f"hipBLASLt mm {M}x{N}x{K}: {t:.1f} us = {flop / t / 1e6:.0f} TFLOP/s")
at = torch.randn(K, M, device="cuda", dtype=torch.float16); bb = torch.randn(M, N, device="cuda", dtype=torch.float16)
t = timed(lambda: torch.mm(at, bb)); print(f"hipBLASLt mm (wgrad shape) {K}x{N}x{M}: {t:.1f} us = {flop / t / 1e6:.0f} TFLOP/s")
t = timed(lambda: torch.ops.aten.convolution(x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)); print(f"MIOpen conv fwd: {t:.1f} us = {flop / t / 1e6:.0f} TFLOP/s")
t = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])); print(f"MIOpen conv dgrad: {t:.1f} us = {flop / t / 1e6:.0f} TFLOP/s")
t = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])); print(f"MIOpen conv wgrad: {t:.1f} us = {flop / t / 1e6:.0f} TFLOP/s")
