"""Time fgvc_conv_s2_split_f32 alone (layer-2 shapes of the 480p clip) against MIOpen's f32 strided convolution."""
import sys, torch
sys.path.insert(0, ".")
import torch.nn.functional as F
from fgvc_amd import ops
dev = torch.device("cuda", 0)
N, Cin, Cout, H, W = 8, 64, 128, 240, 427
g = torch.Generator().manual_seed(0)
x = torch.randn(N, Cin, H, W, generator=g).to(dev)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
xs = ops.nchw_to_split_nhwc(x)
x_cl = x.contiguous(memory_format=torch.channels_last)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


import os
DEBUGS = [int(v) for v in os.environ.get("S2_DEBUG", "0").split(",")]
for KS in (3, 1):
    wt = torch.randn(Cout, Cin, KS, KS, generator=g).to(dev) * 0.05
    bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
    wp, bias = ops.prepare_conv_s2(wt, bn)
    out_s = ops.alloc_split_nhwc(N, Cout, Ho, Wo, dev)
    out_f = ops.alloc_nhwc(N, Cout, Ho, Wo, dev)
    w_cl = wt.contiguous(memory_format=torch.channels_last)
    flops = 2.0 * N * Ho * Wo * Cout * Cin * KS * KS
    for name, fn in (("split out", lambda: ops.conv_s2_split(xs, wp, bias, H, W, True, out_split=out_s)),
                     ("f32 out", lambda: ops.conv_s2_split(xs, wp, bias, H, W, False, out_f32=out_f)),
                     ("MIOpen f32 NHWC", lambda: F.conv2d(x_cl, w_cl, bias, 2, KS // 2))):
        for dbg in (DEBUGS if "MIOpen" not in name else [0]):
            ops.set_option("conv_s2_debug", dbg)
            ms = timeit(fn)
            ops.set_option("conv_s2_debug", 0)
            print(f"KS={KS} {name:16s} debug={dbg:2d} {ms:.4f} ms   {3 * flops / ms / 1e9:.0f} bf16 TFLOP/s-equivalent")

# 64 -> 64 3x3 stride 1 at 240 x 427: generic kernel vs register-resident weights
Hh, Ww = 240, 427
wt = torch.randn(64, 64, 3, 3, generator=g).to(dev) * 0.05
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
wg, bg = ops.prepare_conv_split(wt, bn)
w6, b6 = ops.prepare_conv64(wt, bn)
o_s, o_f = ops.alloc_split_nhwc(N, 64, Hh, Ww, dev), ops.alloc_nhwc(N, 64, Hh, Ww, dev)
r_f = torch.randn(N, Hh, Ww, 64, device=dev)
for name, fn in (("generic, split out", lambda: ops.conv_split(xs, wg, bg, Hh, Ww, True, out_split=o_s)),
                 ("conv64,  split out", lambda: ops.conv64_split(xs, w6, b6, Hh, Ww, True, out_split=o_s)),
                 ("generic, residual + f32 + split", lambda: ops.conv_split(xs, wg, bg, Hh, Ww, True, residual=r_f, out_split=o_s, out_f32=o_f)),
                 ("conv64,  residual + f32 + split", lambda: ops.conv64_split(xs, w6, b6, Hh, Ww, True, residual=r_f, out_split=o_s, out_f32=o_f))):
    print(f"64->64 3x3 {name:34s} {timeit(fn):.4f} ms")
