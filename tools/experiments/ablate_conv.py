"""Ablations of fgvc_conv_split_f32 (256->256 3x3 on 8x120x214) through the conv_debug option."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
shapes = {"256": (8, 256, 120, 214), "128": (8, 128, 120, 214), "64": (8, 64, 240, 427)}
N, C, H, W = shapes[os.environ.get("CONV_SHAPE", "256")]
wt = torch.randn(C, C, 3, 3, device=dev) * 0.02
bn = torch.nn.BatchNorm2d(C).eval().to(dev)
wp, bs = ops.prepare_conv_split(wt, bn)
xs = ops.nchw_to_split_nhwc(torch.relu(torch.randn(N, C, H, W, device=dev)))
ys = ops.alloc_split_nhwc(N, C, H, W, dev)
yf = ops.alloc_nhwc(N, C, H, W, dev)
full = os.environ.get("CONV_FULL", "0") == "1"          # residual + f32 output like the second conv of a block
fn = (lambda: ops.conv_split(xs, wp, bs, H, W, True, out_split=ys, out_f32=yf, residual=yf)) if full else \
     (lambda: ops.conv_split(xs, wp, bs, H, W, True, out_split=ys))
print(f"shape {C}->{C} @{H}x{W}, full epilogue: {full}")
for dbg in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]:
    ops.set_option("conv_debug", dbg)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"conv_debug={dbg}: {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
ops.set_option("conv_debug", 0)
