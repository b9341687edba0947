"""256->256 3x3 at 8x120x214 under the tiling options of fgvc_conv_split_f32 (conv_cot_cap, conv_narrow)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, C, H, W = 8, 256, 120, 214
wp, bias = ops.prepare_conv_split(torch.randn(C, C, 3, 3, device=dev) * 0.02, torch.nn.BatchNorm2d(C).eval().to(dev))
xs = ops.nchw_to_split_nhwc(torch.randn(N, C, H, W, device=dev))
ys, yf = ops.alloc_split_nhwc(N, C, H, W, dev), ops.alloc_nhwc(N, C, H, W, dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for cap, narrow in ((0, 1), (128, 1), (128, 3), (64, 1), (64, 0)):
    ops.set_option("conv_cot_cap", cap); ops.set_option("conv_narrow", narrow)
    t1 = timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=ys))
    t2 = timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, residual=yf, out_split=ys, out_f32=yf))
    print(f"cot_cap={cap:3d} narrow={narrow}: split out {t1:.3f} ms, residual + f32 + split {t2:.3f} ms", flush=True)
ops.set_option("conv_cot_cap", 0); ops.set_option("conv_narrow", 1)
