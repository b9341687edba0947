"""256->256 3x3 at 8x120x214 in the f16f6 arithmetic under the tiling options of fgvc_conv_split_fmt_f32 (conv_cot_cap, conv_narrow)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, C, H, W = 8, 256, 120, 214
bn = torch.nn.BatchNorm2d(C).eval().to(dev)
wt = torch.randn(C, C, 3, 3, device=dev) * 0.02
wp0, b0 = ops.prepare_conv_split(wt, bn)
wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, ops.ACT_F16F6)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
xs = ops.alloc_split_nhwc(N, C, H, W, dev)
ops.conv_split(ops.nchw_to_split_nhwc(torch.relu(torch.randn(N, C, H, W, device=dev))), wp0, b0, H, W, True, out_split=xs, out_fmt=ops.ACT_F16F6, out_scale_log2=4, overflow=ovf)
ys, yf = ops.alloc_split_nhwc(N, C, H, W, dev), ops.alloc_nhwc(N, C, H, W, dev)
kw = dict(in_fmt=ops.ACT_F16F6, in_scale_log2=4 + sw, out_fmt=ops.ACT_F16F6, out_scale_log2=4, overflow=ovf)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)


ref = None
for cap, narrow in ((0, 1), (128, 1), (128, 3), (64, 1), (64, 0), (0, 1)):
    ops.set_option("conv_cot_cap", cap); ops.set_option("conv_narrow", narrow)
    t1 = timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=ys, **kw))
    t2 = timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, residual=yf, out_split=ys, out_f32=yf, **kw))
    ops.conv_split(xs, wp, bias, H, W, True, out_split=ys, **kw)
    same = None if ref is None else bool(torch.equal(ys, ref))
    if ref is None:
        ref = ys.clone()
    print(f"cot_cap={cap:3d} narrow={narrow}: split out {t1:.3f} ms, residual + f32 + split {t2:.3f} ms; same bytes as the first form: {same}", flush=True)
ops.set_option("conv_cot_cap", 0); ops.set_option("conv_narrow", 1)
