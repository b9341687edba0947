"""256 -> 256 and 128 -> 128 3 x 3 in f16f6 under static wave priorities (conv_debug 32: younger half 1, 64: older half 1, 128: younger half 3) and
with every weight DMA of a stage issued by the older wave of each SIMD (256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, H, W = 8, 120, 214
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)
for C in (256, 128):
    bn = torch.nn.BatchNorm2d(C).eval().to(dev)
    wt = torch.randn(C, C, 3, 3, device=dev) * 0.02
    wp0, b0 = ops.prepare_conv_split(wt, bn)
    wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, ops.ACT_F16F6)
    xs = ops.alloc_split_nhwc(N, C, H, W, dev)
    ops.conv_split(ops.nchw_to_split_nhwc(torch.relu(torch.randn(N, C, H, W, device=dev))), wp0, b0, H, W, True, out_split=xs, out_fmt=ops.ACT_F16F6, out_scale_log2=4, overflow=ovf)
    ys = ops.alloc_split_nhwc(N, C, H, W, dev)
    kw = dict(in_fmt=ops.ACT_F16F6, in_scale_log2=4 + sw, out_fmt=ops.ACT_F16F6, out_scale_log2=4, overflow=ovf)
    ref = None
    for name, dbg in (("default", 0), ("DMAs by older waves", 256), ("younger half 1", 32), ("DMAs older + younger 1", 256 + 32), ("DMAs older + older 1", 256 + 64), ("default", 0)):
        ops.set_option("conv_debug", dbg)
        t = timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=ys, **kw))
        ops.conv_split(xs, wp, bias, H, W, True, out_split=ys, **kw)
        ops.set_option("conv_debug", 0)
        same = None if ref is None else bool(torch.equal(ys, ref))
        if ref is None:
            ref = ys.clone()
        print(f"{C} -> {C}: {name:24s} {t:.4f} ms  same bytes: {same}", flush=True)
