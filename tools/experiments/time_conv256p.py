import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0")
F6 = ops.ACT_F16F6
N, Cin, Cout, H, W = 8, 256, 256, 120, 214
g = torch.Generator().manual_seed(1)
wt = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03).to(dev)
bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, F6)
xs = ops.alloc_split_nhwc(N, Cin, H, W, dev)
xs.copy_(ops.nchw_to_split_nhwc(torch.randn(N, Cin, H, W, generator=g).abs().to(dev)))
xs[..., 56:] = 0   # (FP6 scale bytes: zero -- values do not matter for timing, NaNs might)
o_s = ops.alloc_split_nhwc(N, Cout, H, W, dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
def call(debug):
    ops.set_option("conv_debug", debug)
    ops.conv_split(xs, wp, bias, H, W, True, out_split=o_s, in_fmt=F6, in_scale_log2=sw, out_fmt=F6, out_scale_log2=4, overflow=ovf)
def timeit(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t = {0: [], 1024: []}
for rep in range(6):
    for d in t:
        ms = timeit(lambda: call(d))
        if rep: t[d].append(ms)
ops.set_option("conv_debug", 0)
print(os.environ.get("FGVC_HIP_LIB", "default lib"), {("conv256p" if d == 0 else "conv_split"): round(statistics.median(v), 4) for d, v in t.items()})
