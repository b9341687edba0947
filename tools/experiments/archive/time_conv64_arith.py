"""fgvc_conv64_split_fmt_f32 at layer 1's size (8 x 240 x 427 x 64): the bf16x3 arithmetic against f16 + fp8, in the forms the
encoder launches (conv1 of a block: split out; conv2: f32 residual + split out + f32 out).  Round-robin, median."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W = 8, 240, 427
wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
w0, b0 = ops.prepare_conv64(wt, bn)
w1, b1, sw = ops.prepare_conv64_f16(wt, bn)
xs = ops.nchw_to_split_nhwc(torch.randn(N, 64, H, W, generator=g).to(dev))      # (timing only: the f16f8 kernel reads the same bytes)
r_f = torch.randn(N, H, W, 64, device=dev)
o_s, o_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
F8 = ops.ACT_F16F8
forms = {
    "bf16x3 conv1 (in + split out)": lambda: ops.conv64_split(xs, w0, b0, H, W, True, out_split=o_s),
    "f16f8  conv1 (in + split out)": lambda: ops.conv64_split(xs, w1, b1, H, W, True, out_split=o_s, in_fmt=F8, in_scale_log2=sw, out_fmt=F8, overflow=ovf),
    "bf16x3 conv2 (+ f32 residual, + f32 out)": lambda: ops.conv64_split(xs, w0, b0, H, W, True, residual=r_f, out_split=o_s, out_f32=o_f),
    "f16f8  conv2 (+ f32 residual, + f32 out)": lambda: ops.conv64_split(xs, w1, b1, H, W, True, residual=r_f, out_split=o_s, out_f32=o_f, in_fmt=F8,
                                                                        in_scale_log2=sw, out_fmt=F8, overflow=ovf),
}


def timeit(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t = {k: [] for k in forms}
for r in range(7):
    for k, fn in forms.items():
        ms = timeit(fn)
        if r:
            t[k].append(ms)
for k, v in t.items():
    print(f"{k:44s} {statistics.median(v):.4f} ms", flush=True)
