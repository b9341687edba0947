"""A/B of fgvc_conv64_split_f32's schedule bits (option "conv64_variant") at layer 1's size, 8 frames: results must be bit-identical
to variant 0; round-robin timing, median."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W = 8, 240, 427
wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
w6, b6 = ops.prepare_conv64(wt, bn)
xs = ops.nchw_to_split_nhwc(torch.randn(N, 64, H, W, generator=g).to(dev))
r_f = torch.randn(N, H, W, 64, device=dev)
VARIANTS = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 3, 7]


def timeit(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ref = {}
outs = {}
for v in VARIANTS:
    ops.set_option("conv64_variant", v)
    o_s, o_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
    ops.conv64_split(xs, w6, b6, H, W, True, residual=r_f, out_split=o_s, out_f32=o_f)
    o2 = ops.alloc_split_nhwc(N, 64, H, W, dev)
    ops.conv64_split(xs, w6, b6, H, W, True, out_split=o2)
    torch.cuda.synchronize()
    if not ref:
        ref = dict(s=o_s, f=o_f, s2=o2)
    print(f"variant {v}: identical to variant {VARIANTS[0]}: {torch.equal(o_s, ref['s']) and torch.equal(o_f, ref['f']) and torch.equal(o2, ref['s2'])}", flush=True)
o_s, o_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
forms = {"in + split out": lambda: ops.conv64_split(xs, w6, b6, H, W, True, out_split=o_s),
         "in + f32 residual + split out": lambda: ops.conv64_split(xs, w6, b6, H, W, True, residual=r_f, out_split=o_s),
         "in + f32 residual + split + f32 out": lambda: ops.conv64_split(xs, w6, b6, H, W, True, residual=r_f, out_split=o_s, out_f32=o_f)}
t = {(v, k): [] for v in VARIANTS for k in forms}
for r in range(6):
    for v in VARIANTS:
        ops.set_option("conv64_variant", v)
        for k, fn in forms.items():
            ms = timeit(fn)
            if r:
                t[(v, k)].append(ms)
ops.set_option("conv64_variant", 0)
for (v, k), ms in t.items():
    print(f"variant {v} {k:38s} {statistics.median(ms):.4f} ms", flush=True)
