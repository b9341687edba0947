"""fgvc_conv64_split_res_f32 alone at layer 1's size (240 x 427 x 64), 8 and 4 frames, in the forms the encoder launches: which of
its inputs / outputs the time follows (bytes per launch beside each line).  Round-robin over the forms, median of the rounds."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, W = 240, 427
wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
w6, b6 = ops.prepare_conv64(wt, bn)


def timeit(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for N in (8, 4):
    xs = ops.nchw_to_split_nhwc(torch.randn(N, 64, H, W, generator=g).to(dev))
    rs = ops.nchw_to_split_nhwc(torch.randn(N, 64, H, W, generator=g).to(dev))
    r_f = torch.randn(N, H, W, 64, device=dev)
    o_s, o_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
    unit = N * H * W * 64 * 4 / 1e6                              # MB of one dense 4-byte tensor
    forms = {
        "in + split out (conv1 of a block)            2 units": lambda: ops.conv64_split(xs, w6, b6, H, W, True, out_split=o_s),
        "in + f32 out                                 2 units": lambda: ops.conv64_split(xs, w6, b6, H, W, True, out_f32=o_f),
        "in + f32 residual + split out                3 units": lambda: ops.conv64_split(xs, w6, b6, H, W, True, residual=r_f, out_split=o_s),
        "in + split residual + split out              3 units": lambda: ops.conv64_split(xs, w6, b6, H, W, True, residual_split=rs, out_split=o_s),
        "in + f32 residual + split out + f32 out      4 units": lambda: ops.conv64_split(xs, w6, b6, H, W, True, residual=r_f, out_split=o_s, out_f32=o_f),
    }
    t = {k: [] for k in forms}
    for r in range(6):
        for k, fn in forms.items():
            ms = timeit(fn)
            if r:
                t[k].append(ms)
    for k, v in t.items():
        print(f"N={N} {k}: {statistics.median(v):.4f} ms  (one unit = {unit:.0f} MB)", flush=True)
