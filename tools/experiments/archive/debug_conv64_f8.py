import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from fgvc_amd import ops
from test_gpu_parity import _pack_act, _nhwc_to_nchw
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
N, H, W = 1, 4, 32
fmt = ops.ACT_F16F8
for mode in ("ones", "rand"):
    x = torch.ones(N, 64, H, W) if mode == "ones" else torch.rand(N, 64, H, W, generator=g)
    wt = torch.zeros(64, 64, 3, 3)
    if mode == "ones":
        wt[:, :, 1, 1] = 1.0 / 64          # centre tap: y = mean over channels = 1
    else:
        wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    bn = torch.nn.BatchNorm2d(64).eval()
    sx = ops.act_scale_log2(float(x.abs().max()))
    wp, bias, sw = ops.prepare_conv64_f16(wt.to(dev), bn.to(dev))
    wg, bg, swg = ops.prepare_conv_split_f16(wt.to(dev), bn.to(dev), fmt)
    xs = _pack_act(x, fmt, sx, dev)
    a, b = ops.alloc_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
    ops.conv64_split(xs, wp, bias, H, W, False, out_f32=a, in_fmt=fmt, in_scale_log2=sx + sw)
    ops.conv_split(xs, wg, bg, H, W, False, out_f32=b, in_fmt=fmt, in_scale_log2=sx + swg)
    torch.cuda.synchronize()
    print(mode, "sx", sx, "sw", sw, "conv64 f16f8:", a[0, 1, 5, :6].tolist(), "generic:", b[0, 1, 5, :6].tolist())
    print("   max |a|", float(a.abs().max()), "max |b|", float(b.abs().max()), "max diff", float((a - b).abs().max()))
    d = (a - b).abs()[0]
    bad = (d > 1e-3).nonzero()
    print("   bad count", bad.shape[0], "of", d.numel(), "rows", sorted(set(bad[:, 0].tolist())), "cols", sorted(set(bad[:, 1].tolist()))[:40], "chans", sorted(set(bad[:, 2].tolist()))[:70])
