"""where do the f16 + FP6 rows of a convolution's split output differ from the oracle's model of them?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
from oracle import fgvc_oracle as O
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(3)
N, Cin, Cout, KS, H, W = 1, 64, 128, 3, 9, 40
x = torch.randn(N, Cin, H, W, generator=g).abs() ** 1.5 * (torch.rand(N, Cin, H, W, generator=g) > 0.4)
wt = torch.randn(Cout, Cin, KS, KS, generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
bn = torch.nn.BatchNorm2d(Cout).eval()
wp0, bias0 = ops.prepare_conv_split(wt.to(dev), bn.to(dev))
out_s = ops.alloc_split_nhwc(N, Cout, H, W, dev); out_f = ops.alloc_nhwc(N, Cout, H, W, dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
so = 5
ops.conv_split(ops.nchw_to_split_nhwc(x.to(dev)), wp0, bias0, H, W, True, out_split=out_s, out_f32=out_f, out_fmt=ops.ACT_F16F6, out_scale_log2=so, overflow=ovf)
f = out_f.cpu().reshape(-1, 32).numpy()
want = O.act_f16f6_rows(f, so)
got = out_s[:, 1:H + 1, 1:W + 1].contiguous().cpu().view(torch.uint8).reshape(-1, 128).numpy()
print("rows", got.shape, "bytes differing", int((got != want).sum()), "rows differing", int((got != want).any(1).sum()))
names = ("h", "h6", "l6")
for nm, a, b in zip(names, O.act_f16f6_decode(got), O.act_f16f6_decode(want)):
    bad = np.argwhere(a != b)
    print(nm, "mismatches", len(bad))
    for r, c in bad[:8]:
        xs = np.float32(f[r, c]) * np.float32(2.0 ** so)
        h = np.float16(xs); l = (xs - np.float32(h)) * np.float32(2048)
        print(f"   row {r} ch {c}: got {a[r, c]!r} want {b[r, c]!r}; s x = {xs!r} h = {float(h)!r} 2^11 l = {float(l)!r} f16 {float(np.float16(l))!r}; scale bytes got {got[r, 104]}, {got[r, 120]} want {want[r, 104]}, {want[r, 120]}; block max |h| {np.abs(np.float16(f[r] * np.float32(2.0 ** so)).astype(np.float32)).max()}")
