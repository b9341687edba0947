"""fgvc_conv64_split_fmt_f32 (f16 + fp8) at layer 1's size, 8 x 240 x 427 x 64: round 5's kernel with K split across the waves
(conv64k_kernel) against round 3's (option conv64_variant = 16), in the two forms the encoder launches; the two kernels' results
against each other (another accumulation order: last bits); the s_memtime breakdown of workgroup 77 (variant 8).  Round-robin, median."""
import os, sys, statistics, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W = 8, 240, 427
F8 = ops.ACT_F16F8
wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
w1, b1, sw = ops.prepare_conv64_f16(wt, bn)
x = torch.randn(N, 64, H, W, generator=g).abs().to(dev)
sx = ops.act_scale_log2(float(x.abs().max()))
xs = ops.alloc_split_nhwc(N, 64, H, W, dev)
v = (x.permute(0, 2, 3, 1) * 2.0 ** sx).reshape(N, H, W, 2, 32).contiguous()       # [h = f16(s x) | l8 = e4m3(2^BX l) | h8 = e4m3(h / 2^AX)]
hh = v.to(torch.float16)
l8 = ((v - hh.float()) * 2.0 ** ops.F8_BX).to(torch.float8_e4m3fn).view(torch.uint8)
h8 = (hh.float() * 2.0 ** -ops.F8_AX).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
xs[:, 1:H + 1, 1:W + 1] = torch.cat([hh.view(torch.uint8), l8, h8], -1).contiguous().view(torch.int16)
del v, hh, l8, h8
r_f = torch.randn(N, H, W, 64, device=dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)


def run(variant, with_res, o_s, o_f):
    ops.set_option("conv64_variant", variant)
    ops.conv64_split(xs, w1, b1, H, W, True, residual=r_f if with_res else None, out_split=o_s, out_f32=o_f if with_res else None, in_fmt=F8,
                     in_scale_log2=sx + sw, out_fmt=F8, out_scale_log2=3, overflow=ovf)


outs = {}
for v in (0, 16):
    o_s, o_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
    run(v, True, o_s, o_f)
    torch.cuda.synchronize()
    outs[v] = (o_s, o_f)
d = (outs[0][1] - outs[16][1]).abs().max().item()
print(f"new vs old kernel: max |f32 out difference| {d:.3e} of max |y| {outs[16][1].abs().max().item():.3e}; split outputs equal on "
      f"{(outs[0][0] == outs[16][0]).float().mean().item():.6f} of the words", flush=True)
o_s, o_f = outs[0]


def timeit(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


forms = {(v, r): (lambda v=v, r=r: run(v, r, o_s, o_f)) for v in (0, 16) for r in (False, True)}
t = {k: [] for k in forms}
for rep in range(7):
    for k, fn in forms.items():
        ms = timeit(fn)
        if rep:
            t[k].append(ms)
for (v, r), ms in t.items():
    print(f"{'K-split kernel' if v == 0 else 'round-3 kernel'}  {'conv2 (+ f32 residual, + f32 out)' if r else 'conv1 (in + split out)          '} {statistics.median(ms):.4f} ms", flush=True)
for v in (8, 24, 8 | 32, 8 | 64, 8 | 96):
    for r in (False, True):
        for _ in range(3):
            run(v, r, o_s, o_f)
        torch.cuda.synchronize()
        buf = (ctypes.c_int64 * 32)()
        _lib.call("fgvc_conv64_probe", ctypes.cast(buf, ctypes.c_void_p))
        vals = list(buf)
        pb, pm, pw, pe, pn = vals[0:5]
        if pn:
            print(f"variant {v} res={r}: wave 0, {pn} tiles; per tile (memtime ticks): barrier {pb / pn:.0f}  multiply {pm / pn:.0f}  "
                  f"{'exchange write + dma wait' if v == 8 else 'dma wait'} {pw / pn:.0f}  epilogue {pe / pn:.0f}", flush=True)
ops.set_option("conv64_variant", 0)
