"""conv64p_kernel (round 5: one instruction stream per tile, the previous tile's epilogue in the matrix instructions' gaps) against
conv64_kernel<1> (option conv64_variant = 16): the same accumulation order, so every output must be BIT-identical -- split rows with
their zero borders, f32 rows, the overflow flag -- in the four forms the encoder launches, on sizes with partial tile columns, rows
that are no multiple of the tile height, fewer tiles than workgroups and several tiles per workgroup.  Then the timing at layer 1's
size (8 x 240 x 427) and the s_memtime split of workgroup 77."""
import os, sys, statistics, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops, _lib
dev = torch.device("cuda:0")
F8, BF = ops.ACT_F16F8, ops.ACT_BF16X2


def pack(x, sx):
    N, C, H, W = x.shape
    xs = ops.alloc_split_nhwc(N, 64, H, W, dev)
    v = (x.permute(0, 2, 3, 1) * 2.0 ** sx).reshape(N, H, W, 2, 32).contiguous()
    hh = v.to(torch.float16)
    l8 = ((v - hh.float()) * 2.0 ** ops.F8_BX).to(torch.float8_e4m3fn).view(torch.uint8)
    h8 = (hh.float() * 2.0 ** -ops.F8_AX).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    xs[:, 1:H + 1, 1:W + 1] = torch.cat([hh.view(torch.uint8), l8, h8], -1).contiguous().view(torch.int16)
    return xs


FORMS = {"conv1 (split f16f8 out)": dict(res=False, f32=False, fmt=F8), "conv2 (residual, f32 + split f16f8 out)": dict(res=True, f32=True, fmt=F8),
         "conv2 (residual, split f16f8 out)": dict(res=True, f32=False, fmt=F8), "conv2 (residual, split bf16 out)": dict(res=True, f32=False, fmt=BF)}


def run(variant, xs, w1, b1, sw, sx, H, W, r_f, form, relu, so):
    N = xs.shape[0]
    ops.set_option("conv64_variant", variant)
    o_s = ops.alloc_split_nhwc(N, 64, H, W, dev)
    o_f = ops.alloc_nhwc(N, 64, H, W, dev) if form["f32"] else None
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.conv64_split(xs, w1, b1, H, W, relu, residual=r_f if form["res"] else None, out_split=o_s, out_f32=o_f, in_fmt=F8, in_scale_log2=sx + sw,
                     out_fmt=form["fmt"], out_scale_log2=so, overflow=ovf)
    torch.cuda.synchronize()
    return o_s, o_f, int(ovf.item())


bad = 0
for (N, H, W) in [(1, 4, 32), (1, 3, 5), (2, 21, 50), (2, 120, 214), (3, 64, 96), (8, 240, 427)]:
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).to(dev)
    bn = torch.nn.BatchNorm2d(64).eval().to(dev)
    bn.bias.data = torch.randn(64, generator=g).to(dev) * 0.1
    w1, b1, sw = ops.prepare_conv64_f16(wt, bn)
    x = (torch.randn(N, 64, H, W, generator=g).abs() ** 1.5).to(dev)
    sx = ops.act_scale_log2(float(x.abs().max()))
    xs = pack(x, sx)
    r_f = torch.randn(N, H, W, 64, generator=g).to(dev)
    for name, form in FORMS.items():
        for relu, so in ((True, 3), (False, 3), (True, 16)):
            a = run(0, xs, w1, b1, sw, sx, H, W, r_f, form, relu, so)
            b = run(16, xs, w1, b1, sw, sx, H, W, r_f, form, relu, so)
            ok = torch.equal(a[0], b[0]) and (a[1] is None or torch.equal(a[1], b[1])) and a[2] == b[2]
            if not ok:
                bad += 1
                ds = (a[0] != b[0]).sum().item()
                df = (a[1] != b[1]).sum().item() if a[1] is not None else 0
                print(f"MISMATCH {N}x{H}x{W} {name} relu={relu} so={so}: split words {ds}, f32 words {df}, overflow {a[2]} vs {b[2]}", flush=True)
    print(f"{N} x {H} x {W}: checked", flush=True)
print("all identical" if bad == 0 else f"{bad} mismatching cases", flush=True)
if bad:
    sys.exit(1)

N, H, W = 8, 240, 427
g = torch.Generator().manual_seed(0)
wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
w1, b1, sw = ops.prepare_conv64_f16(wt, bn)
x = torch.randn(N, 64, H, W, generator=g).abs().to(dev)
sx = ops.act_scale_log2(float(x.abs().max()))
xs = pack(x, sx)
r_f = torch.randn(N, H, W, 64, device=dev)
o_s, o_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)


def call(variant, form):
    ops.set_option("conv64_variant", variant)
    ops.conv64_split(xs, w1, b1, H, W, True, residual=r_f if form["res"] else None, out_split=o_s, out_f32=o_f if form["f32"] else None, in_fmt=F8,
                     in_scale_log2=sx + sw, out_fmt=form["fmt"], out_scale_log2=3, overflow=ovf)


def timeit(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t = {(v, k): [] for v in (0, 128, 16) for k in FORMS}
for rep in range(7):
    for (v, k) in t:
        ms = timeit(lambda: call(v, FORMS[k]))
        if rep:
            t[(v, k)].append(ms)
for (v, k), ms in t.items():
    print(f"{'conv64p_kernel' if v == 0 else 'conv64p, raster' if v == 128 else 'conv64_kernel '}  {k:42s} {statistics.median(ms):.4f} ms", flush=True)
for v in (8, 8 | 128):
    for k, form in FORMS.items():
        for _ in range(3):
            call(v, form)
        torch.cuda.synchronize()
        buf = (ctypes.c_int64 * 32)()
        _lib.call("fgvc_conv64_probe", ctypes.cast(buf, ctypes.c_void_p))
        pb, pm, pw, pe, pn = list(buf)[0:5]
        if pn:
            print(f"probe ({'raster order' if v & 128 else 'XCD-aware column-major order'}) {k}: wave 0, {pn} tiles; cycles per tile: barrier {pb / pn:.0f}  tile loop body {pm / pn:.0f}  drain (once) {pe:.0f}", flush=True)
ops.set_option("conv64_variant", 0)
