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
# the loop's cycles (s_memtime inside the assembly statement), workgroup 300
ops.set_option("conv_debug", 8)
for _ in range(2):
    ops.conv_split(xs, wp, bias, H, W, True, out_split=o_s, in_fmt=F6, in_scale_log2=sw, out_fmt=F6, out_scale_log2=4, overflow=ovf)
torch.cuda.synchronize()
ops.set_option("conv_debug", 0)
v = o_s.view(-1)[:4 * 8 * 4].view(torch.int64).cpu().view(4, 8)
for w in range(4):
    print(f"wave {w}: before the loop {int(v[w, 0])}, whole workgroup {int(v[w, 1])} cycles;", end=" ")
    print(f"loop {int(v[w, 4])} cycles, {int(v[w, 5])} stages -> {int(v[w, 4]) / max(int(v[w, 5]), 1):.0f} per stage (48 matrix instructions = 1536 cycles of pipe)")
# epilogue forms: whole-workgroup cycles of workgroup 300 minus the loop
o_f = ops.alloc_nhwc(N, Cout, H, W, dev)
res_t = torch.randn(N, H, W, Cout, device=dev)
for name, kw in (("split f16f6 out", dict(out_split=o_s, out_fmt=F6, out_scale_log2=4)), ("split bf16 out", dict(out_split=o_s, out_fmt=ops.ACT_BF16X2)),
                 ("split f16f8 out", dict(out_split=o_s, out_fmt=ops.ACT_F16F8, out_scale_log2=4)),
                 ("split f16f6 + f32 out", dict(out_split=o_s, out_f32=o_f, out_fmt=F6, out_scale_log2=4)),
                 ("residual, f16f6 + f32 out", dict(out_split=o_s, out_f32=o_f, residual=res_t, out_fmt=F6, out_scale_log2=4))):
    for dbg in (8, 8 | 1024):
        ops.set_option("conv_debug", dbg)
        for _ in range(2):
            ops.conv_split(xs, wp, bias, H, W, True, in_fmt=F6, in_scale_log2=sw, overflow=ovf, **kw)
        torch.cuda.synchronize()
        v = o_s.view(-1)[:8 * 8 * 4].view(torch.int64).cpu().view(8, 8)
        if dbg == 8:
            print(f"{name:24s} epilogue: row 0 total {int(v[0, 2])}; row 1: bias + fma + residual {int(v[0, 3])}, conversion + LDS {int(v[0, 6])}, read back + stores {int(v[0, 7])}")
            print(f"{name:24s} conv256p: before loop {int(v[0, 0])}, loop {int(v[0, 4])}, after loop {int(v[0, 1]) - int(v[0, 4]) - int(v[0, 0])} cycles")
        else:
            print(f"{name:24s} conv_split (wave 0): wait {int(v[0, 0])} issue {int(v[0, 1])} mma {int(v[0, 2])} boundaries {int(v[0, 3])} loop {int(v[0, 4])}")
ops.set_option("conv_debug", 0)
