"""s_memtime breakdown of conv64_kernel (workgroup 77, per wave, cycles per tile): barrier wait, multiply loop (with the interleaved
DMA issue), wait for the next patch's DMAs, epilogue issue.  100 MHz counter -> shader cycles at the clock the launch ran at are
reported as the ratio to the multiply loop's MFMA time (216 MFMAs x 8 passes x 4 cycles = 6912 cycles)."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W = 8, 240, 427
wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
w6, b6 = ops.prepare_conv64(wt, bn)
xs = ops.nchw_to_split_nhwc(torch.randn(N, 64, H, W, generator=g).to(dev))
r_f = torch.randn(N, H, W, 64, device=dev)
o_s, o_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
forms = {"in + split out": lambda: ops.conv64_split(xs, w6, b6, H, W, True, out_split=o_s),
         "in + f32 residual + split + f32 out": lambda: ops.conv64_split(xs, w6, b6, H, W, True, residual=r_f, out_split=o_s, out_f32=o_f)}
for extra in (0,):
    ops.set_option("conv64_variant", 8 | extra)
    for name, fn in forms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        buf = (ctypes.c_int64 * 32)()
        _lib.call("fgvc_conv64_probe", ctypes.cast(buf, ctypes.c_void_p))
        v = list(buf)
        print(f"variant {8 | extra} {name}: launch {a.elapsed_time(b):.4f} ms")
        for w in range(4):
            pb, pm, pw, pe, pn = v[w * 8: w * 8 + 5]
            if pn:
                tot = pb + pm + pw + pe
                print(f"   wave {w}: {pn} tiles; per tile (100 MHz ticks): barrier {pb / pn:.1f}  multiply {pm / pn:.1f}  dma wait {pw / pn:.1f}  epilogue {pe / pn:.1f}"
                      f"  | shares {pb / tot:.2f} {pm / tot:.2f} {pw / tot:.2f} {pe / tot:.2f}", flush=True)
ops.set_option("conv64_variant", 0)
