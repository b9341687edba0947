"""s_memtime stamps of conv256p_kernel's bank-writing epilogue (option conv_debug 8; workgroup 300, pixel row 1 of wave 0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0")
F6 = ops.ACT_F16F6
N, Cin, H, W = 4, 256, 120, 214
g = torch.Generator().manual_seed(1)
wt = (torch.randn(256, Cin, 3, 3, generator=g) * 0.03).to(dev)
bn = torch.nn.BatchNorm2d(256).eval().to(dev)
wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, F6)
xs = ops.alloc_split_nhwc(N, Cin, H, W, dev)
xs.copy_(ops.nchw_to_split_nhwc(torch.randn(N, Cin, H, W, generator=g).abs().to(dev)))
xs[..., 56:] = 0
res = torch.randn(N, H, W, 256, device=dev)
bank = torch.zeros(N, H * W, 4, 256, dtype=torch.int16, device=dev)
for dbg in (0, 8):
    ops.set_option("conv_debug", dbg)
    for _ in range(3):
        ops.conv_split_to_bank(xs, wp, bias, H, W, True, bank, residual=res, in_fmt=F6, in_scale_log2=4 + sw)
    torch.cuda.synchronize()
    if dbg == 0:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv_split_to_bank(xs, wp, bias, H, W, True, bank, residual=res, in_fmt=F6, in_scale_log2=4 + sw)
        b.record(); torch.cuda.synchronize()
        print(f"bank-writing launch, 4 frames: {a.elapsed_time(b) / 10:.4f} ms")
ops.set_option("conv_debug", 0)
v = bank.view(-1)[:4 * 8 * 4].view(torch.int64).cpu().view(4, 8)
print("pixel row 1 of wave 0 (cycles): residual + fma", int(v[0, 0]), "| sum of squares, partner, 3 barriers", int(v[0, 1]), "| normalise + FP6 rows", int(v[0, 2]),
      "| barrier + stores (first KiB)", int(v[0, 3]), "| second KiB (f32 row) + stores", int(v[0, 4]), "| whole workgroup", int(v[0, 5]))
