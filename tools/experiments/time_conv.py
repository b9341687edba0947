"""s_memtime probe of one workgroup of fgvc_conv_split_f32 (conv_debug = 8): cycles per stage in wait+barrier, DMA issue,
MFMA segment, chunk boundary."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, H, W = 8, 120, 214
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
for Cin, Cout, arith in [(256, 256, "bf16x3"), (256, 256, "f16f8"), (256, 256, "f16f6"), (128, 128, "f16f8"), (128, 128, "f16f6")]:
    wt = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
    bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
    fmt = ops.ACT_FMT[arith]
    wp, bs, sw = (ops.prepare_conv_split(wt, bn) + (0,)) if fmt == 0 else ops.prepare_conv_split_f16(wt, bn, fmt)
    xs = ops.nchw_to_split_nhwc(torch.relu(torch.randn(N, Cin, H, W, device=dev)))
    ys = ops.alloc_split_nhwc(N, Cout, H, W, dev)
    ops.set_option("conv_debug", 8)
    ops.conv_split(xs, wp, bs, H, W, True, out_split=ys, in_fmt=fmt, in_scale_log2=sw, out_fmt=fmt, overflow=ovf)
    torch.cuda.synchronize()
    ops.set_option("conv_debug", 0)
    v = ys.view(-1)[:256].view(torch.int64).view(8, 8).cpu()
    print(f"{Cin}->{Cout} {arith}")
    for w in range(8):
        tw, ti, tm, tb, tot, ns = v[w, :6].tolist()
        print(f"  wave {w}: per stage wait+barrier {tw / ns:6.0f}  issue {ti / ns:5.0f}  mfma {tm / ns:6.0f}  | boundaries total {tb}  | loop {tot} cycles, {ns} stages -> {tot / ns:.0f}/stage")
