"""fgvc_conv_split_f32 against MIOpen's f32 convolution at the encoder's layer-3 shapes (8 x 120 x 214)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
torch.backends.cudnn.benchmark = True


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


N = 8
for Cin, Cout, KS, H, W in [(256, 256, 3, 120, 214), (128, 256, 3, 120, 214), (128, 256, 1, 120, 214),
                            (128, 128, 3, 120, 214), (64, 64, 3, 240, 427)]:
    x = torch.randn(N, Cin, H, W, device=dev)
    wt = torch.randn(Cout, Cin, KS, KS, device=dev) * 0.05
    bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
    wp, bias = ops.prepare_conv_split(wt, bn)
    xs = ops.nchw_to_split_nhwc(x)
    out_s = ops.alloc_split_nhwc(N, Cout, H, W, dev)
    out_f = ops.alloc_nhwc(N, Cout, H, W, dev)
    t_m = timeit(lambda: F.conv2d(x, wt, padding=KS // 2))
    t_c = timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s))
    caps = {}
    for cap in (128, 64):
        if cap < Cout:
            ops.set_option("conv_cot_cap", cap)
            caps[cap] = round(timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s)), 3)
    ops.set_option("conv_cot_cap", 0)
    if Cout == 128 and KS == 3:
        ops.set_option("conv_narrow", 3)
        caps["4-row tiles"] = round(timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s)), 3)
        caps["4-row tiles, full epilogue"] = round(timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s, out_f32=out_f, residual=out_f)), 3)
        ops.set_option("conv_narrow", 1)
    if Cout == 64:
        ops.set_option("conv_narrow", 0)
        caps["8-row tiles"] = round(timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s)), 3)
        caps["8-row tiles, full epilogue"] = round(timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s, out_f32=out_f, residual=out_f)), 3)
        ops.set_option("conv_narrow", 1)
    t_c2 = timeit(lambda: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s, out_f32=out_f, residual=out_f))
    t_x = timeit(lambda: ops.nchw_to_split_nhwc(x, out=xs))
    fl = 2.0 * N * H * W * Cin * Cout * KS * KS
    print(f"{Cin}->{Cout} {KS}x{KS} @{H}x{W}: MIOpen f32 {t_m:.3f} ms ({fl / t_m / 1e9:.0f} TF) | conv_split {t_c:.3f} ms "
          f"({fl / t_c / 1e9:.0f} TF f32-eq, {3 * fl / t_c / 1e9:.0f} TF bf16) | cot caps {caps} | +res+f32 out {t_c2:.3f} ms | nchw->split {t_x:.3f} ms", flush=True)
