"""fgvc_conv_split_fmt_f32 in its three arithmetics (bf16x3 / f16f8 / f16x3) at the encoder's layer-2/3 shapes of an 8-frame 480p clip,
timed round-robin (the first kernel timed in a process runs slow): HIP-event ms per launch, every form with the same epilogue
(split output in its own format; "+f32" adds the dense f32 output and a residual).

    python tools/experiments/time_conv_arith.py [--rounds 5] [--reps 10]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--frames", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = a.frames
out = {}
for Cin, Cout, KS, H, W in [(256, 256, 3, 120, 214), (128, 256, 3, 120, 214), (128, 128, 3, 120, 214), (128, 256, 1, 120, 214),
                            (256, 256, 3, 128, 128), (128, 128, 3, 128, 128)]:
    wt = torch.randn(Cout, Cin, KS, KS, device=dev) * 0.05
    bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
    x = torch.randn(N, Cin, H, W, device=dev).abs()
    out_f = ops.alloc_nhwc(N, Cout, H, W, dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    forms = {}
    for name, fmt in ops.ACT_FMT.items():
        if fmt == ops.ACT_BF16X2:
            wp, bias = ops.prepare_conv_split(wt, bn)
            xs, sw = ops.nchw_to_split_nhwc(x), 0
        else:
            wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, fmt)
            xs = ops.alloc_split_nhwc(N, Cin, H, W, dev)          # operand VALUES do not matter for timing: reuse the bf16 bytes
            xs.copy_(ops.nchw_to_split_nhwc(x))
        out_s = ops.alloc_split_nhwc(N, Cout, H, W, dev)
        kw = dict(in_fmt=fmt, in_scale_log2=sw, out_fmt=fmt, out_scale_log2=0, overflow=ovf)
        forms[name] = (lambda xs=xs, wp=wp, bias=bias, out_s=out_s, kw=kw: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s, **kw),
                       lambda xs=xs, wp=wp, bias=bias, out_s=out_s, kw=kw: ops.conv_split(xs, wp, bias, H, W, True, out_split=out_s, out_f32=out_f,
                                                                                          residual=out_f, **kw))
    if KS == 3 and Cout >= 128:            # tilings of the f16f8 form: 128 output channels per workgroup, 8-row / 4-row tiles (two workgroups per CU)
        base = forms["f16f8"]

        def with_opts(fn, cap, narrow):
            def run():
                ops.set_option("conv_cot_cap", cap); ops.set_option("conv_narrow", narrow)
                fn()
                ops.set_option("conv_cot_cap", 0); ops.set_option("conv_narrow", 1)
            return run
        forms["f16f8 cot128"] = (with_opts(base[0], 128, 1), with_opts(base[1], 128, 1))
        forms["f16f8 cot128 4-row"] = (with_opts(base[0], 128, 3), with_opts(base[1], 128, 3))
        forms["f16f8 cot64 4-row"] = (with_opts(base[0], 64, 1), with_opts(base[1], 64, 1))

    times = {k: [[], []] for k in forms}
    for r in range(a.rounds + 1):
        for k, fns in forms.items():
            for j, fn in enumerate(fns):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[k][j].append(e0.elapsed_time(e1) / a.reps)
    fl = 2.0 * N * H * W * Cin * Cout * KS * KS
    key = f"{Cin}->{Cout} {KS}x{KS} @{N}x{H}x{W}"
    out[key] = {k: dict(ms=round(sorted(v[0])[len(v[0]) // 2], 4), ms_f32_res=round(sorted(v[1])[len(v[1]) // 2], 4),
                        tflops_f32_equiv=round(fl / sorted(v[0])[len(v[0]) // 2] / 1e9, 1)) for k, v in times.items()}
    print(key, json.dumps(out[key]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/r03_conv_arith.json", "w") as f:
    json.dump(out, f, indent=1)
