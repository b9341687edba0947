"""s_memtime probe of fgvc_corr_volume_f16f6 (corr6_debug = 32; + 1 without stores, + 2 without MFMAs): waves 0 and 4 of every
workgroup record cycles in the prologue and, per 64-key stage, in the multiply parts, the store bursts and wait + barrier, their
start stamp and the XCD.  Prints per dispatch round (start-time order, 256 workgroups each) the averages and the in-kernel clock."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = (120, 214) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
HW = H * W
f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
sp = ops.split_f16f6(f)
vol = torch.empty((HW, HW), device=dev)
for dbg, name in ((32, "with stores"), (33, "no stores"), (34, "no MFMAs")):
    for _ in range(3):
        ops.set_option("corr6_debug", dbg)
        vol[0].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ops.set_option("corr6_debug", 0)
    raw = vol[0].view(torch.int32).cpu().numpy()
    n = len(raw) // 12
    rec = raw[:n * 12].reshape(n, 12)
    rec = rec[rec[:, 5] > 0]
    start = rec[:, 6:8].copy().view(np.int64)[:, 0]
    real = rec[:, 8:10].copy().view(np.int64)[:, 0]
    t0 = start.min()
    end = start + rec[:, 4]
    span = end.max() - t0
    # s_memrealtime ticks at 100 MHz: clock = d(memtime) / d(realtime) * 100 MHz over the whole launch
    clk = span / max(1, (real.max() - (real - rec[:, 4] * 0).min())) * 0.1
    print(f"{name}: {ms:.3f} ms (event), {len(rec)} wave records, launch span {span} cycles -> {span / ms / 1e6:.2f} GHz by the event time")
    order = np.argsort(start)
    rec, start = rec[order], start[order]
    per = 2 * 256
    for r0 in range(0, len(rec), per):
        r = rec[r0:r0 + per]
        ns = r[:, 5]
        print(f"  workgroups started {r0 // 2:4d}..: start +{(start[r0:r0 + per] - t0).mean():9.0f}  prologue {r[:, 0].mean():7.0f} (min {r[:, 0].min()}, max {r[:, 0].max()})"
              f" | per stage: multiply {(r[:, 1] / ns).mean():6.0f} stores {(r[:, 2] / ns).mean():6.0f} wait+barrier {(r[:, 3] / ns).mean():6.0f}"
              f" | total {r[:, 4].mean():8.0f}, stages {ns.mean():.1f}")
    w0, w4 = rec[(rec[:, 11] >= 0)][0::1], None
