"""s_memtime / s_memrealtime probe of fgvc_corr_volume_f16f6 (corr6_debug = 32; + 1 without stores, + 2 without MFMAs): waves 0 and 4 of
every workgroup segment record cycles in the prologue and, per 64-key stage, in the multiply parts, the store bursts and wait +
barrier, their start and end on the chip-wide 100 MHz clock, and the XCD.  Prints averages, the in-kernel clock, and how much of the
launch's span the 256 CUs were occupied."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = (120, 214) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
HW = H * W
f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
sp = ops.split_f16f6(f)
vol = torch.empty((HW, HW), device=dev)
for dbg, name in ((32, "with stores"), (33, "no stores"), (34, "no MFMAs"), (32 + 256, "no f16 MFMAs"), (32 + 512, "no FP6 MFMAs")):
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
    ok = (rec[:, 5] > 0) & (rec[:, 5] < 1000) & (rec[:, 4] > 0) & (rec[:, 0] > 0) & (rec[:, 0] < 200000)
    w0 = rec[ok & (np.arange(n) % 2 == 0)]                    # wave 0 of each segment
    rt0 = w0[:, 6:8].copy().view(np.int64)[:, 0].astype(np.float64) / 100.0      # us
    rt1 = w0[:, 8:10].copy().view(np.int64)[:, 0].astype(np.float64) / 100.0
    good = (rt1 > rt0) & (rt1 - rt0 < 5000)
    w0, rt0, rt1 = w0[good], rt0[good], rt1[good]
    span = rt1.max() - rt0.min()
    dur = rt1 - rt0
    clk = w0[:, 4] / dur / 1e3                                # cycles / us -> GHz
    ns = w0[:, 5]
    print(f"{name}: {ms:.3f} ms by events; {len(w0)} segments recorded; span first start -> last end {span:.1f} us; "
          f"sum of segment times {dur.sum() / 256:.1f} us per CU = {dur.sum() / 256 / span:.3f} of the span; "
          f"in-kernel clock {np.median(clk):.2f} GHz (median), {clk.min():.2f}-{clk.max():.2f}")
    print(f"   per segment: prologue {w0[:, 0].mean():7.0f} cycles; per stage: multiply {(w0[:, 1] / ns).mean():6.0f} stores {(w0[:, 2] / ns).mean():6.0f} "
          f"wait+barrier {(w0[:, 3] / ns).mean():6.0f} = {((w0[:, 1] + w0[:, 2] + w0[:, 3]) / ns).mean():6.0f}; stages {ns.mean():.1f}; "
          f"segment {dur.mean():.1f} us (min {dur.min():.1f}, max {dur.max():.1f})")
    xcc = w0[:, 10] & 15
    per = "  ".join(f"{x}: {dur[xcc == x].mean():5.1f} us x {int((xcc == x).sum()):3d} seg, busy {dur[xcc == x].sum() / 32:5.1f}, last end {(rt1[xcc == x].max() - rt0.min()):5.1f}"
                    for x in sorted(set(xcc.tolist())))
    print(f"   per XCD (mean segment, segments, busy us per CU, last end): {per}")
    order = np.argsort(rt0)
    st = rt0[order] - rt0.min()
    print(f"   segment starts (us after the first): 10 % {np.percentile(st, 10):.1f}  50 % {np.percentile(st, 50):.1f}  90 % {np.percentile(st, 90):.1f}; "
          f"ends: 10 % {np.percentile(rt1 - rt0.min(), 10):.1f}  50 % {np.percentile(rt1 - rt0.min(), 50):.1f}  90 % {np.percentile(rt1 - rt0.min(), 90):.1f}  last {span:.1f}")
