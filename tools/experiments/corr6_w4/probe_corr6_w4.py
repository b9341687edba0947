import os, sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = 120, 214
HW = H * W
f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
sp = ops.split_f16f6(f)
vol = torch.empty((HW, HW), device=dev)
for base, nm in ((0, "W8"), (2048, "W4")):
    for dbg, name in ((32, "with stores"), (33, "no stores"), (34, "no MFMAs")):
        for _ in range(3):
            ops.set_option("corr6_debug", base + dbg)
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
        w0 = rec[ok & (np.arange(n) % 2 == 0)]
        rt0 = w0[:, 6:8].copy().view(np.int64)[:, 0].astype(np.float64) / 100.0
        rt1 = w0[:, 8:10].copy().view(np.int64)[:, 0].astype(np.float64) / 100.0
        good = (rt1 > rt0) & (rt1 - rt0 < 5000)
        w0, rt0, rt1 = w0[good], rt0[good], rt1[good]
        dur = rt1 - rt0
        clk = w0[:, 4] / dur / 1e3
        ns = w0[:, 5]
        print(f"{nm} {name}: {ms:.3f} ms; clock {np.median(clk):.2f} GHz; prologue {w0[:, 0].mean():7.0f}; per stage: multiply {(w0[:, 1] / ns).mean():6.0f} stores {(w0[:, 2] / ns).mean():6.0f} "
              f"wait+barrier {(w0[:, 3] / ns).mean():6.0f} = {((w0[:, 1] + w0[:, 2] + w0[:, 3]) / ns).mean():6.0f}; busy {dur.sum() / 256 / (rt1.max() - rt0.min()):.3f} of span")
