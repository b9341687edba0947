"""fgvc_corr_volume_f16f6 with the keys' h6 operands made in registers (corr6_debug 2048) against the classic form that reads them from
the LDS: difference between the two volumes (the f16 sums run over the same products in another order; the FP6 operands are the same
codes), both against the float64 product on a sample, then timed round-robin.     python tools/experiments/corr6_h6reg/time_corr6_h6reg.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
for H, W in [(37, 53), (120, 214)]:
    HW = H * W
    f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
    f[0, :7] = 0; f[0, :7, 3] = 1.0                      # one-hot rows: all-zero FP6 blocks
    sp = ops.split_f16f6(f)
    vols = {}
    for dbg in (0, 2048):
        ops.set_option("corr6_debug", dbg)
        v = torch.full((HW, HW), float("nan"), device=dev)
        ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=v)
        torch.cuda.synchronize()
        vols[dbg] = v
    ops.set_option("corr6_debug", 0)
    d = (vols[0] - vols[2048]).abs()
    rows = torch.randperm(HW, device=dev)[:512]
    ref = (f[0][rows].double() @ f[1].double().t()) / 0.07          # vol[key j][query i]: key = sp[0]?  (checked below by the smaller error)
    ref2 = (f[1][rows].double() @ f[0].double().t()) / 0.07
    e = {k: min(float((v[rows].double() - ref).abs().max()), float((v[rows].double() - ref2).abs().max())) for k, v in vols.items()}
    print(f"{H}x{W}: finite {bool(torch.isfinite(vols[2048]).all())}; max |classic - h6reg| {float(d.max()):.3e} (logit units); "
          f"max error against float64 on 512 rows: classic {e[0]:.3e}, h6reg {e[2048]:.3e}")
    del vols, d
    vol = torch.empty((HW, HW), device=dev)
    res = {0: [], 2048: []}
    for rnd in range(5):
        for dbg in res:
            ops.set_option("corr6_debug", dbg)
            for _ in range(2):
                ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
            e1.record(); torch.cuda.synchronize()
            res[dbg].append(e0.elapsed_time(e1) / 10)
    ops.set_option("corr6_debug", 0)
    for dbg, name in ((0, "h6 from the LDS (classic)"), (2048, "h6 made in registers")):
        ms = min(res[dbg])
        print(f"   {name:28s} min {ms:.4f} ms = {HW * HW * 4 / ms / 1e9:.2f} TB/s = {HW * HW * 4 / ms / 1e9 / 8:.3f} of 8 TB/s   all {[round(x, 4) for x in res[dbg]]}")
