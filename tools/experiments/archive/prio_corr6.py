"""fgvc_corr_volume_f16f6 at 480p under static wave priorities (corr6_skew bits 8 / 9 / 10) -- round 4's attempts on the north_star's
0.50 kernel; round-robin, median of rounds."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = 120, 214; HW = H * W
f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
sp = ops.split_f16f6(f)
vol = torch.empty((HW, HW), device=dev)
ref = None
modes = (("priority 0", 0), ("younger half 1", 0x100), ("older half 1", 0x200), ("younger half 3", 0x400))
res = {m: [] for m, _ in modes}
for rnd in range(7):
    for name, sk in modes:
        ops.set_option("corr6_skew", sk)
        ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
        e1.record(); torch.cuda.synchronize()
        if rnd:
            res[name].append(e0.elapsed_time(e1) / 10)
        if ref is None:
            ref = vol.clone()
        else:
            assert torch.equal(vol, ref)
ops.set_option("corr6_skew", 0)
for name, _ in modes:
    m = statistics.median(res[name])
    print(f"{name:16s} median {m:.4f} ms = {2.69044224 / m / 8:.4f} of 8 TB/s   {[round(v, 4) for v in res[name]]}")
