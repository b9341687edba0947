"""fgvc_corr_volume_f16f6: the four-wave form (corr6_debug 2048: one wave per SIMD, 64 queries each) against the eight-wave form at the
480p shape (and a ragged one): volumes bit-identical (the same products in the same order per output element), then timed round-robin.
    python tools/experiments/time_corr6_w4.py [HxW]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
shapes = [(37, 53), (120, 214)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1].split("x"))]
for H, W in shapes:
    HW = H * W
    f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
    sp = ops.split_f16f6(f)
    vols = {}
    for dbg in (0, 2048):
        ops.set_option("corr6_debug", dbg)
        v = torch.full((HW, HW), float("nan"), device=dev)
        ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=v)
        torch.cuda.synchronize()
        vols[dbg] = v
    ops.set_option("corr6_debug", 0)
    same = torch.equal(vols[0], vols[2048])
    print(f"{H}x{W}: bit-identical {same}; finite {bool(torch.isfinite(vols[2048]).all())}")
    if not same:
        d = (vols[0] != vols[2048]) | torch.isnan(vols[2048])
        print("   differing elements:", int(d.sum()), "first:", d.nonzero()[:5].tolist())
    del vols
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
    for dbg, name in ((0, "eight waves x 32 queries"), (2048, "four waves x 64 queries")):
        ms = min(res[dbg])
        print(f"   {name:28s} min {ms:.4f} ms = {HW * HW * 4 / ms / 1e9:.0f} GB/s = {HW * HW * 4 / ms / 1e9 / 8000:.3f} of 8 TB/s   all {[round(x, 4) for x in res[dbg]]}")
