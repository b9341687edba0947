"""fgvc_pair_topk_f16f6 at the cfg2 shape under three work orders: runs of pairs (default), one pair per workgroup, and one pair per
workgroup with every XCD owning a contiguous range of the (pair, tile) sequence (debug bit 512): time + equality of the lists.
With `pmc` as the first argument: a few launches of each for rocprofv3 --pmc passes (no timing)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
args = [a for a in sys.argv[1:] if a != "pmc"]
if len(args) == 3:
    H, W, T = (int(a) for a in args)
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
sp6 = ops.split_f16f6p(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
print(f"{T} frames of {H} x {W}: {len(plan.pairs)} pairs", flush=True)
pairs = ops.make_pairs(plan.pairs, dev)
def run(use_runs, dbg):
    ops.set_option("pair_f16_debug", dbg)
    r = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6", use_runs=use_runs)
    ops.set_option("pair_f16_debug", 0)
    return r
modes = [("runs, XCD ranges + strips, aligned walks (default)", True, 0), ("runs, XCD ranges + strips", True, 1024), ("runs, aligned walks", True, 512), ("runs (first build)", True, 512 + 1024), ("pair per workgroup, XCD ranges + strips, aligned walks", False, 0), ("pair per workgroup, XCD ranges + strips", False, 1024)]
if "pmc" in sys.argv[1:]:
    for _ in range(3):
        for name, ur, dbg in modes:
            run(ur, dbg)
    torch.cuda.synchronize(); sys.exit(0)
ref = run(True, 0)
for name, ur, dbg in modes:
    for _ in range(10):
        out = run(ur, dbg)
    same = bool((out[0] == ref[0]).all()) and bool((out[1] == ref[1]).all())
    ts = []
    for rnd in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(ur, dbg)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print(f"{name:58s} min {min(ts):.3f} ms {[round(t, 3) for t in ts]} lists equal to the default's: {same}; timed out {ops.pair_f16x3_timed_out()}", flush=True)
