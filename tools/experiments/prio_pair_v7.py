"""fgvc_pair_topk_f16f6 under static wave priorities (s_setprio once per wave): selector / consumer / producer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
if len(sys.argv) == 4:
    H, W, T = (int(a) for a in sys.argv[1:])
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
sp6 = ops.split_f16f6p(feats)
cfg = engine.TrackerConfig(); plan = engine.plan_clip(T, [0], cfg); pairs = ops.make_pairs(plan.pairs, dev)
def run(dbg):
    ops.set_option("pair_f16_debug", dbg)
    r = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
    ops.set_option("pair_f16_debug", 0)
    return r
ref = run(0)
for name, dbg in (("selector 1 (default)", 0), ("all priority 0", 16384), ("selector 1 + consumer 1", 65536), ("selector 1 + producer 1", 131072), ("selector 1 (default)", 0)):
    for _ in range(10):
        out = run(dbg)
    same = bool((out[0] == ref[0]).all()) and bool((out[1] == ref[1]).all())
    ts = []
    for rnd in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(dbg)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print(f"{name:28s} min {min(ts):.3f} ms {[round(t, 3) for t in ts]} same lists: {same}; timed out {ops.pair_f16x3_timed_out()}", flush=True)
