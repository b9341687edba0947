"""fgvc_merge_topk_f32 at the bench shape (27 pair lists of an 8-frame 480p clip -> 7 frames x 6 slots), alone."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.split_f16x2(ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev)))
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pl = engine.run_pairs(feats, H, W, plan, cfg)
tk = engine.merge_pairs(pl, cfg)
torch.cuda.synchronize()
ts = []
for r in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        engine.merge_pairs(pl, cfg)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20)
print(f"merge_pairs: {statistics.median(ts[1:]):.4f} ms; checksum {int(tk.idx.sum())} {float(tk.weight.sum()):.3f}")
