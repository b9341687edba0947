import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
for k in (3, 2):
    ops.set_option("pair_kernel", k)
    for _ in range(3):
        ops.pair_topk(feats, feats, pairs, H, W, H, W, cfg.mask, 10, validate=False)
torch.cuda.synchronize()
