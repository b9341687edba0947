"""a few launches of fgvc_pair_topk_f16f6 and fgvc_pair_topk_f16x3 at the cfg2 shape, for rocprofv3 --pmc passes"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
sp6, sp3 = ops.split_f16f6p(feats), ops.split_f16x2(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
for _ in range(4):
    ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
    ops.pair_topk_split(sp3, sp3, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
torch.cuda.synchronize()
