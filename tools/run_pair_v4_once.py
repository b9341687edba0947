import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
hl = ops.split_bf16(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
dbg = [int(a) for a in sys.argv[1:]] or [0]
for d in dbg:
    ops.set_option("pair_bf16_debug", d)
    for _ in range(3):
        ops.pair_topk_split(hl, hl, pairs, H, W, H, W, cfg.mask, 10, validate=False)
torch.cuda.synchronize()
