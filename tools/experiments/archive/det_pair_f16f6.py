import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = engine.TrackerConfig()
for (H, W, T) in ((128, 128, 12), (37, 53, 9)):
    x = torch.relu(torch.nn.functional.avg_pool2d(torch.randn(T, 256, H + 4, W + 4, device=dev), 5, 1) + 0.1 * torch.randn(T, 256, H, W, device=dev))
    feats = ops.normalize_to_hwc(x)
    sp6 = ops.split_f16f6p(feats)
    plan = engine.plan_clip(T, [0], cfg)
    pairs = ops.make_pairs(plan.pairs, dev)
    ref = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6", use_runs=True)
    for trial in range(3):
        a = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6", use_runs=True)
        b = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6", use_runs=False)
        # a subset of the pairs (every second one)
        sel = torch.arange(0, pairs.shape[0], 2, device=dev)
        c = ops.pair_topk_split(sp6, sp6, pairs[sel].contiguous(), H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6", use_runs=True)
        torch.cuda.synchronize()
        print(f"{H}x{W}x{T} trial {trial}: repeat equal {torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1])}; runs vs single pairs equal "
              f"{torch.equal(b[0], ref[0]) and torch.equal(b[1], ref[1])} (rows differing {int((b[0] != ref[0]).any(-1).sum())}, max score diff {(b[1]-ref[1]).abs().max().item():.2e}); "
              f"subset equal {torch.equal(c[0], ref[0][sel]) and torch.equal(c[1], ref[1][sel])}")
print("timed out", ops.pair_f16x3_timed_out())
