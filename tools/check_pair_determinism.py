"""fgvc_pair_topk_f16x3: the three-role form against the two-role form (pair_f16_debug = 1024) and against itself, many launches on
fixed inputs, bit for bit; several grid shapes (edge tiles, small frames)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0")
bad = 0
for (H, W, T, seed) in [(120, 214, 8, 0), (16, 24, 5, 1), (33, 70, 6, 2), (9, 13, 4, 3)]:
    torch.manual_seed(seed)
    feats = ops.normalize_to_hwc(torch.randn(T, 256, H, W, device=dev))
    h16 = ops.split_f16x2(feats)
    cfg = engine.TrackerConfig(neighbor_range=30 if H > 20 else 12)
    plan = engine.plan_clip(T, [0], cfg)
    pairs = ops.make_pairs(plan.pairs, dev)
    run = lambda: ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
    ops.set_option("pair_f16_debug", 1024)
    i0, s0 = run()
    ops.set_option("pair_f16_debug", 0)
    n_bad = 0
    for it in range(60 if H < 100 else 25):
        i1, s1 = run()
        if not (torch.equal(i0, i1) and torch.equal(s0, s1)):
            n_bad += 1
            d = (i0 != i1).any(-1)
            print(f"  {H}x{W} launch {it}: {int(d.sum())} rows differ, first at {d.nonzero()[0].tolist()}, max |score diff| {float((s0 - s1).abs().nan_to_num(0, 0, 0).max()):.3e}", flush=True)
    print(f"{H}x{W} T={T}: {n_bad} launches differ from the two-role form; timed out: {ops.pair_f16x3_timed_out()}", flush=True)
    bad += n_bad
sys.exit(1 if bad else 0)
