import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = engine.TrackerConfig()
H, W, T = 37, 53, 3
feats = ops.normalize_to_hwc(torch.randn(T, 256, H, W, device=dev))
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
sp3, sp6 = ops.split_f16x2(feats), ops.split_f16f6p(feats)
i3, s3 = ops.pair_topk_split(sp3, sp3, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
for dbg in (0, 16384, 0, 16384):
    ops.set_option("pair_f16_debug", dbg)
    for use_runs in (True, False):
        i6, s6 = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6", use_runs=use_runs)
        torch.cuda.synchronize()
        same = (i3 == i6).all(-1)
        ds = (s3 - s6).abs()
        bad = (ds > 2e-5).any(-1)
        # where are the bad rows (pair, y, x)?
        b = bad.nonzero()
        print(f"debug {dbg} runs {use_runs}: rows equal {same.float().mean().item()*100:.2f} %, bad-score rows {int(bad.sum())} of {bad.numel()}, max diff {ds.max().item():.2e}")
        if len(b):
            ys = (b[:, 1] // W); xs = b[:, 1] % W
            print("   pairs:", torch.bincount(b[:, 0], minlength=pairs.shape[0]).tolist(), " y%8:", torch.bincount(ys % 8, minlength=8).tolist(), " x%16:", torch.bincount(xs % 16, minlength=16).tolist())
            r0 = b[0]
            print("   first bad row", r0.tolist(), "idx3", i3[r0[0], r0[1]].tolist(), "idx6", i6[r0[0], r0[1]].tolist())
            print("   s3", [round(v, 6) for v in s3[r0[0], r0[1]].tolist()], "\n   s6", [round(v, 6) for v in s6[r0[0], r0[1]].tolist()])
ops.set_option("pair_f16_debug", 0)
print("timed out:", ops.pair_f16x3_timed_out())
