"""Ablation timings of the pair top-k kernel (GPU box only; debug options give WRONG results by design)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import engine, ops
from tools.microbench import timeit

dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
one = ops.make_pairs([(1, 0)], dev)
res = {}
for kern in (3, 2, 1):
    ops.set_option("pair_kernel", kern)
    for dbg, name in [(0, "full"), (1, "no_select"), (2, "no_mfma"), (4, "no_stage"), (3, "no_select_no_mfma"),
                      (7, "nothing")]:
        if kern != 2 and dbg:
            continue
        ops.set_option("pair_debug", dbg)
        m27, _ = timeit(lambda: ops.pair_topk(feats, feats, pairs, H, W, H, W, cfg.mask, 10, validate=False), 5)
        m1, _ = timeit(lambda: ops.pair_topk(feats, feats, one, H, W, H, W, cfg.mask, 10, validate=False), 5)
        res[f"v{kern}_{name}"] = {"27pairs_ms": round(m27, 3), "1pair_ms": round(m1, 3)}
ops.set_option("pair_debug", 0)
ops.set_option("pair_kernel", 3)
print(json.dumps(res, indent=1))
