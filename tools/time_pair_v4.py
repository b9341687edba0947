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
for d in [256]:
    ops.set_option("pair_bf16_debug", d)
    idx, sc = ops.pair_topk_split(hl, hl, pairs, H, W, H, W, cfg.mask, 10, validate=False)
    torch.cuda.synchronize()
    v = idx.view(-1)[:128].view(torch.int64).view(8, 8).cpu()
    print("debug", d)
    for w in range(8):
        ta, tb, tbar, tl, na, nb, ns = v[w, :7].tolist()
        print(f"  wave {w}: A {ta / max(na,1):7.0f} x{na}  B {tb / max(nb,1):7.0f} x{nb}  barrier-wait {tbar / max(ns,1):7.0f}/step  loop {tl} cycles, {ns} steps -> {tl / max(ns,1):.0f}/step")
ops.set_option("pair_bf16_debug", 0)
