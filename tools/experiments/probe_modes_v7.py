"""consumer probe of fgvc_pair_topk_f16f6 under ablation bits (results wrong): 2048 = the selector only receives, 32768 = LDS-DMA producers"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
sp6 = ops.split_f16f6p(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
f6 = lambda: ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
buf = (ctypes.c_int64 * 32)()
for extra, name in ((0, "default"), (1, "producers move nothing"), (2, "no hand-over, no selectors"), (3, "consumers alone"), (2048, "selector only receives"), (8192, "alternating list")):
    ops.set_option("pair_f16_debug", 256 + extra)
    for _ in range(3):
        f6()
    torch.cuda.synchronize()
    _lib.call("fgvc_pair_topk_f16x3_probe", ctypes.cast(buf, ctypes.c_void_p))
    v = list(buf)
    tot, wait, hand, hwait, chain, nt, slow, ns = v[0:8]
    print(f"{name:26s}: chain {chain / max(nt, 1):6.0f}  hand-over {hand / max(nt, 1):5.0f}  waiting for blocks {wait / max(nt, 1):5.0f}  overall {tot / max(nt, 1):6.0f} cycles per tile")
ops.set_option("pair_f16_debug", 0)
