"""s_memtime probe of one workgroup of fgvc_pair_topk_f16f6 (pair_f16_debug = 256): per consumer wave the cycles of its loop, of its
waits for key blocks, of its hand-overs (and of waiting for the selector inside them) and of its chains."""
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
buf = (ctypes.c_int64 * 32)()
for dbg, name in ((256, "registers -> LDS producers"), (256 + 32768, "LDS-DMA producers"), (256 + 2048, "no selection"), (256 + 8192, "alternating list")):
    ops.set_option("pair_f16_debug", dbg)
    for _ in range(3):
        ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
    torch.cuda.synchronize()
    _lib.call("fgvc_pair_topk_f16x3_probe", ctypes.cast(buf, ctypes.c_void_p))
    v = list(buf)
    print(name)
    for q in range(4):
        tot, wait, hand, hwait, chain, nt, slow, ns = v[8 * q:8 * q + 8]
        print(f"  consumer {q}: loop {tot:8d} cycles over {ns} blocks/pair; waiting for key blocks {wait:8d}; hand-overs {hand:7d} (waiting for the selector {hwait:7d} in {slow} of them); "
              f"chains {chain:8d} = {chain / max(nt, 1):6.0f} per tile x {nt} tiles; per tile overall {tot / max(nt, 1):6.0f}")
ops.set_option("pair_f16_debug", 0)
