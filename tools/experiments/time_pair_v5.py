"""s_memtime probe of one workgroup of fgvc_pair_topk_f16x3 (pair_f16_debug = 256 [+ 1 no selection, + 2 no MFMA]): per consumer
wave the cycles of its loop, of its waits for key blocks and of its chains; per producer wave its loop and its waits for free slots."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
h16 = ops.split_f16x2(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
buf = (ctypes.c_int64 * 32)()
for dbg, name in ((256, "full"), (257, "no selection"), (258, "no MFMA"), (259, "neither")):
    ops.set_option("pair_f16_debug", dbg)
    for _ in range(3):
        ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
    torch.cuda.synchronize()
    _lib.call("fgvc_pair_topk_f16x3_probe", ctypes.cast(buf, ctypes.c_void_p))
    v = list(buf)
    print(name)
    for q in range(4):
        tot, wait, chain, nt = v[4 * q:4 * q + 4]
        print(f"  consumer {q}: loop {tot:7d} cycles, waiting {wait:7d}, chains {chain:7d} = {chain / max(nt, 1):6.0f} per tile x {nt} tiles")
    for q in range(4):
        tot, wait, ns = v[16 + 4 * q:16 + 4 * q + 3]
        print(f"  producer {q}: loop {tot:7d} cycles, waiting for a free slot {wait:7d}, {ns} key blocks -> {(tot - wait) / max(ns, 1):5.0f} busy cycles per block")
ops.set_option("pair_f16_debug", 0)
