"""time fgvc_pair_topk_f16f6 at the cfg2 shape (no parity check: used by ablation builds whose results are wrong) + the consumer probe"""
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
for _ in range(30):
    f6()
ts = []
for rnd in range(4):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f6()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
ops.set_option("pair_f16_debug", 256)
for _ in range(3):
    f6()
torch.cuda.synchronize()
buf = (ctypes.c_int64 * 32)()
_lib.call("fgvc_pair_topk_f16x3_probe", ctypes.cast(buf, ctypes.c_void_p))
ops.set_option("pair_f16_debug", 0)
v = list(buf)
tot, wait, hand, hwait, chain, nt, slow, ns = v[0:8]
print(f"min {min(ts):.3f} ms {[round(t, 3) for t in ts]}; probe consumer 0: chain {chain / max(nt, 1):.0f} cycles per tile, hand-over {hand / max(nt, 1):.0f}, waiting for blocks {wait / max(nt, 1):.0f}, overall {tot / max(nt, 1):.0f}; timed out {ops.pair_f16x3_timed_out()}")
