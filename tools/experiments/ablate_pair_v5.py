"""fgvc_pair_topk_f16x3 with parts switched off (pair_f16_debug: 1 = no selection, 2 = no MFMA), round-robin timed: what a kernel
whose consumers only multiply could reach."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
h16 = ops.split_f16x2(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)


def ms(reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


names = {0: "three roles (default)", 2048: "three roles, hand-over only (no selection)", 1024: "two roles", 1025: "two roles, no selection", 1026: "two roles, no MFMA", 1027: "two roles, neither",
         1024 + 16: "prologue only", 1024 + 64: "prologue + epilogue"}
res = {k: [] for k in names}
for _ in range(200):
    ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
for rnd in range(4):
    for dbg in names:
        ops.set_option("pair_f16_debug", dbg)
        ms(2)
        res[dbg].append(ms())
ops.set_option("pair_f16_debug", 0)
for dbg, name in names.items():
    print(f"{name:46s} {min(res[dbg]):.3f} ms")
