"""fgvc_pair_topk_f16f6 with roles switched off (pair_f16_debug: 1 = producers move nothing, 2 = no hand-over and no selectors, 2048 = the selector
only receives, 8192 = alternating block list), round-robin timed at the cfg2 shape; results of the ablated builds are wrong."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
sp6 = ops.split_f16f6p(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
f6 = lambda: ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")


def ms(reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f6()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


names = {0: "default", 1: "producers move nothing", 2: "no hand-over, no selectors", 3: "consumers alone (1 + 2)", 2048: "selector only receives",
         8192: "alternating list", 8192 + 3: "consumers alone, alternating list"}
for _ in range(50):
    f6()
res = {k: [] for k in names}
for rnd in range(4):
    for dbg in names:
        ops.set_option("pair_f16_debug", dbg)
        ms(2)
        res[dbg].append(ms())
ops.set_option("pair_f16_debug", 0)
for dbg, name in names.items():
    print(f"{name:36s} min {min(res[dbg]):.3f} ms  all {[round(x, 3) for x in res[dbg]]}")
print("timed out:", ops.pair_f16x3_timed_out())
