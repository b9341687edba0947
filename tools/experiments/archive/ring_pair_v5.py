"""What bounds the key-block ring of fgvc_pair_topk_f16x3 on its own (two-role form, no MFMA, no selection: pair_f16_debug = 1027)?
+ 16384: half of every row (bytes halved, DMA instructions unchanged); + 32768: every workgroup reads the same four key blocks (L2-hot)."""
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


def ms(reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


names = {1027: "ring alone", 1027 + 16384: "ring alone, half rows", 1027 + 32768: "ring alone, L2-hot blocks", 1027 + 16384 + 32768: "ring alone, half rows, L2-hot",
         1027 + 8: "ring alone, row-major list"}
res = {k: [] for k in names}
for _ in range(100):
    ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
for rnd in range(4):
    for dbg in names:
        ops.set_option("pair_f16_debug", dbg)
        ms(2)
        res[dbg].append(ms())
ops.set_option("pair_f16_debug", 0)
for dbg, name in names.items():
    print(f"{name:36s} min {min(res[dbg]):.3f} ms  all {[round(x, 3) for x in res[dbg]]}")
