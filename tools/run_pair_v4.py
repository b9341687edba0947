"""fgvc_pair_topk_bf16x4 against fgvc_pair_topk_f32: agreement on ragged sizes and timing at the bench size."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import engine, ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def compare(H, W, nr, mode="circle", topk=10, T=3):
    f = ops.normalize_to_hwc(torch.randn(T, 256, H, W, device=dev))
    hl = ops.split_bf16(f)
    mask = ops.MaskSpec.from_neighbor_range(nr, mode)
    pairs = ops.make_pairs([(2, 0, nr is not None), (2, 1, nr is not None), (1, 0, nr is not None)][:T], dev)
    i3, s3 = ops.pair_topk(f, f, pairs, H, W, H, W, mask, topk)
    i4, s4 = ops.pair_topk_split(hl, hl, pairs, H, W, H, W, mask, topk)
    torch.cuda.synchronize()
    same = (i3 == i4).all(-1).float().mean().item()
    ds = (s3 - s4)[torch.isfinite(s3) & torch.isfinite(s4)].abs().max().item() if s3.numel() else 0.0
    infeq = bool((torch.isfinite(s3) == torch.isfinite(s4)).all())
    print(f"H={H} W={W} nr={nr} {mode} k={topk}: rows identical {same:.5f}  max|ds| {ds:.2e}  inf-pattern-equal {infeq}", flush=True)
    return same, ds


if len(sys.argv) > 1 and sys.argv[1] == "check":
    for args in [(37, 53, 30), (8, 8, 30), (5, 3, 4), (33, 70, 30), (20, 20, None), (17, 23, 9, "square"), (64, 64, 12, "circle", 5),
                 (120, 214, 30)]:
        compare(*args)

H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
hl = ops.split_bf16(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"f32 v3: {timeit(lambda: ops.pair_topk(feats, feats, pairs, H, W, H, W, cfg.mask, 10, validate=False)):.3f} ms / 27 pairs")
fn = lambda: ops.pair_topk_split(hl, hl, pairs, H, W, H, W, cfg.mask, 10, validate=False)
for prod in (3, 4):
    ops.set_option("pair_bf16_products", prod)
    for dbg in [int(a) for a in sys.argv[2:]] or [0]:
        ops.set_option("pair_bf16_debug", dbg)
        print(f"split-bf16 products={prod} debug={dbg}: {timeit(fn):.3f} ms / 27 pairs", flush=True)
ops.set_option("pair_bf16_debug", 0)
ops.set_option("pair_bf16_products", 4)
