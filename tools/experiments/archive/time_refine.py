"""The refining merge at the cfg2 shape (8 frames of 120 x 214, 27 pairs): launch time of merge_refine against the plain merge, and its
statistics; under `rocprofv3 --kernel-trace --stats` the two kernels of fgvc_merge_refine_topk_f32 separately.
    python tools/experiments/time_refine.py [eps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0")
H, W, T = 120, 214, 8
g = torch.Generator().manual_seed(480)
x = torch.nn.functional.avg_pool2d(torch.randn(T, 256, H + 2, W + 2, generator=g), 3, 1) + 0.3 * torch.randn(T, 256, H, W, generator=g)
f = ops.normalize_to_hwc(torch.relu(x).to(dev))
eps = float(sys.argv[1]) if len(sys.argv) > 1 else ops.REFINE_EPS
cfg = engine.TrackerConfig(pair_split_fmt="f16f6", pair_precision="split", pair_refine_eps=eps)
plan = engine.plan_clip(T, [0], cfg)
bank = ops.split_f16f6x(f) if cfg.bank_fmt == "f16f6x" else ops.split_f16f6p(f)
pl = engine.run_pairs(bank, H, W, plan, cfg)
plain = engine.PairLists(pl.plan, pl.idx, pl.score, pl.HW, pl.channels)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tk = engine.merge_pairs(pl, cfg)
print("stats (queries re-scored, from scratch, candidates):", tk.refine_stats.cpu().tolist(), "of", len(plan.slot_pair) * H * W, "eps", eps)
print(f"pair kernel      {timeit(lambda: engine.run_pairs(bank, H, W, plan, cfg)):.3f} ms")
print(f"plain merge      {timeit(lambda: engine.merge_pairs(plain, cfg)):.3f} ms")
print(f"refining merge   {timeit(lambda: engine.merge_pairs(pl, cfg)):.3f} ms")
p6 = ops.split_f16f6p(f)
cfg1 = engine.TrackerConfig(pair_split_fmt="f16f6", pair_precision="split", pair_refine=False)
for _ in range(2):
    print(f"pair kernel, 1 KiB rows {timeit(lambda: engine.run_pairs(p6, H, W, plan, cfg1)):.3f} ms;  2 KiB rows {timeit(lambda: engine.run_pairs(bank, H, W, plan, cfg)):.3f} ms")
