"""fgvc_pair_topk_f16f6 with the runs of pairs cut into sub-runs of at most m pairs (shorter workgroup lives: walks stay aligned longer;
more query prologues): launch time, equality with the default's lists; with `pmc` as the first argument a few launches of each for counter passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
args = [a for a in sys.argv[1:] if a != "pmc"]
H, W, T = (120, 214, 8) if len(args) < 3 else (int(a) for a in args[:3])
feats = ops.normalize_to_hwc(torch.randn(T, 256, H, W, device=dev))
sp6 = ops.split_f16f6p(feats)
cfg = engine.TrackerConfig(); plan = engine.plan_clip(T, [0], cfg)
def pairs_with(m):
    pr = ops.make_pairs(plan.pairs, dev)
    runs = ops.pair_runs(pr).cpu().tolist()
    if m:
        cut = []
        for first, cnt in runs:
            i = 0
            while i < cnt:
                c = min(m, cnt - i)
                cut.append((first + i, c)); i += c
        cut.sort(key=lambda r: -r[1])
        pr._fgvc_runs = torch.tensor(cut, dtype=torch.int32, device=dev).reshape(-1, 2)
    return pr
modes = [("whole runs (default)", 0), ("sub-runs of <= 3", 3), ("sub-runs of <= 2", 2), ("one pair", 1)]
prs = {name: pairs_with(m) for name, m in modes}
run = lambda name: ops.pair_topk_split(sp6, sp6, prs[name], H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
if "pmc" in sys.argv[1:]:
    for _ in range(3):
        for name, _m in modes:
            run(name)
    torch.cuda.synchronize(); sys.exit(0)
ref = run(modes[0][0])
for name, _m in modes + [modes[0]]:
    for _ in range(10):
        out = run(name)
    same = bool((out[0] == ref[0]).all()) and bool((out[1] == ref[1]).all())
    ts = []
    for rnd in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(name)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print(f"{T} x {H} x {W}: {name:24s} {len(prs[name]._fgvc_runs):4d} groups  min {min(ts):.3f} ms  same lists: {same}", flush=True)
