"""fgvc_pair_topk_f16x3 (27 pairs of an 8-frame 480p clip) or the dense volume kernel in a loop for a few seconds, for
tools/experiments/watch_clocks.sh: does the kernel alone reach the board's power cap?   python loop_pair.py [pair|volume] [seconds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
what = sys.argv[1] if len(sys.argv) > 1 else "pair"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
if what == "pair":
    cfg = engine.TrackerConfig()
    plan = engine.plan_clip(T, [0], cfg)
    pairs = ops.make_pairs(plan.pairs, dev)
    h16 = ops.split_f16x2(feats)
    fn = lambda: ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True)
else:
    sp6 = ops.split_f16f6(feats[:2])
    vol = torch.empty((H * W, H * W), device=dev)
    fn = lambda: ops.corr_volume(sp6[1], sp6[0], 0.07, "f16f6", out=vol)
fn(); torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(f"{what}: {dt / n * 1e3:.4f} ms per launch")
