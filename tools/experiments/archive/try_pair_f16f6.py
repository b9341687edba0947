"""fgvc_pair_topk_f16f6 against fgvc_pair_topk_f16x3 and the float64 slab: parity on a small clip (ragged edges), then timing at the
cfg2 shape (27 pairs of an 8-frame 480p clip), round-robin."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops
from oracle import fgvc_oracle as O
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = engine.TrackerConfig()


def lists(feats, H, W, fmt, pairs):
    sp = ops.split_f16x2(feats) if fmt == "f16" else ops.split_f16f6p(feats)
    return ops.pair_topk_split(sp, sp, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt=fmt)


for (H, W, T, kind) in ((37, 53, 3, "gauss"), (40, 56, 3, "relu"), (24, 40, 4, "smooth")):
    x = torch.randn(T, 256, H, W, device=dev)
    if kind == "relu":
        x = torch.relu(x)
    if kind == "smooth":
        x = torch.nn.functional.avg_pool2d(torch.randn(T, 256, H + 6, W + 6, device=dev), 7, 1) + 0.05 * x
    feats = ops.normalize_to_hwc(x)
    plan = engine.plan_clip(T, [0], cfg)
    pairs = ops.make_pairs(plan.pairs, dev)
    i3, s3 = lists(feats, H, W, "f16", pairs)
    i6, s6 = lists(feats, H, W, "f16f6", pairs)
    torch.cuda.synchronize()
    assert not ops.pair_f16x3_timed_out()
    same = (i3 == i6).all(-1).float().mean().item()
    fin = torch.isfinite(s3) & torch.isfinite(s6)
    ds = (s3 - s6)[fin].abs().max().item()
    print(f"{kind} {H}x{W}x{T}: lists equal on {same * 100:.2f} % of rows, max |score diff| {ds:.2e} cos = {ds / 0.07:.2e} logit; "
          f"empty entries equal: {bool(((i3 < 0) == (i6 < 0)).all())}")
    # against the float64 dense slab of one pair
    pi = len(plan.pairs) - 1
    qf, kf, _ = plan.pairs[pi]
    fq = feats[qf].double().cpu(); fk = feats[kf].double().cpu()
    dense = (fk @ fq.T)                                     # [key][query], cos
    m = O.mask_slab(H, W, H, W, 1, torch.arange(H * W), 30, "circle")
    st = O.check_topk(dense.masked_fill(~m, float("-inf")) / 0.07, i6[pi].cpu().long(), s6[pi].cpu() / 0.07, 10, tol=1e-3, gap=3e-4)
    print("   f16f6 vs float64:", st)

H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
sp3, sp6 = ops.split_f16x2(feats), ops.split_f16f6p(feats)


def ms(fn, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


f3 = lambda: ops.pair_topk_split(sp3, sp3, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
f6 = lambda: ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
for _ in range(30):
    f3(); f6()
res = {"f16x3": [], "f16f6": [], "f16f6 alternating list": [], "f16f6 hand-over only": [], "f16f6 LDS-DMA producers": [], "f16f6 LDS-DMA, alternating": []}
for rnd in range(4):
    res["f16x3"].append(ms(f3))
    res["f16f6"].append(ms(f6))
    ops.set_option("pair_f16_debug", 8192); ms(f6, 2); res["f16f6 alternating list"].append(ms(f6)); ops.set_option("pair_f16_debug", 0)
    ops.set_option("pair_f16_debug", 2048); ms(f6, 2); res["f16f6 hand-over only"].append(ms(f6)); ops.set_option("pair_f16_debug", 0)
    ops.set_option("pair_f16_debug", 32768); ms(f6, 2); res["f16f6 LDS-DMA producers"].append(ms(f6)); ops.set_option("pair_f16_debug", 0)
    ops.set_option("pair_f16_debug", 32768 + 8192); ms(f6, 2); res["f16f6 LDS-DMA, alternating"].append(ms(f6)); ops.set_option("pair_f16_debug", 0)
for k, v in res.items():
    print(f"{k:28s} min {min(v):.3f} ms  all {[round(x, 3) for x in v]}")
i3, s3 = f3(); i6, s6 = f6()
torch.cuda.synchronize()
print("timed out:", ops.pair_f16x3_timed_out(), " rows equal:", (i3 == i6).all(-1).float().mean().item(), " max score diff (logit):", ((s3 - s6).abs().max() / 0.07).item())
