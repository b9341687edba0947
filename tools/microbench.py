"""Kernel micro-benchmarks (HIP-event timed) at BASELINE config sizes.  GPU box only.

    python tools/microbench.py [--hw 120x214] [--C 256] [--reps 5]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import engine, ops  # noqa: E402


def timeit(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", default="120x214")
    ap.add_argument("--C", type=int, default=256)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-dense", action="store_true")
    a = ap.parse_args()
    H, W = map(int, a.hw.split("x"))
    HW, C, T = H * W, a.C, a.frames
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    raw = torch.randn(T, C, H, W, device=dev)
    res = {"H": H, "W": W, "C": C, "frames": T, "device": torch.cuda.get_device_name(0)}

    med, best = timeit(lambda: ops.normalize_to_hwc(raw), a.reps)
    res["normalize_ms"] = med
    res["normalize_GBps"] = 2 * raw.numel() * 4 / med / 1e6
    feats = ops.normalize_to_hwc(raw)

    cfg = engine.TrackerConfig()
    plan = engine.plan_clip(T, [0], cfg)
    n_pairs = len(plan.pairs)
    pairs = ops.make_pairs(plan.pairs, dev)
    mask = cfg.mask
    med, best = timeit(lambda: ops.pair_topk(feats, feats, pairs, H, W, H, W, mask, 10, validate=False), a.reps)
    # algorithmic windowed FLOPs: 2 * HW * N_disc * C per pair (interior disc count)
    ndisc = sum(1 for dy in range(-15, 16) for dx in range(-15, 16) if dy * dy + dx * dx <= mask.r2max)
    res.update(pair_topk_ms=med, pair_topk_best_ms=best, n_pairs=n_pairs, ms_per_pair=med / n_pairs,
               n_disc=ndisc, pair_topk_TFLOPs_windowed=2 * HW * ndisc * C * n_pairs / med / 1e9,
               pair_topk_frac_f32_peak=2 * HW * ndisc * C * n_pairs / med / 1e9 / 157.3)

    one = ops.make_pairs([(1, 0)], dev)
    med1, _ = timeit(lambda: ops.pair_topk(feats, feats, one, H, W, H, W, mask, 10, validate=False), a.reps)
    res["pair_topk_single_ms"] = med1
    nm = ops.make_pairs([(1, 0, False)], dev)
    if HW <= 32768:
        medn, _ = timeit(lambda: ops.pair_topk(feats, feats, nm, H, W, H, W, mask, 10, validate=False), max(2, a.reps // 2), 1)
        res["pair_topk_fullframe_ms"] = medn
        res["pair_topk_fullframe_TFLOPs"] = 2.0 * HW * HW * C / medn / 1e9

    tk = engine.run_affinity(feats, H, W, plan, cfg)
    med, _ = timeit(lambda: engine.run_affinity(feats, H, W, plan, cfg), a.reps)
    res["run_affinity_ms"] = med
    pts = torch.rand(16, 2, device=dev) * torch.tensor([W * 4.0, H * 4.0], device=dev)
    h, w = H * 4, W * 4
    med, _ = timeit(lambda: engine.run_propagation(tk, 0, pts, H, W, h, w, cfg), a.reps)
    res["run_propagation_ms"] = med

    if not a.skip_dense:
        vol = torch.empty((HW, HW), device=dev, dtype=torch.float32)
        gb = HW * HW * 4 / 1e9
        med, _ = timeit(lambda: ops.corr_volume(feats[1], feats[0], 0.07, "f32", out=vol), a.reps)
        res.update(corr_f32_ms=med, corr_f32_TBps=gb / med, corr_f32_TFLOPs=2.0 * HW * HW * C / med / 1e9)
        qs = ops.split_bf16(feats)
        med, _ = timeit(lambda: ops.corr_volume(qs[1], qs[0], 0.07, "bf16x3", out=vol), a.reps)
        res.update(corr_bf16x3_ms=med, corr_bf16x3_TBps=gb / med, corr_bf16x3_TFLOPs=6.0 * HW * HW * C / med / 1e9)
        med, _ = timeit(lambda: ops.corr_volume(qs[1], qs[0], 0.07, "bf16", out=vol), a.reps)
        res.update(corr_bf16_ms=med, corr_bf16_TBps=gb / med, corr_bf16_TFLOPs=2.0 * HW * HW * C / med / 1e9)
        med, _ = timeit(lambda: vol.fill_(1.0), a.reps)
        res.update(fill_ms=med, fill_TBps=gb / med)
        med, _ = timeit(lambda: ops.split_bf16(feats), a.reps)
        res["split_bf16_ms"] = med
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
