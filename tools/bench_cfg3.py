"""BASELINE.json configs[2]: coarse-to-fine two-scale local-window correlation (fine radius 6) and the single-scale
local window, at 480p sizes (coarse 120x214x256 stride-4 features, fine 480x856x64 stride-1 features, 6 key slots)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


H, W, C, T, P, scale, Cf, Rf, k = 120, 214, 256, 6, 16, 4, 64, 6, 10
HW = H * W
coarse = ops.normalize_to_hwc(torch.randn(T + 1, C, H, W, device=dev))                 # frame 0 = query
fine = ops.normalize_to_hwc(torch.randn(T + 1, Cf, H * scale, W * scale, device=dev))
vfine = torch.rand(T, H * scale * W * scale, P, device=dev)
mask = ops.MaskSpec.from_neighbor_range(30)
pairs = ops.make_pairs([(0, 1 + t, True) for t in range(T)], dev)
t_coarse = timeit(lambda: ops.pair_topk_auto(coarse, coarse, pairs, H, W, H, W, mask, 1, normalized=True))
cidx, _ = ops.pair_topk_auto(coarse, coarse, pairs, H, W, H, W, mask, 1, normalized=True)
arg = cidx[:, :, 0].clamp_min(0).contiguous()
t_fine = timeit(lambda: ops.c2f_refine(arg, fine[0], fine[1:], vfine, H, W, scale, Rf, k, 0.07))
cand = T * (2 * Rf + 1) ** 2
print(f"c2f (A6) per query frame, {T} key slots: coarse arg-max stage {t_coarse:.3f} ms + fine stage {t_fine:.3f} ms "
      f"({HW} queries x {cand} fine candidates x {Cf} ch = {2.0 * HW * cand * Cf / t_fine / 1e9:.1f} TFLOP/s f32 VALU)")
R = 6
lw = ops.normalize_to_hwc(torch.randn(T + 1, C, H, W, device=dev))
for R in (6, 12):
    t_f32 = timeit(lambda: ops.local_corr_topk(lw[:1], lw[1:], H, W, R, k, 0.07))
    t_split = timeit(lambda: ops.local_corr_topk(lw[:1], lw[1:], H, W, R, k, 0.07, normalized=True))
    print(f"local window (A7) radius {R}, {T} key slots at {H}x{W}x{C}: f32-MFMA kernel {t_f32:.3f} ms, f16x3 kernel "
          f"{t_split:.3f} ms per query frame (the 16-bit form includes its split pass)")

# ---- the configuration as BASELINE.json states it: the single-scale local window on the stride-1 grid (480 x 854 x 256, radius 6, 6
#      key slots): 409 920 queries x 6 x 169 candidates x 256 channels = 2.13e11 FLOP, 7 frames x 420 MB of f32 features
del coarse, fine, vfine, lw
torch.cuda.empty_cache()
H1, W1, R1 = 480, 854, 6
g = torch.Generator(device=dev).manual_seed(3)
big = torch.empty(T + 1, H1 * W1, C, device=dev)
for t in range(T + 1):                                            # frame by frame: the f32 NCHW source of a frame is 420 MB
    big[t] = ops.normalize_to_hwc(torch.randn(1, C, H1, W1, device=dev, generator=g))[0]
t_split = timeit(lambda: ops.local_corr_topk(big[:1], big[1:], H1, W1, R1, k, 0.07, normalized=True), reps=3)
q16, k16 = ops.split_f16x2(big[:1]), ops.split_f16x2(big[1:])
t_s = timeit(lambda: (ops.split_f16x2(big[:1]), ops.split_f16x2(big[1:])), reps=3)
t_pre = timeit(lambda: ops.local_corr_topk(q16, k16, H1, W1, R1, k, 0.07, presplit=True), reps=3)
a, b = ops.local_corr_topk(big[:1], big[1:], H1, W1, R1, k, 0.07, normalized=True), ops.local_corr_topk(q16, k16, H1, W1, R1, k, 0.07, presplit=True)
assert all(torch.equal(x, y) for x, y in zip(a, b))
fl = 2.0 * H1 * W1 * T * (2 * R1 + 1) ** 2 * C
gb = (T + 1) * H1 * W1 * C * 4 / 1e9
print(f"local window (A7) radius {R1}, {T} key slots at {H1}x{W1}x{C} (configs[2] as stated): f16x3 kernel {t_split:.2f} ms per query frame from f32 rows "
      f"(incl. {t_s:.2f} ms for splitting the 7 frames), {t_pre:.2f} ms on a bank split once (presplit=True: identical lists) = "
      f"{fl / 1e9:.0f} GFLOP windowed -> {fl / (t_pre * 1e-3) / 1e12:.0f} TFLOP/s f32-grade; features read {gb:.2f} GB -> {gb / (t_pre * 1e-3) / 1e3:.2f} TB/s = "
      f"{gb / (t_pre * 1e-3) / 8e3:.3f} of the 8 TB/s roofline (SURVEY 8(d): 0.37 ms at 8 TB/s)")
