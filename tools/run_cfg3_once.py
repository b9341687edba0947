"""The kernels of BASELINE configs[2] (the local window R = 6 on the 480 x 854 stride-1 grid, 6 key slots, on a bank split once; the
coarse-to-fine fine stage at 120 x 214 coarse / 480 x 856 fine), three launches each, for rocprofv3 --kernel-trace and --pmc passes
(tools/pmc_passes.sh r06_cfg3 tools/run_cfg3_once.py).  Run the interpreter directly after `--`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
T, C, k = 6, 256, 10
H1, W1, R1 = 480, 854, 6
g = torch.Generator(device=dev).manual_seed(3)
bank = torch.empty(T + 1, H1 * W1, 2, 256, dtype=torch.int16, device=dev)
for t in range(T + 1):
    bank[t] = ops.split_f16x2(ops.normalize_to_hwc(torch.randn(1, C, H1, W1, device=dev, generator=g)))[0]
for _ in range(3):
    ops.local_corr_topk(bank[:1], bank[1:], H1, W1, R1, k, 0.07, presplit=True)
torch.cuda.synchronize()
del bank
torch.cuda.empty_cache()
H, W, scale, Cf, Rf, P = 120, 214, 4, 64, 6, 16
fine = ops.normalize_to_hwc(torch.randn(T + 1, Cf, H * scale, W * scale, device=dev))
vfine = torch.rand(T, H * scale * W * scale, P, device=dev)
# the fine windows' centres as the coarse stage gives them (bench_cfg3.py): the arg-max of the masked coarse affinity -- on these noise features
# a uniformly random cell of the radius-15 disc, the WORST case for the fine stage's locality (a real video's motion is coherent)
coarse = ops.normalize_to_hwc(torch.randn(T + 1, C, H, W, device=dev))
cidx, _ = ops.pair_topk_auto(coarse, coarse, ops.make_pairs([(0, 1 + t, True) for t in range(T)], dev), H, W, H, W,
                             ops.MaskSpec.from_neighbor_range(30), 1, normalized=True)
arg = cidx[:, :, 0].clamp_min(0).contiguous()
del coarse
for _ in range(3):
    ops.c2f_refine(arg, fine[0], fine[1:], vfine, H, W, scale, Rf, k, 0.07)
torch.cuda.synchronize()
