import os, sys, ctypes
sys.path.insert(0, "/root/repo")
import torch
from fgvc_amd import engine, ops, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = engine.TrackerConfig()
H, W, T = 37, 53, 3
feats = ops.normalize_to_hwc(torch.randn(T, 256, H, W, device=dev))
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
sp6 = ops.split_f16f6p(feats)
i6, s6 = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
torch.cuda.synchronize()
buf = (ctypes.c_int64 * 32)()
_lib.call("fgvc_pair_topk_f16x3_probe", ctypes.cast(buf, ctypes.c_void_p))
print("flag", ops.pair_f16x3_timed_out(), [hex(v & 0xffffffffffffffff) for v in list(buf)[:8]])
