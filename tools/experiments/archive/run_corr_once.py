import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C = 120, 214, 256; HW = H * W
feats = ops.normalize_to_hwc(torch.randn(2, C, H, W, device=dev))
hl = ops.split_bf16(feats)
vol = torch.empty((HW, HW), device=dev)
for prec in ("bf16x3", "bf16", "f32"):
    for _ in range(3):
        if prec == "f32":
            ops.corr_volume(feats[1], feats[0], 0.07, prec, out=vol)
        else:
            ops.corr_volume(hl[1], hl[0], 0.07, prec, out=vol)
torch.cuda.synchronize()
