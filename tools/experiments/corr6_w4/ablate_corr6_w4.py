"""four-wave volume kernel (four_waves.patch applied), all without stores: what the multiply parts cost with and without their LDS
fragment reads and key-row DMAs.   FGVC_HIP_LIB=fgvc_amd/lib/libfgvc_hip_ablations.so python tools/experiments/corr6_w4/ablate_corr6_w4.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = 120, 214
HW = H * W
f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
sp = ops.split_f16f6(f)
vol = torch.empty((HW, HW), device=dev)
names = {1: "W8 no stores", 2048 + 1: "W4 no stores", 2048 + 65: "W4 no stores, no LDS fragment reads", 2048 + 129: "W4 no stores, no DMA", 2048 + 193: "W4 no stores, neither",
         0: "W8", 2048: "W4", 1024: "W8 stores only", 2048 + 1024: "W4 stores only", 2: "W8 no MFMAs", 2048 + 2: "W4 no MFMAs"}
res = {k: [] for k in names}
for rnd in range(4):
    for dbg in names:
        ops.set_option("corr6_debug", dbg)
        for _ in range(2):
            ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
        e1.record(); torch.cuda.synchronize()
        res[dbg].append(e0.elapsed_time(e1) / 10)
ops.set_option("corr6_debug", 0)
for dbg, name in names.items():
    print(f"{name:40s} min {min(res[dbg]):.4f} ms   all {[round(x, 4) for x in res[dbg]]}")
