"""fgvc_corr_volume_f16f6: stages moved from the two-segment (middle) piece to the others (corr6_skew), round-robin timed; the volume must
not change."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = (120, 214) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
HW = H * W
f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
sp = ops.split_f16f6(f)
vol = torch.empty((HW, HW), device=dev)
ref = None
skews = [2, 102, 0, 100]          # + 100: staging DMAs with 64-bit lane addresses (corr6_sdma = 0)
res = {k: [] for k in skews}


def ms(reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(300):
    ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
for rnd in range(5):
    for k in skews:
        ops.set_option("corr6_skew", k % 100)
        ops.set_option("corr6_sdma", 0 if k >= 100 else 1)
        vol.fill_(float("nan"))
        ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
        if ref is None:
            ref = vol[::97].clone()
        assert torch.equal(vol[::97], ref), k
        ms(3)
        res[k].append(ms())
ops.set_option("corr6_skew", 0)
ops.set_option("corr6_sdma", 1)
for k in skews:
    v = sorted(res[k])
    print(f"skew {k:3d}: min {v[0]:.4f}  median {v[len(v) // 2]:.4f} ms", flush=True)
