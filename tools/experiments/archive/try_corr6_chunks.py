"""fgvc_corr_volume_f16f6: half-chunks per tile pair (corr6_debug >> 12 overrides the cost model's choice), round-robin timed."""
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
cs = [0, 4, 5, 6, 7, 8, 10, 12, 15]
res = {c: [] for c in cs}
ref = None


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
for rnd in range(4):
    for c in cs:
        ops.set_option("corr6_debug", c << 12)
        if rnd == 0:
            vol.fill_(float("nan"))
            ops.corr_volume(sp[1], sp[0], 0.07, "f16f6", out=vol)
            if ref is None:
                ref = vol[::53].clone()
            assert torch.equal(vol[::53], ref), c
        ms(3)
        res[c].append(ms())
ops.set_option("corr6_debug", 0)
for c in cs:
    v = sorted(res[c])
    print(f"half-chunks per tile pair {c:2d}{' (cost model)' if c == 0 else ''}: min {v[0]:.4f}  median {v[len(v) // 2]:.4f} ms", flush=True)
