"""s_memtime probe of one workgroup of fgvc_corr_volume_f16f8 (corr8_debug = 32, + 1 without stores): cycles in the prologue,
and per 64-key stage in the two multiply phases, the two store bursts and wait + barrier."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
HW = 120 * 214
f = torch.nn.functional.normalize(torch.randn(2, HW, 256, device=dev), dim=2)
sp = ops.split_f16f8(f)
vol = torch.empty((HW, HW), device=dev)
for dbg, name in ((32, "with stores"), (33, "no stores")):
    for _ in range(3):
        ops.set_option("corr8_debug", dbg)
        ops.corr_volume(sp[1], sp[0], 0.07, "f16f8", out=vol)
    torch.cuda.synchronize()
    ops.set_option("corr8_debug", 0)
    v = vol.view(-1)[:128].view(torch.int64).view(8, 8).cpu()
    print(name)
    for w in range(8):
        pro, comp, st, sync, tot, ns = v[w, :6].tolist()
        print(f"  wave {w}: prologue {pro:6d} | per stage: multiply {comp / ns:6.0f}  stores {st / ns:6.0f}  wait+barrier {sync / ns:6.0f} = {(comp + st + sync) / ns:6.0f} | total {tot} cycles, {ns} stages")
