import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C = 120, 214, 256; HW = H * W
feats = ops.normalize_to_hwc(torch.randn(2, C, H, W, device=dev))
hl = ops.split_bf16(feats)
vol = torch.empty((HW, HW), device=dev)
for kchunk in [int(v) for v in os.environ.get("KCHUNKS", "0").split(",")]:     # key blocks per workgroup (0 = the default)
    res = {"kchunk": kchunk}
    for dbg in (0, 1):
        ops.set_option("corr_debug", dbg | (kchunk << 8))
        for prec in ("bf16x3", "bf16"):
            m, _ = timeit(lambda: ops.corr_volume(hl[1], hl[0], 0.07, prec, out=vol), 5)
            res[f"{prec}_{'nostore' if dbg else 'full'}_ms"] = round(m, 4)
    print(json.dumps(res))
ops.set_option("corr_debug", 0)
