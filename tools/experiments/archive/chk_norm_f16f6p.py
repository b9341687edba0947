import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
x = torch.relu(torch.randn(3, 17, 23, 256, device=dev)) * torch.rand(3, 17, 23, 1, device=dev)
x[0, 0, 0] = 0                      # an all-zero pixel
x[0, 0, 1, 5:] = 0                  # a pixel with all-zero blocks
a = ops.normalize_nhwc(x, True, split="f16f6")
b = ops.split_f16f6p(ops.normalize_nhwc(x, True))
ab, bb = a.view(torch.uint8).reshape(-1, 1024), b.view(torch.uint8).reshape(-1, 1024)
neq = (ab != bb)
print("rows", ab.shape[0], "differing bytes", int(neq.sum()), "rows with a difference", int(neq.any(1).sum()))
if neq.any():
    cols = neq.any(0).nonzero().flatten().tolist()
    print("byte offsets that differ:", cols[:40], "...", len(cols))
    r = neq.any(1).nonzero()[0, 0].item()
    c = neq[r].nonzero().flatten().tolist()[:8]
    print("row", r, "cols", c, "fused", ab[r, c].tolist(), "split", bb[r, c].tolist())
import numpy as np
f = ops.normalize_nhwc(x, True).reshape(-1, 256).cpu().numpy()
r = neq.any(1).nonzero()[0, 0].item()
cb = neq[r].nonzero().flatten().tolist()[0]
off = cb - 704
v, rem = off // 32, off % 32
hi, byte = rem // 16, rem % 16
e0 = (byte * 8) // 6
print("row", r, "byte", cb, "group", v, "hi", hi, "first element touching the byte", e0)
xs = f[r].astype(np.float32) * np.float32(256.0)
h = xs.astype(np.float16).astype(np.float32)
l = (xs - h) * np.float32(256.0)
ch = [64 * v + 16 * (e // 8) + 8 * hi + (e % 8) for e in range(32)]
lb = l[ch]
ml = np.abs(lb).max()
import math
s = math.ceil(math.log2(ml / 7.5)) if ml > 0 else -40
print("block max |l|", ml, "scale exp", s, "stored scale bytes fused/split", ab[r, 896 + 16 * hi + 4 + v].item(), bb[r, 896 + 16 * hi + 4 + v].item())
for e in (e0, e0 + 1):
    y = lb[e] / 2.0 ** s
    print("  element", e, "l", lb[e], "l / 2^s", y, " (x256 exact? xs", xs[ch[e]], "h", h[ch[e]], ")")
def codes(rowb):
    piece = bytes(rowb[704 + 32 * v + 16 * hi: 704 + 32 * v + 16 * hi + 16].tolist()) + bytes(rowb[832 + 32 * (v >> 1) + 16 * hi + 8 * (v & 1): 832 + 32 * (v >> 1) + 16 * hi + 8 * (v & 1) + 8].tolist())
    n = int.from_bytes(piece, "little")
    return [(n >> (6 * e)) & 63 for e in range(32)]
print("fused codes", codes(ab[r].cpu().numpy())[e0:e0 + 3], "split codes", codes(bb[r].cpu().numpy())[e0:e0 + 3])
