import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
src = open("tools/experiments/chk_conv64p.py").read().split("bad = 0")[0]
exec(src)
N, H, W = 1, 16, 96
g = torch.Generator().manual_seed(5)
wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).to(dev)
bn = torch.nn.BatchNorm2d(64).eval().to(dev)
w1, b1, sw = ops.prepare_conv64_f16(wt, bn)
x = (torch.randn(N, 64, H, W, generator=g).abs() ** 1.5).to(dev)
sx = ops.act_scale_log2(float(x.abs().max()))
xs = pack(x, sx)
z = torch.zeros(N, H, W, 64, device=dev)
form = FORMS["conv2 (residual, f32 + split f16f8 out)"]
a = run(0, xs, w1, b1, sw, sx, H, W, z, form, False, 3)
b = run(16, xs, w1, b1, sw, sx, H, W, z, form, False, 3)
ref = F.conv2d(x.double(), wt.double(), padding=1)
sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().view(1, -1, 1, 1)
ref = ((ref - bn.running_mean.double().view(1, -1, 1, 1)) * sc + bn.bias.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
scale = ref.abs().max().item()
print("max |new - ref| / scale", ((a[1].double() - ref).abs().max() / scale).item(), " max |old - ref| / scale", ((b[1].double() - ref).abs().max() / scale).item())
d = (a[1] - b[1]).abs()
print("f32 words differing:", (d > 0).sum().item(), "of", d.numel(), " max |new - old| / scale", (d.max() / scale).item())
print("by row:", (d > 0)[0].sum((1, 2)).tolist())
ulp = (d / (b[1].abs() * 2 ** -23 + 1e-30))
print("differences in ulps of the old value: median", ulp[d > 0].median().item(), "max", ulp[d > 0].max().item())
