import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(1)
N, H, W = 1, 8, 32
x = torch.zeros(N, 64, H, W)
wp, bias = ops.prepare_conv64(torch.zeros(64, 64, 3, 3).to(dev), torch.nn.BatchNorm2d(64).eval().to(dev))
xs = ops.nchw_to_split_nhwc(x.to(dev))
res = torch.arange(64, dtype=torch.float32).view(1, 64, 1, 1).expand(N, 64, H, W).contiguous() + 1.0 + 1.0 / 1024       # channel c -> c + 1 + 2^-10 (hi + lo)
rs = ops.nchw_to_split_nhwc(res.to(dev))
r32 = ops.unsplit_act(rs, ops.ACT_BF16X2)[:, 1:H + 1, 1:W + 1].contiguous()
a_f, b_f = ops.alloc_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
ops.conv64_split(xs, wp, bias, H, W, False, residual=r32, out_f32=a_f)
ops.conv64_split(xs, wp, bias, H, W, False, residual_split=rs, out_f32=b_f)
print("f32 identity  :", a_f[0, 0, 0, :16].tolist())
print("split identity:", b_f[0, 0, 0, :16].tolist())
print("pixel 5 split :", b_f[0, 3, 5, 30:40].tolist())
