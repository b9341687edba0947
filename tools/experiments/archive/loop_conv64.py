"""fgvc_conv64_split_f32 in a loop for a few seconds (for tools/experiments/watch_clocks.sh): prints launches per second."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W = 8, 240, 427
w6, b6 = ops.prepare_conv64((torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev), torch.nn.BatchNorm2d(64).eval().to(dev))
xs = ops.nchw_to_split_nhwc(torch.randn(N, 64, H, W, generator=g).to(dev))
o_s = ops.alloc_split_nhwc(N, 64, H, W, dev)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(200):
        ops.conv64_split(xs, w6, b6, H, W, True, out_split=o_s)
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print(f"{n / dt:.0f} launches/s = {dt / n * 1e3:.4f} ms per launch ({os.environ.get('FGVC_HIP_LIB', 'in-tree')})")
