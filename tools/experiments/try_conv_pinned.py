"""fgvc_conv_split_f32 256 -> 256: the hand-placed operand reads (default) against the compiler-scheduled stage (conv_debug = 16), round-robin."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, H, W = 8, 120, 214
wt = torch.randn(256, 256, 3, 3, device=dev) * 0.02
bn = torch.nn.BatchNorm2d(256).eval().to(dev)
wp, bs = ops.prepare_conv_split(wt, bn)
xs = ops.nchw_to_split_nhwc(torch.randn(N, 256, H, W, device=dev))
ys = ops.alloc_split_nhwc(N, 256, H, W, dev)
fn = lambda: ops.conv_split(xs, wp, bs, H, W, True, out_split=ys)


def ms(reps=40):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(500):
    fn()
res = {0: [], 16: [], 32: []}
ref = None
for rnd in range(6):
    for m in (0, 16):
        ops.set_option("conv_debug", m)
        fn()
        if rnd == 0:
            out = ys.clone()
            if ref is None:
                ref = out
            print(m, 'equal to the default form:', bool(torch.equal(out, ref)), float((out.float() - ref.float()).abs().max()), flush=True)
        res[m].append(ms())
ops.set_option("conv_debug", 0)
print({m: (round(min(v), 4), round(sorted(v)[len(v) // 2], 4)) for m, v in res.items()})
