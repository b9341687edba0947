"""fgvc_corr_volume_f16f8: correctness against f32/f64 and timing next to bf16x3 / bf16 (GPU box)."""
import sys, time
sys.path.insert(0, ".")
import torch
from fgvc_amd import ops
dev = torch.device("cuda:0")
def ev_time(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (H, W) in [(23, 37), (120, 214), (128, 128), (180, 320), (97, 131)]:
    HW, C, tau = H * W, 256, 0.07
    g = torch.Generator(device=dev).manual_seed(1)
    f = torch.nn.functional.normalize(torch.randn(2, HW, C, generator=g, device=dev), dim=2)
    sp = ops.split_f16f8(f)
    hl = ops.split_bf16(f)
    v8 = ops.corr_volume(sp[1], sp[0], tau, "f16f8")
    v3 = ops.corr_volume(hl[1], hl[0], tau, "bf16x3")
    kk = torch.randint(0, HW, (20000,), device=dev); qq = torch.randint(0, HW, (20000,), device=dev)
    ref = (f[0][kk].double() * f[1][qq].double()).sum(1) / tau
    e8 = float((v8[kk, qq].double() - ref).abs().max()); e3 = float((v3[kk, qq].double() - ref).abs().max())
    d = float((v8 - v3).abs().max())
    print(f"{H}x{W}: f16f8 vs f64 sample max {e8:.2e}, bf16x3 {e3:.2e}, max |f16f8 - bf16x3| over the volume {d:.2e}")
    out = torch.empty_like(v8)
    res = {}
    for name, fn in (("f16f8", lambda: ops.corr_volume(sp[1], sp[0], tau, "f16f8", out=out)),
                     ("bf16x3", lambda: ops.corr_volume(hl[1], hl[0], tau, "bf16x3", out=out)),
                     ("bf16", lambda: ops.corr_volume(hl[1], hl[0], tau, "bf16", out=out))):
        res[name] = ev_time(fn)
    for dbg, label in ((8, "burst stores"), (4, "no classes"), (12, "burst, no classes"), (1, "no stores"), (2, "no MFMA"), (10, "no MFMA burst")):
        ops.set_option("corr8_debug", dbg)
        res[label] = ev_time(lambda: ops.corr_volume(sp[1], sp[0], tau, "f16f8", out=out))
    ops.set_option("corr8_debug", 0)
    gb = (HW * HW * 4 + 2 * HW * 1024) / 1e9
    print("   " + "  ".join(f"{k} {v:.3f} ms ({gb / v:.2f} TB/s)" for k, v in res.items()))
    del v8, v3, out
