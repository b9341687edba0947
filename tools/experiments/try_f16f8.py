"""fgvc_corr_volume_f16f8: correctness against f64 and round-robin timing of its variants next to bf16x3 / bf16 (GPU box).
Timing: every configuration is timed in turn, several rounds; the minimum and the last round are printed (the first
configuration timed in a process runs up to 15 % slow -- never compare a first measurement with a later one)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgvc_amd import ops
dev = torch.device("cuda:0")
def ev_time(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
sizes = [(23, 37), (120, 214), (128, 128), (180, 320), (97, 131)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (H, W) in sizes:
    HW, C, tau = H * W, 256, 0.07
    g = torch.Generator(device=dev).manual_seed(1)
    f = torch.nn.functional.normalize(torch.randn(2, HW, C, generator=g, device=dev), dim=2)
    sp = ops.split_f16f8(f)
    hl = ops.split_bf16(f)
    kk = torch.randint(0, HW, (20000,), device=dev); qq = torch.randint(0, HW, (20000,), device=dev)
    ref = (f[0][kk].double() * f[1][qq].double()).sum(1) / tau
    v3 = ops.corr_volume(hl[1], hl[0], tau, "bf16x3")
    errs = {}
    for dbg, label in ((0, "default"), (8, "no stagger"), (16, "v1")):
        ops.set_option("corr8_debug", dbg)
        v8 = ops.corr_volume(sp[1], sp[0], tau, "f16f8")
        errs[label] = (float((v8[kk, qq].double() - ref).abs().max()), float((v8 - v3).abs().max()))
        del v8
    ops.set_option("corr8_debug", 0)
    print(f"{H}x{W}: max err vs f64 sample / vs bf16x3 whole volume: " + "  ".join(f"{k} {a:.1e}/{b:.1e}" for k, (a, b) in errs.items()))
    out = torch.empty_like(v3)
    del v3
    cfgs = [("f16f8", 0), ("no stagger", 8), ("no classes", 4), ("no stagger no classes", 12), ("v1 (32x32)", 16), ("no stores", 1), ("no stagger no stores", 9),
            ("kchunk 40", 40 << 8), ("kchunk 120", 120 << 8), ("kchunk 200", 200 << 8)]
    best, last = {}, {}
    def run_all():
        for name, dbg in cfgs:
            ops.set_option("corr8_debug", dbg)
            t = ev_time(lambda: ops.corr_volume(sp[1], sp[0], tau, "f16f8", out=out))
            best[name] = min(best.get(name, 1e9), t); last[name] = t
        ops.set_option("corr8_debug", 0)
        for name in ("bf16x3", "bf16"):
            t = ev_time(lambda: ops.corr_volume(hl[1], hl[0], tau, name, out=out))
            best[name] = min(best.get(name, 1e9), t); last[name] = t
    for _ in range(4):
        run_all()
    gb = (HW * HW * 4 + 2 * HW * 1024) / 1e9
    print("   " + "  ".join(f"{k} {v:.3f}/{last[k]:.3f} ms ({gb / v:.2f} TB/s)" for k, v in best.items()))
    del out
