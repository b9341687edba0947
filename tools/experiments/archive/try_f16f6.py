"""fgvc_corr_volume_f16f6: split-format check, correctness against f64 / bf16x3 and round-robin timing next to f16f8 / bf16x3 / bf16
(GPU box).  Every configuration is timed in turn, several rounds; minimum and last round are printed (the first configuration
timed in a process runs slow -- never compare a first measurement with a later one).
    python tools/experiments/try_f16f6.py [HxW ...] [--quick]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from fgvc_amd import ops

dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
quick = "--quick" in sys.argv


def ev_time(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def decode_f16f6(sp):
    """(n, 1024) uint8 rows -> h (n, 256) f64, h6 and l6 dequantised (n, 256) f64 incl. their 2^(s-4) scales"""
    sp = sp.cpu().numpy()
    n = sp.shape[0]
    h = sp[:, :512].copy().view(np.float16).astype(np.float64)
    lut = np.array([(m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)) for e in range(4) for m in range(8)])
    lut = np.concatenate([lut, -lut])
    outs = []
    for base in (512, 704):
        vals = np.zeros((n, 256))
        for u in range(2):
            for g in range(4):
                b16 = sp[:, base + 96 * u + 16 * g: base + 96 * u + 16 * g + 16]
                b8 = sp[:, base + 96 * u + 64 + 8 * g: base + 96 * u + 64 + 8 * g + 8]
                bits = np.unpackbits(np.concatenate([b16, b8], axis=1), axis=1, bitorder="little")     # (n, 192)
                codes = (bits.reshape(n, 32, 6) * (1 << np.arange(6))).sum(-1)
                which = u if base == 512 else 2 + u
                sc = sp[:, 896 + 4 * g + which].astype(np.int64) - 127
                vals[:, 128 * u + 32 * g: 128 * u + 32 * g + 32] = lut[codes] * (2.0 ** sc)[:, None]
        outs.append(vals)
    return h, outs[0], outs[1]


# ---- the split format against its definition
g = torch.Generator(device=dev).manual_seed(5)
f = torch.nn.functional.normalize(torch.randn(4096, 256, generator=g, device=dev), dim=1)
f[:64] = 0
f[:64, 3] = 1.0
f[64:128] *= (torch.rand(64, 256, device=dev, generator=g) < 0.05)
f[64:128] = torch.nn.functional.normalize(f[64:128] + 1e-6, dim=1)
sp = ops.split_f16f6(f)
h, h6, l6 = decode_f16f6(sp)
x = f.cpu().double().numpy()
h_ref = (x.astype(np.float32) * np.float32(256)).astype(np.float16).astype(np.float64)
l_ref = (x * 256 - h_ref) * 256
assert (h == h_ref).all(), "h differs"
assert (sp[:, 912:] == 0).all()
# h6 / l6 carry 2^-4 each; e2m3 inside a block: |err| <= max(2^-4 |v|, blockmax / 120) (half a step)
for name, got, ref in (("h6", h6 * 16, h_ref), ("l6", l6 * 16, l_ref)):
    bm = np.abs(ref).reshape(-1, 8, 32).max(-1, keepdims=True).repeat(32, -1).reshape(ref.shape)
    bound = np.maximum(np.abs(ref) / 16, bm / 7.5 / 8) * 1.0001 + 1e-30      # half of the step 2 bm / 7.5 / 8 (worst scale choice)
    err = np.abs(got - ref)
    print(f"split {name}: max err / bound = {(err / bound).max():.3f}, rms rel err {np.sqrt((err ** 2).sum() / (ref ** 2).sum()):.3e}")
    assert (err <= bound).all(), name

sizes = [(23, 37), (120, 214), (128, 128), (97, 131)] if not args else [tuple(int(v) for v in a.split("x")) for a in args]
for (H, W) in sizes:
    HW, C, tau = H * W, 256, 0.07
    g = torch.Generator(device=dev).manual_seed(1)
    f = torch.nn.functional.normalize(torch.randn(2, HW, C, generator=g, device=dev), dim=2)
    s6, s8, hl = ops.split_f16f6(f), ops.split_f16f8(f), ops.split_bf16(f)
    kk = torch.randint(0, HW, (20000,), device=dev)
    qq = torch.randint(0, HW, (20000,), device=dev)
    ref = (f[0][kk].double() * f[1][qq].double()).sum(1) / tau
    v3 = ops.corr_volume(hl[1], hl[0], tau, "bf16x3")
    errs = {}
    for dbg, label in ((0, "default"), (8, "no stagger"), (4, "no classes"), (16, "late dma"), (64, "whole F"), (80, "both")):
        ops.set_option("corr6_debug", dbg)
        v6 = ops.corr_volume(s6[1], s6[0], tau, "f16f6")
        errs[label] = (float((v6[kk, qq].double() - ref).abs().max()), float((v6 - v3).abs().max()))
        del v6
    ops.set_option("corr6_debug", 0)
    print(f"{H}x{W}: max err vs f64 sample / vs bf16x3 whole volume: " + "  ".join(f"{k} {a:.1e}/{b:.1e}" for k, (a, b) in errs.items()))
    out = torch.empty_like(v3)
    del v3
    if quick:
        continue
    cfgs = [("f16f6", 0), ("f6 pair-major ids", 128), ("f6 no stagger", 8), ("f6 late dma", 16), ("f6 whole F", 64), ("f6 late dma whole F", 80), ("f6 no stores", 1), ("f6 no mfma", 2),
            ("f6 neither", 3)]
    for c in (4, 5, 7, 10):
        cfgs.append((f"f6 {c / 2:g} chunks/tile", c << 12))
    best, last = {}, {}

    def run_all():
        for name, dbg in cfgs:
            ops.set_option("corr6_debug", dbg)
            t = ev_time(lambda: ops.corr_volume(s6[1], s6[0], tau, "f16f6", out=out))
            best[name] = min(best.get(name, 1e9), t)
            last[name] = t
        ops.set_option("corr6_debug", 0)
        t = ev_time(lambda: ops.corr_volume(s8[1], s8[0], tau, "f16f8", out=out))
        best["f16f8"] = min(best.get("f16f8", 1e9), t)
        last["f16f8"] = t
        for name in ("bf16x3", "bf16"):
            t = ev_time(lambda: ops.corr_volume(hl[1], hl[0], tau, name, out=out))
            best[name] = min(best.get(name, 1e9), t)
            last[name] = t
    for _ in range(4):
        run_all()
    gb = (HW * HW * 4 + 2 * HW * 1024) / 1e9
    for k, v in best.items():
        print(f"   {k:22s} {v:.3f} / {last[k]:.3f} ms  ({gb / v:.2f} TB/s, {gb / v / 8:.3f} of 8 TB/s)")
    del out
