"""Energy per launch of the step's kernels at the bench's shapes (8 x 480 x 854 clip): every kernel looped alone for a few seconds while
rocm-smi is polled beside it (a child started BEFORE this process touches the GPU); joules per launch = mean socket power x time per
launch, and the share above the idle floor.      python3 tools/experiments/joules_table.py [seconds per kernel] > profiles/rNN_joules.log"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = os.path.join(ROOT, "gpurun_out", "joules_smi.log")
open(LOG, "w").close()
sampler = subprocess.Popen(["bash", "-c", f"while true; do echo \"$(date +%s.%N) $(rocm-smi -d 0 --showpower --showclocks 2>/dev/null | grep -i 'power\\|sclk' | tr '\\n' ' ')\" >> {LOG}; sleep 0.2; done"])
time.sleep(2.5)                      # idle floor
import torch
sys.path.insert(0, ROOT)
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
T, H1, W1, H, W = 8, 240, 427, 120, 214
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
bn = lambda c: torch.nn.BatchNorm2d(c).eval().to(dev)
kern = []          # (name, launches per 8-frame clip, fn)
frames = torch.randn(T, 3, 480, 854, device=dev)
sw_, sb_ = ops.prepare_stem7(torch.randn(64, 3, 7, 7, device=dev) * 0.1, bn(64))
st_s, st_f = ops.alloc_split_nhwc(T, 64, H1, W1, dev), ops.alloc_nhwc(T, 64, H1, W1, dev)
kern.append(("stem7 3->64 7x7 s2 (f32 NCHW in, split + f32 out)", 1, lambda: ops.stem7_split(frames, sw_, sb_, True, out_split=st_s, out_f32=st_f)))
ops.stem7_split(frames, sw_, sb_, True, out_split=st_s, out_f32=st_f)
w64, b64 = ops.prepare_conv64(torch.randn(64, 64, 3, 3, device=dev) * 0.05, bn(64))
y64 = ops.alloc_split_nhwc(T, 64, H1, W1, dev); y64f = ops.alloc_nhwc(T, 64, H1, W1, dev)
kern.append(("conv64 64->64 3x3 (layer 1; residual in, split + f32 out)", 4, lambda: ops.conv64_split(st_s, w64, b64, H1, W1, True, residual=st_f, out_split=y64, out_f32=y64f)))
w2, b2 = ops.prepare_conv_s2(torch.randn(128, 64, 3, 3, device=dev) * 0.05, bn(128))
s2o = ops.alloc_split_nhwc(T, 128, H, W, dev)
kern.append(("conv_s2 64->128 3x3 s2 (f16f6 out)", 1, lambda: ops.conv_s2_split(st_s, w2, b2, H1, W1, True, out_split=s2o, out_fmt=ops.ACT_F16F6, out_scale_log2=3, overflow=ovf)))
w2d, b2d = ops.prepare_conv_s2(torch.randn(128, 64, 1, 1, device=dev) * 0.1, bn(128))
s2f = ops.alloc_nhwc(T, 128, H, W, dev)
kern.append(("conv_s2 64->128 1x1 s2 (downsample, f32 out)", 1, lambda: ops.conv_s2_split(st_s, w2d, b2d, H1, W1, False, out_f32=s2f)))
ops.conv_s2_split(st_s, w2, b2, H1, W1, True, out_split=s2o, out_fmt=ops.ACT_F16F6, out_scale_log2=3, overflow=ovf)
def s1(cin, cout, ks, arith, x_in, res=None):
    fmt = ops.ACT_FMT[arith]
    wt = torch.randn(cout, cin, ks, ks, device=dev) * (2.0 / (cin * ks * ks)) ** 0.5
    wp, bs, sw = ops.prepare_conv_split_f16(wt, bn(cout), fmt)
    yo = ops.alloc_split_nhwc(T, cout, H, W, dev); yf = ops.alloc_nhwc(T, cout, H, W, dev)
    return yo, (lambda: ops.conv_split(x_in, wp, bs, H, W, True, residual=res, out_split=yo, out_f32=yf, in_fmt=fmt, in_scale_log2=3 + sw, out_fmt=fmt, out_scale_log2=3, overflow=ovf))
y128, f128 = s1(128, 128, 3, "f16f6", s2o, s2f)
kern.append(("conv_split 128->128 3x3 f16f6 (residual in, split + f32 out)", 3, f128)); f128()
y256a, f256a = s1(128, 256, 3, "f16f6", y128)
kern.append(("conv_split 128->256 3x3 f16f6", 1, f256a)); f256a()
y256d, f256d = s1(128, 256, 1, "f16f6", y128)
kern.append(("conv_split 128->256 1x1 f16f6 (downsample)", 1, f256d))
y256, f256 = s1(256, 256, 3, "f16f6", y256a)
kern.append(("conv_split 256->256 3x3 f16f6", 3, f256))
# the same layer in round 3's arithmetic (operands re-made in its own format)
x8 = ops.alloc_split_nhwc(T, 256, H, W, dev)
wt_ = torch.randn(256, 128, 3, 3, device=dev) * 0.03
wp0, bs0 = ops.prepare_conv_split(wt_, bn(256))
ops.conv_split(ops.nchw_to_split_nhwc(torch.relu(torch.randn(T, 128, H, W, device=dev))), wp0, bs0, H, W, True, out_split=x8, out_fmt=ops.ACT_F16F8, out_scale_log2=3, overflow=ovf)
_, f256_8 = s1(256, 256, 3, "f16f8", x8)
kern.append(("conv_split 256->256 3x3 f16f8 (round 3, for comparison)", 0, f256_8))
feat_f = ops.alloc_nhwc(T, 256, H, W, dev); feat_f.normal_()
kern.append(("normalize + f16f6p split of the features", 1, lambda: ops.normalize_nhwc(feat_f, split="f16f6")))
feats = ops.normalize_to_hwc(torch.randn(T, 256, H, W, device=dev))
cfg = engine.TrackerConfig(); plan = engine.plan_clip(T, [0], cfg); pairs = ops.make_pairs(plan.pairs, dev)
sp6 = ops.split_f16f6p(feats); sp3 = ops.split_f16x2(feats)
kern.append(("pair_topk f16f6, 27 pairs", 1, lambda: ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")))
kern.append(("pair_topk f16x3, 27 pairs (round 3, for comparison)", 0, lambda: ops.pair_topk_split(sp3, sp3, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")))
spv = ops.split_f16f6(feats[:2]); vol = torch.empty((H * W, H * W), device=dev)
kern.append(("corr_volume f16f6 (not part of the step)", 0, lambda: ops.corr_volume(spv[1], spv[0], 0.07, "f16f6", out=vol)))
torch.cuda.synchronize()
time.sleep(2.0)
wins = []
for name, per_clip, fn in kern:
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(40):
            fn()
        torch.cuda.synchronize(); n += 40
    t1 = time.time()
    wins.append((name, per_clip, t0, t1, n))
    time.sleep(1.5)                  # back to idle between kernels
assert int(ovf.item()) in (0, 1)
sampler.terminate(); sampler.wait()
samples = []
for line in open(LOG):
    p = line.split()
    try:
        t = float(p[0]); w = float(line.split("(W):")[1].split()[0])
        clk = line.split("sclk clock level")[1].split("(")[1].split("Mhz")[0] if "sclk clock level" in line else "?"
        samples.append((t, w, clk))
    except Exception:
        pass
idle = [w for t, w, _ in samples if t < wins[0][2] - 0.2 and t > samples[0][0] + 0.5]
idle_w = sum(idle) / max(len(idle), 1)
print(f"idle floor before the first kernel: {idle_w:.0f} W ({len(idle)} samples); {secs:.0f} s per kernel, rocm-smi every ~0.25 s")
print(f"{'kernel':64s} {'ms/launch':>9s} {'W':>6s} {'sclk':>5s} {'J/launch':>9s} {'above idle':>10s} {'per clip':>8s} {'J/clip':>7s}")
tot = tot_ms = 0.0
for name, per_clip, t0, t1, n in wins:
    ws = [(w, c) for t, w, c in samples if t0 + 0.8 < t < t1 - 0.1]
    pw = sum(w for w, _ in ws) / max(len(ws), 1)
    clk = ws[len(ws) // 2][1] if ws else "?"
    ms = (t1 - t0) / n * 1e3
    j = pw * ms * 1e-3
    print(f"{name:64s} {ms:9.4f} {pw:6.0f} {clk:>5s} {j:9.3f} {(pw - idle_w) * ms * 1e-3:10.3f} {per_clip:8d} {j * per_clip:7.2f}")
    tot += j * per_clip; tot_ms += ms * per_clip
print(f"sum over the step's kernels: {tot:.2f} J and {tot_ms:.2f} ms of launches per 8-frame clip (the step itself: ~4.7 ms on two stream lanes at 1.27-1.37 kW = ~6.2 J)")
