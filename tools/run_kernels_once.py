"""A few launches each of the roofline kernels at the bench shapes (cfg2: 8 x 480x854 -> 120x214x256), for rocprofv3 --pmc passes
(tools/pmc_report.py turns the result directories into profiles/rNN_pmc.json).  Run the interpreter directly after `--`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgvc_amd import engine, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8; HW = H * W
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
h16_all = ops.split_f16x2(feats)
h6_all = ops.split_f16f6p(feats)      # round 4: the rows of fgvc_pair_topk_f16f6
h6x_all = ops.split_f16f6x(feats)     # round 5: the 2 KiB rows of the default bank (fgvc_pair_topk_f16f6x + the refining merge)
cfg6 = engine.TrackerConfig(pair_split_fmt="f16f6", pair_precision="split")
hl = ops.split_bf16(feats[:2])
sp = ops.split_f16f8(feats[:2])
sp6 = ops.split_f16f6(feats[:2])
vol = torch.empty((HW, HW), device=dev)
for _ in range(3):
    pl = engine.run_pairs(h6x_all, H, W, plan, cfg6)          # the default: fgvc_pair_topk_f16f6x on 2 KiB rows ...
    engine.merge_pairs(pl, cfg6)                              # ... + fgvc_merge_refine_topk_f32 (merge_mark / refine / refine_scan kernels)
    ops.pair_topk_split(h6_all, h6_all, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
    ops.set_option("pair_f16_debug", 4194304)                 # round 6: the one-role kernel (opt-in), for its fetch / busy figures beside v7's
    ops.pair_topk_split(h6_all, h6_all, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6")
    ops.set_option("pair_f16_debug", 0)
    ops.pair_topk_split(h16_all, h16_all, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16")
    ops.pair_topk(feats, feats, pairs, H, W, H, W, cfg.mask, 10, validate=False)
    ops.corr_volume(sp6[1], sp6[0], 0.07, "f16f6", out=vol)
    ops.corr_volume(sp[1], sp[0], 0.07, "f16f8", out=vol)
    ops.corr_volume(hl[1], hl[0], 0.07, "bf16x3", out=vol)
    ops.corr_volume(hl[1], hl[0], 0.07, "bf16", out=vol)
    ops.corr_volume(feats[1], feats[0], 0.07, "f32", out=vol)
torch.cuda.synchronize()
# one 256 -> 256 3x3 convolution at the layer-3 size: the f16f8 form (round 3's default: ARITH 1 in the kernel's name) and the bf16x3 form
wt = torch.randn(256, 256, 3, 3, device=dev) * 0.02
bn = torch.nn.BatchNorm2d(256).eval().to(dev)
wp, bs = ops.prepare_conv_split(wt, bn)
xs = ops.nchw_to_split_nhwc(torch.relu(torch.randn(T, 256, H, W, device=dev)))
ys = ops.alloc_split_nhwc(T, 256, H, W, dev)
wp8, bs8, sw8 = ops.prepare_conv_split_f16(wt, bn, ops.ACT_F16F8)
wp6, bs6, sw6 = ops.prepare_conv_split_f16(wt, bn, ops.ACT_F16F6)        # round 4's default arithmetic: f16 + block-scaled FP6
xs6 = ops.alloc_split_nhwc(T, 256, H, W, dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
ops.conv_split(xs, wp, bs, H, W, True, out_split=xs6, out_fmt=ops.ACT_F16F6, out_scale_log2=4, overflow=ovf)     # a genuine f16f6 input tensor
for _ in range(3):
    ops.conv_split(xs, wp, bs, H, W, True, out_split=ys)
    ops.conv_split(xs, wp8, bs8, H, W, True, out_split=ys, in_fmt=ops.ACT_F16F8, in_scale_log2=sw8, out_fmt=ops.ACT_F16F8, out_scale_log2=0, overflow=ovf)
    ops.conv_split(xs6, wp6, bs6, H, W, True, out_split=ys, in_fmt=ops.ACT_F16F6, in_scale_log2=4 + sw6, out_fmt=ops.ACT_F16F6, out_scale_log2=4, overflow=ovf)
# ... the two special forms of the same layer (round 4): with layer 3's 1 x 1 projection shortcut in its sums, and the last one writing the feature bank
wt2 = torch.randn(256, 128, 1, 1, device=dev) * 0.05
x2 = ops.alloc_split_nhwc(T, 128, H, W, dev)
wq0, bq0 = ops.prepare_conv_split(torch.randn(128, 128, 3, 3, device=dev) * 0.03, torch.nn.BatchNorm2d(128).eval().to(dev))
ops.conv_split(ops.nchw_to_split_nhwc(torch.relu(torch.randn(T, 128, H, W, device=dev))), wq0, bq0, H, W, True, out_split=x2, out_fmt=ops.ACT_F16F6, out_scale_log2=4, overflow=ovf)
wp2, bs2, _ = ops.prepare_conv_split_f16(wt2, torch.nn.BatchNorm2d(256).eval().to(dev), ops.ACT_F16F6, force_exp=sw6)     # s_x2 s_w2 = s_x s_w (both inputs at 2^4)
bank = torch.empty((T, H * W, 4, 256), dtype=torch.int16, device=dev)      # round 5: split_f16f6x rows (+ the f32 channels)
idt = ops.alloc_nhwc(T, 256, H, W, dev); idt.normal_()
for _ in range(3):
    ops.conv_split(xs6, wp6, bs6 + bs2, H, W, True, out_split=ys, out_f32=idt, in_fmt=ops.ACT_F16F6, in_scale_log2=4 + sw6, out_fmt=ops.ACT_F16F6, out_scale_log2=4,
                   overflow=ovf, x2_split=x2, w2=wp2)
    ops.conv_split_to_bank(xs6, wp6, bs6, H, W, True, bank, residual=idt, in_fmt=ops.ACT_F16F6, in_scale_log2=4 + sw6)
torch.cuda.synchronize()
# layer 1 (64 -> 64, register-resident weights), the stem and the stride-2 block of layer 2 at the 480p clip's sizes
frames = torch.randn(T, 3, 480, 854, device=dev)
sw, sb = ops.prepare_stem7(torch.randn(64, 3, 7, 7, device=dev) * 0.1, torch.nn.BatchNorm2d(64).eval().to(dev))
st_s, st_f = ops.alloc_split_nhwc(T, 64, 240, 427, dev), ops.alloc_nhwc(T, 64, 240, 427, dev)
w64, b64 = ops.prepare_conv64(torch.randn(64, 64, 3, 3, device=dev) * 0.05, torch.nn.BatchNorm2d(64).eval().to(dev))
y64 = ops.alloc_split_nhwc(T, 64, 240, 427, dev)
w2, b2 = ops.prepare_conv_s2(torch.randn(128, 64, 3, 3, device=dev) * 0.05, torch.nn.BatchNorm2d(128).eval().to(dev))
s2_out = ops.alloc_split_nhwc(T, 128, 120, 214, dev)
# (round 5) layer 1 as the encoder runs it: f16 + fp8 operands -- a block's first convolution (split out) and its second (f32 residual,
# f32 + split out) on conv64p_kernel
w64f, b64f, sw64 = ops.prepare_conv64_f16(torch.randn(64, 64, 3, 3, device=dev) * 0.05, torch.nn.BatchNorm2d(64).eval().to(dev))
st8 = ops.alloc_split_nhwc(T, 64, 240, 427, dev)
y64f = ops.alloc_nhwc(T, 64, 240, 427, dev)
for _ in range(3):
    ops.stem7_split(frames, sw, sb, True, out_split=st_s, out_f32=st_f)
    ops.conv64_split(st_s, w64, b64, 240, 427, True, out_split=y64)
    ops.conv_s2_split(st_s, w2, b2, 240, 427, True, out_split=s2_out)
    ops.stem7_split(frames, sw, sb, True, out_split=st8, out_f32=st_f, out_fmt=ops.ACT_F16F8, out_scale_log2=4, overflow=ovf)
    ops.conv64_split(st8, w64f, b64f, 240, 427, True, out_split=y64, in_fmt=ops.ACT_F16F8, in_scale_log2=4 + sw64, out_fmt=ops.ACT_F16F8, out_scale_log2=4,
                     overflow=ovf)
    ops.conv64_split(st8, w64f, b64f, 240, 427, True, residual=st_f, out_split=y64, out_f32=y64f, in_fmt=ops.ACT_F16F8, in_scale_log2=4 + sw64,
                     out_fmt=ops.ACT_F16F8, out_scale_log2=4, overflow=ovf)
torch.cuda.synchronize()
