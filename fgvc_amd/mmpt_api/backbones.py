"""ResNet trunk behind the reference's registry name and constructor keys
(mmpt/models/backbones/resnet.py:329-638).  On the GPU in eval mode the whole ResNet-18 trunk of the shipped configs
runs as hand-written HIP on the bf16 pipe (hi/lo bf16 split, f32-grade, BatchNorm folded, identity and ReLU in the
epilogues): fgvc_stem7_split_f32 (7x7 stride-2 stem), fgvc_conv64_split_f32 / fgvc_conv_split_f32 (stride-1 3x3),
fgvc_conv_s2_split_f32 (stride-2 3x3 and 1x1 projection).  Everything outside that path (pooling stems, other block
types, `ResNet.use_split_conv = False`) runs in MIOpen with the BN(eval) [+ residual] [+ ReLU] tail as one hand-written
launch (fgvc_bn_act_f32).

What must match the reference: the state_dict key names (mmcv ConvModule nesting:
`conv1.conv.weight`, `layer2.0.downsample.bn.running_var`, ...), the strides/out_indices/pool_type
semantics and the arithmetic (conv -> BN(eval) -> ReLU).  What deliberately differs: forward()
stops after the last requested stage (the reference runs all four stages even when only stage 2 is
returned, resnet.py:619-627 -- ~17 GFLOP/frame of dead work at 256x256).
"""
from __future__ import annotations

import re
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import BACKBONES

_CAPTURING = [False]      # set while ResNet.forward_hwc captures a HIP graph: allocations / evictions / device syncs then raise


class ConvBN(nn.Module):
    """conv(no bias) + BatchNorm2d [+ ReLU]; attribute names follow mmcv.cnn.ConvModule."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, relu=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, dilation, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.relu = relu

    def forward(self, x, residual=None, relu=None):
        """conv -> BN [-> + residual] [-> ReLU].  On the GPU in eval mode the BN / add / ReLU tail is ONE
        fgvc_bn_act_f32 launch (in place on the conv output); otherwise plain torch modules."""
        relu = self.relu if relu is None else relu
        y = self.conv(x)
        if y.is_cuda and not self.training and y.dtype == torch.float32 and y.is_contiguous():
            from .. import ops
            return ops.bn_act(y, self.bn, residual, relu, inplace=True)
        y = self.bn(y)
        if residual is not None:
            y = y + residual
        return F.relu(y, inplace=True) if relu else y


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = ConvBN(cin, planes, 3, stride, dilation, dilation, relu=True)
        self.conv2 = ConvBN(planes, planes, 3, 1, 1, 1, relu=False)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        return self.conv2(self.conv1(x), residual=idt, relu=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = ConvBN(cin, planes, 1, 1, 0, 1, relu=True)
        self.conv2 = ConvBN(planes, planes, 3, stride, dilation, dilation, relu=True)
        self.conv3 = ConvBN(planes, planes * 4, 1, 1, 0, 1, relu=False)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        return self.conv3(self.conv2(self.conv1(x)), residual=idt, relu=True)


def _stage(block, cin, planes, n, stride, dilation):
    ds = None
    if stride != 1 or cin != planes * block.expansion:
        ds = ConvBN(cin, planes * block.expansion, 1, stride, 0, 1, relu=False)
    first_dil = dilation if dilation == 1 else dilation // 2           # resnet.py:304
    layers = [block(cin, planes, stride, first_dil, ds)]
    layers += [block(planes * block.expansion, planes, 1, dilation) for _ in range(1, n)]
    return nn.Sequential(*layers)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, pretrained=None, torchvision_pretrain=True, in_channels=3, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(3,), planes_custom=None,
                 pool_type="max", zero_init_residual=True, **unused):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f"invalid depth {depth} for resnet")
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages
        self.depth, self.pretrained, self.torchvision_pretrain = depth, pretrained, torchvision_pretrain
        self.strides, self.dilations, self.out_indices = tuple(strides), tuple(dilations), tuple(out_indices)
        self.zero_init_residual = zero_init_residual
        block, counts = self.arch_settings[depth]
        self.conv1 = ConvBN(in_channels, 64, 7, 2, 3, 1, relu=True)                 # resnet.py:457-466
        self.pool = {"max": nn.MaxPool2d(3, 2, 1), "mean": nn.AvgPool2d(3, 2, 1)}.get(pool_type)
        cin = 64
        self.res_layers = []
        for i, n in enumerate(counts[:num_stages]):
            planes = planes_custom[i] if planes_custom is not None else 64 * 2 ** i
            self.add_module(f"layer{i + 1}", _stage(block, cin, planes, n, strides[i], dilations[i]))
            self.res_layers.append(f"layer{i + 1}")
            cin = planes * block.expansion
        self.feat_dim = cin

    def init_weights(self):
        """resnet.py:566-601: `pretrained` (a path or a state dict) is a TORCHVISION checkpoint when `torchvision_pretrain` (the
        constructor default) -- its keys are remapped onto the ConvModule nesting (:525-563) -- else one of the reference's own
        (prefixes stripped like revise_keys at :580; a dict there raises, :590); random init (:592-601) when `pretrained` is None."""
        self.reset_split_cache()
        if isinstance(self.pretrained, (str, dict)):
            if self.torchvision_pretrain:
                load_torchvision_checkpoint(self, self.pretrained)
            elif isinstance(self.pretrained, str):
                load_checkpoint(self, self.pretrained)
            else:
                raise Exception("a state dict is only accepted as a torchvision checkpoint (resnet.py:590)")
            return
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.conv2.bn.weight, 0)
                elif isinstance(m, Bottleneck):
                    nn.init.constant_(m.conv3.bn.weight, 0)

    # ---- stages on the bf16 matrix pipe (fgvc_conv_split_f32) ------------------------------------------------------
    use_conv64 = True              # 64 -> 64 3x3 layers on fgvc_conv64_split_f32 (register-resident weights)
    use_stem7 = True               # 7x7 stride-2 stem on fgvc_stem7_split_f32 (False: MIOpen f32 + ReLU/split pass)
    use_s2_conv = True             # stride-2 blocks on fgvc_conv_s2_split_f32 (False: MIOpen f32 for the two strided convolutions)
    conv64_f16f8 = True            # with arith "f16f6" / "f16f8" layer 1 (fgvc_conv64_split_fmt_f32) computes in the f16 + fp8 form (a third fewer
                                   # matrix passes than bf16x3; layer 1 has no FP6 form) and the stem writes that form.  Round 3 measured nothing
                                   # in the step (4.99-5.01 against 4.95-5.00 ms); round 4, with the rest of the step lighter: 4.60 -> 4.52 ms
                                   # (1737 -> 1768 frames/s, same box, A/B/A/B), trunk error unchanged (1.8e-5 of the largest feature): ON.
                                   # (False: layer 1 in bf16x3, as the "bf16x3" / "f16x3" trunks always run it)
    res_from_split = False         # True: layer 1 adds its identities from the split form and nobody writes f32 copies of them (round 4: the
                                   # conversion deferred to the epilogue: the same step time as f32 identities, 840 MB less traffic; round 3:
                                   # fewer bytes, but 0.08 ms per clip SLOWER (8-byte loads + conversions in the owner wave's MFMA stream;
                                   # profiles/r03_bv_xcd.log)); same precision (tools/experiments/res_split_precision.py)
    use_split_conv = True          # class-level switch (tests / A-B timing): False = every convolution through MIOpen
    arith = "f16f6"                # arithmetic of the stride-1 convolutions behind layer 1: see set_arith() (round 3: "f16f8")
    fold_projection = True         # a block's stride-1 1 x 1 projection shortcut rides in the sums of the block's second convolution
                                   # (fgvc_conv_split_proj_fmt_f32: 256 output channels, 3 x 3; not in the f16f8 arithmetic, whose e4m3
                                   # forms cannot take a forced weight scale): no projection launch, no dense f32 identity (False: A/B)
    layer1_whole_batch = True      # the stem and layer 1 once over the whole batch, the stream lanes fork behind them (see _trunk)
    fuse_bank = True               # the trunk's last convolution writes the pair kernel's feature bank itself (fgvc_conv_split_bank_f16f6p_f32)
                                   # when the caller asks for f16f6 rows of a 256-channel stage: no dense f32 output, no normalise pass;
                                   # byte-identical rows (False: the two-kernel route, kept for A/B and as the reference of the tests)

    @staticmethod
    def supported_arith():
        return ("f16f6", "f16f8", "bf16x3", "f16x3")

    def set_arith(self, arith: str):
        """Arithmetic of the stride-1 convolutions of fgvc_conv_split_f32 (layers 2 and 3 of ResNet-18: 80 % of the trunk's FLOPs) on the
        16-bit matrix pipe; every form accumulates in f32 (csrc/conv_split.hip):
          "bf16x3"  (hi, lo) bf16 operands, hi*hi + hi*lo + lo*hi: three pipe units per product, ~2^-17 per term (round 2);
          "f16f8"   h = f16(s x) + e4m3 forms of h and of its residual: the main product on the f16 form, both cross sums in one K-64
                    fp8 MFMA: two units, ~2^-15.5 per term (round 3's default: 1e-4 logit on the final features, tools/sim_conv_formats.py);
          "f16f6"   (round 4) as "f16f8" with the cross sums on block-scaled FP6 (e2m3) operands -- one E8M0 scale per pixel (or output
                    channel) and 32-channel chunk, made by the producing kernel's epilogue -- which the same K-64 instruction retires
                    in half the cycles: 1.5 units; error as "f16f8" (tools/sim_conv_formats.py: 1.40e-5 against 1.26e-5 of the features);
                    the default: 256 -> 256 layer 0.43 against 0.48 ms, encode phase 3.41 against 3.62 ms per 8-frame 480p clip;
          "f16x3"   (h, l) f16 operands, three units, ~2^-22 per term.
        The f16 forms store s x with a per-tensor power-of-two scale s, calibrated on the first batch a set of weights sees
        (`calibrate`), with 2^7-2^8 of headroom; a value beyond the f16 range raises a device flag that `check_overflow` turns into an
        error (and a re-calibration).  The stride-2 convolutions keep the bf16 form; layer 1 and the stem's output run in f16 + fp8 under the two
        mixed arithmetics (`conv64_f16f8`, default on since round 4; False keeps them in bf16x3)."""
        if arith not in self.supported_arith():
            raise ValueError(f"arith={arith!r}: one of {self.supported_arith()}")
        if arith != self.arith:
            self.arith = arith
            self.reset_split_cache()

    # ---- formats and scales of the split activation tensors ------------------------------------------------------------------------
    def _generic_s1(self, cb) -> bool:
        """Does this ConvBN run on fgvc_conv_split_f32 (stride 1, not the register-resident 64 -> 64 kernel)?"""
        c = cb.conv
        return c.stride == (1, 1) and not (self.use_conv64 and tuple(c.weight.shape) == (64, 64, 3, 3))

    def _is_conv64(self, cb) -> bool:
        c = cb.conv
        return bool(self.use_conv64 and c.stride == (1, 1) and tuple(c.weight.shape) == (64, 64, 3, 3))

    def _stem7_ok(self) -> bool:
        """The stem runs on fgvc_stem7_split_fmt_f32 (7x7 / stride 2 / pad 3, 3 -> 64 channels, ReLU) rather than in MIOpen."""
        c1 = self.conv1.conv
        return bool(self.use_stem7 and c1.kernel_size == (7, 7) and c1.stride == (2, 2) and c1.padding == (3, 3) and c1.in_channels == 3
                    and c1.out_channels == 64 and c1.dilation == (1, 1) and self.conv1.relu)

    def _format_plan(self, last: int):
        """{(stage, block): (format of the block's input, of conv1's output `a`, of the block's split output `y`), "stem": format of the
        stem's split output} as ops.ACT_* codes.  A tensor takes an f16 form iff every convolution that reads it can (the stride-1
        kernel fgvc_conv_split_fmt_f32: the trunk's arithmetic; the register-resident 64 -> 64 kernel fgvc_conv64_split_fmt_f32: the
        f16 + fp8 form only; the stride-2 kernels and MIOpen: bf16) and its producer can write it (the stride-1 and stride-2 kernels
        and the stem: any form; the 64 -> 64 kernel: bf16 or f16 + fp8).  A convolution computes in the format of its input (its
        weights are laid out to match)."""
        from .. import ops
        fmt, bf = ops.ACT_FMT[self.arith], ops.ACT_BF16X2
        cache = self.__dict__.setdefault("_split_cache", {})
        key = ("fmt_plan", self.arith, last, self.use_conv64, self.use_s2_conv, self.use_stem7, self.conv64_f16f8)
        if key not in cache:
            blocks = [(s_i, b_i, blk) for s_i in range(last + 1) for b_i, blk in enumerate(getattr(self, self.res_layers[s_i]))]
            f64 = ops.ACT_F16F8 if (fmt in (ops.ACT_F16F8, ops.ACT_F16F6) and self.conv64_f16f8) else bf      # (layer 1 has no FP6 form: f16 + fp8 under either)

            def reads(cb):                 # the format this convolution takes its input in
                if self._is_conv64(cb):
                    return f64
                return fmt if self._generic_s1(cb) else bf

            def can_write(cb, f):
                if f == bf:
                    return True
                if self._is_conv64(cb):
                    return f == ops.ACT_F16F8
                return self._generic_s1(cb) or (cb.conv.stride == (2, 2) and self.use_s2_conv)

            def wants(blk):
                f = reads(blk.conv1)
                return f if (blk.downsample is None or reads(blk.downsample) == f) else bf
            plan = {}
            f_in = wants(blocks[0][2]) if blocks else bf
            if f_in != bf and not (self._stem7_ok() and f_in == ops.ACT_F16F8):   # (MIOpen stem + fgvc_nhwc_to_split_f32: the bf16 form)
                f_in = bf
            plan["stem"] = f_in
            for i, (s_i, b_i, blk) in enumerate(blocks):
                f_a = reads(blk.conv2) if can_write(blk.conv1, reads(blk.conv2)) else bf
                nxt = blocks[i + 1][2] if i + 1 < len(blocks) else None
                f_y = wants(nxt) if (nxt is not None and can_write(blk.conv2, wants(nxt))) else bf
                plan[(s_i, b_i)] = (f_in, f_a, f_y)
                f_in = f_y
            cache[key] = plan
        return cache[key]

    def _block_formats(self, si: int, bi: int, last: int):
        return self._format_plan(last)[(si, bi)]


    def _projection_fold(self, blk, wt, si, bi, dev, f_in, f_a, s_in, s_a, Cout, banked):
        """(w2, bias) for fgvc_conv_split_proj_fmt_f32 -- the block's 1 x 1 stride-1 projection as extra stages of its second
        convolution -- or None when the fold does not apply: other shapes / kernels, the f16f8 arithmetic (fixed e4m3 scales), the
        bank-writing form of the last convolution, a forced weight scale s_w2 = s_a s_w / s_x2 that would leave the f16 range."""
        from .. import ops
        ds, c2 = blk.downsample, blk.conv2
        if not (self.fold_projection and not banked and f_in == f_a and f_a != ops.ACT_F16F8 and wt["c2"][3] == "s1" and wt["ds"] is not None
                and wt["ds"][3] == "s1" and Cout == 256 and c2.conv.kernel_size == (3, 3) and ds.conv.kernel_size == (1, 1)
                and ds.conv.stride == (1, 1) and ds.conv.in_channels % 32 == 0):
            return None
        cache = self.__dict__.setdefault("_split_cache", {})
        e = s_a + wt["c2"][2] - s_in                                       # log2 of the projection's weight scale
        key = ("wproj", si, bi, dev, self.arith, f_a, e)
        if key not in cache:
            w = ds.conv.weight.detach()
            if f_a == ops.ACT_BF16X2:
                w2, b2 = ops.prepare_conv_split(w, ds.bn)
            else:
                sc = (ds.bn.weight / torch.sqrt(ds.bn.running_var + ds.bn.eps)).detach().float().abs().view(-1, 1, 1, 1)
                top = float((w.detach().float().abs() * sc).max()) * 2.0 ** e
                if not (2.0 ** -6 <= top < 2.0 ** 14) and top != 0.0:        # keep the folded weights well inside f16's normal range
                    cache[key] = None
                    return None
                w2, b2, _ = ops.prepare_conv_split_f16(w, ds.bn, f_a, force_exp=e)
            cache[key] = (w2, (wt["c2"][1] + b2).contiguous())
        return cache[key]

    def _identity_from_split(self, blk) -> bool:
        """Whether `blk` adds its identity from the SPLIT form of its input (hi + lo, the value its first convolution multiplies)
        instead of a dense f32 copy: the 64-channel blocks of layer 1 on fgvc_conv64_split_res_f32 -- their kernels are bound by the
        bytes they move, and the producer of the input then writes 4 bytes per value instead of 8."""
        c2 = getattr(blk, "conv2", None)
        return bool(self.use_conv64 and self.res_from_split and not (self.arith in ("f16f8", "f16f6") and self.conv64_f16f8)      # (hi, lo) bf16 tensors only
                    and isinstance(blk, BasicBlock) and blk.downsample is None
                    and c2 is not None and tuple(c2.conv.weight.shape) == (64, 64, 3, 3) and c2.conv.stride == (1, 1)
                    and tuple(blk.conv1.conv.weight.shape) == (64, 64, 3, 3) and blk.conv1.conv.stride == (1, 1))

    def _scales(self, dev):
        cache = self.__dict__.setdefault("_split_cache", {})
        return cache.get(("scales", dev))

    # Where the f16 scales come from (round 5).  "canonical" (default): from ONE fixed input -- two seeded 192 x 192 frames of smooth
    # structure + noise inside the range of mean / std-normalised images -- run through THESE weights when they are first used, with
    # 2^8 of headroom: a function of the weights alone.  model(B) is then the same bits whether or not model(A) ran first, and the ranks
    # of a data-parallel job (each with its own first video) hold ONE set of scales.  (Round 4 calibrated on the first batch the weights
    # saw and kept it: later videos inherited the first one's scales -- the same results bit for bit wherever no value falls below
    # 2^-22 of its tensor's maximum, but not by construction.)  "first_batch": round 4's rule.  `calibrate(frames)` pins the scales to
    # frames of the caller's choice under either rule.
    calibration = "canonical"
    CANON_TARGET_LOG2 = 8          # the canonical input's largest activation sits at (2^7, 2^8] of the f16 range (top 2^16): 2^8 of headroom over an
                                   # input that spans the whole range of normalised images.  (Tried 2^6: layer 1 computes in f16 + fp8, whose e4m3
                                   # cross terms carry FIXED scales built for tensors near 2^8 -- two binades lower they reach e4m3's subnormals:
                                   # 1.8e-3 px on the 4 x 64 x 64 fixture, a k-th place flipped in the HR tracker's coordinate field)

    @staticmethod
    def canonical_frames(device):
        g = torch.Generator().manual_seed(0x5CA1E)
        base = torch.randn(2, 3, 24, 24, generator=g)
        x = F.interpolate(base, size=(192, 192), mode="bilinear", align_corners=False) * 1.5 + 0.4 * torch.randn(2, 3, 192, 192, generator=g)
        return x.clamp_(-2.7, 2.7).to(device)        # (mean / std-normalised 8-bit images lie in [-2.2, 2.7]; standard deviation ~1)

    def calibrate(self, x=None, last=None, device=None):
        """Per-tensor scales of the f16-format activation tensors from one batch: the trunk runs once in the bf16 form (which needs no
        scales), the largest magnitude of every split tensor is read back, and s = 2^(target - ceil(log2(max))).  `x` = None: the
        canonical frames (target 2^6; what the first forward of a set of weights does under `calibration = "canonical"`); frames of
        your own: target 2^8, and the scales stay until the weights change, an overflow, or the next calibrate()."""
        from .. import ops
        cache = self.__dict__.setdefault("_split_cache", {})
        canonical = x is None
        if canonical:
            x = self.canonical_frames(device if device is not None else next(self.parameters()).device)
        dev = x.device
        saved = (self.arith, self.split_lanes)
        rec = {}
        self.__dict__["_calib"] = rec
        try:
            self.arith, self.split_lanes = "bf16x3", 1
            self._trunk(x, max(self.out_indices) if last is None else last)
            torch.cuda.synchronize(dev)
        finally:
            self.arith, self.split_lanes = saved
            self.__dict__.pop("_calib", None)
        # `_headroom_extra`: binades taken off every scale after an overflow, for the retry of that one video (end_overflow_retry)
        target = (self.CANON_TARGET_LOG2 if canonical else ops.F16_TARGET_LOG2) - int(self.__dict__.get("_headroom_extra", 0))
        scales = {k: ops.act_scale_log2(float(v), target) for k, v in rec.items()}
        cache[("scales", dev)] = scales
        cache[("scales_from", dev)] = "canonical" if canonical else "frames"
        self._drop_graphs(dev)                             # captured graphs carry the OLD scales as kernel arguments
        if ("overflow", dev) not in cache:
            cache[("overflow", dev)] = torch.zeros(1, dtype=torch.int32, device=dev)
        self._cache_filled(dev)
        return scales

    def check_overflow(self):
        """True iff a value left the f16 range of its calibrated tensor since the last check (the results since then are invalid):
        reads and clears the device flag (a synchronisation) and drops the scales, so that the next forward re-calibrates."""
        cache = self.__dict__.get("_split_cache", {})
        hit = False
        for k in [k for k in cache if isinstance(k, tuple) and k and k[0] == "overflow"]:
            if int(cache[k].item()) != 0:
                hit = True
                cache[k].zero_()
                cache.pop(("scales", k[1]), None)
                self._drop_graphs(k[1])                    # ... and so do the graphs captured with them
        if hit and self.calibration == "canonical":
            # the retry of THIS video runs 2^4 further below the f16 top (the same rule on every rank of a job); the video after it
            # starts from the canonical scales again (end_overflow_retry): no video's result depends on the videos before it
            self.__dict__["_headroom_extra"] = min(int(self.__dict__.get("_headroom_extra", 0)) + 4, 12)
        return hit

    def end_overflow_retry(self):
        """Back to the canonical scales after the video that overflowed them has been run again (BaseModel.forward calls this)."""
        if self.__dict__.pop("_headroom_extra", 0):
            cache = self.__dict__.get("_split_cache", {})
            for k in [k for k in cache if isinstance(k, tuple) and k and k[0] == "scales"]:
                del cache[k]
                self._drop_graphs(k[1])

    def _drop_graphs(self, dev=None, shape_sig=None):
        """Forget captured HIP graphs (all, those of one device, or those of one input shape (N, h, w, device)): a graph bakes in the
        scales it was captured with and the addresses of the workspaces of its shape."""
        cache = self.__dict__.get("_split_cache", {})
        for k in [k for k in cache if isinstance(k, tuple) and k and k[0] == "graph"]:
            if dev is not None and k[2] != dev:
                continue
            if shape_sig is not None and not (k[1][0] == shape_sig[0] and tuple(k[1][2:]) == tuple(shape_sig[1:3]) and k[2] == shape_sig[3]):
                continue
            del cache[k]

    def _load_from_state_dict(self, *args, **kwargs):
        # called for this module by ANY load_state_dict (its own or a parent model's): folded / split weights are derived
        # from the parameters and must be rebuilt
        self.__dict__.pop("_split_cache", None)
        return super()._load_from_state_dict(*args, **kwargs)

    def reset_split_cache(self):
        """Drop the cached folded / split weights and workspaces (call after changing parameters in place)."""
        self.__dict__.pop("_split_cache", None)

    def _split_stage_ok(self, stage, x) -> bool:
        """A stage runs on the bf16 pipe when it is made of BasicBlocks with dilation-1 convolutions, Cin % 32 == 0 and
        Cout % 64 == 0, in eval mode on the GPU in f32.  Only the stage's first block may be strided: stride 2 goes to
        fgvc_conv_s2_split_f32 (3x3 and 1x1 projection), any other stride to MIOpen for those two convolutions only."""
        if not (self.use_split_conv and x.is_cuda and not self.training and x.dtype == torch.float32):
            return False
        for bi, blk in enumerate(stage):
            if not isinstance(blk, BasicBlock):
                return False
            convs = [blk.conv1.conv, blk.conv2.conv] + ([blk.downsample.conv] if blk.downsample is not None else [])
            for c in convs:
                if (c.dilation != (1, 1) or c.groups != 1 or c.in_channels % 32 or c.out_channels % 64
                        or c.kernel_size not in ((1, 1), (3, 3))):
                    return False
            strided = blk.conv1.conv.stride != (1, 1)
            if blk.conv2.conv.stride != (1, 1) or (strided and (bi > 0 or blk.downsample is None)):
                return False
            if blk.downsample is not None and blk.downsample.conv.stride != blk.conv1.conv.stride:
                return False
            if blk.downsample is None and blk.conv1.conv.in_channels != blk.conv2.conv.out_channels:
                return False
        return True

    def _split_buffers(self, key, N, C, H, W, device, names):
        """Workspaces, allocated once per (stage, block, shape) and reused by every call: "s_*" = zero-bordered padded
        split NHWC, "f_*" = dense NHWC f32."""
        from .. import ops
        cache = self.__dict__.setdefault("_split_cache", {})
        k = ("b", self.__dict__.get("_ws_sig")) + key + (N, C, H, W, device)
        mk = {"s": ops.alloc_split_nhwc, "f": ops.alloc_nhwc}
        if k not in cache:
            cache[k] = {}
        missing = [nm for nm in names if nm not in cache[k]]      # (the calibration pass and the real one may ask for different sets)
        if missing:
            for nm in missing:
                cache[k][nm] = mk[nm[0]](N, C, H, W, device)
            self._cache_filled(device)
        return cache[k]

    max_workspace_shapes = 2       # padded activation workspaces are kept for this many distinct input shapes (N, h, w); older
                                   # ones are dropped (a variable-resolution dataset would otherwise grow HBM use without bound)

    def _touch_workspace_shape(self, sig):
        """LRU over input shapes: the workspaces of `_split_buffers` are keyed by the input shape that made them."""
        cache = self.__dict__.setdefault("_split_cache", {})
        order = cache.setdefault("_ws_order", [])
        if sig in order:
            order.remove(sig)
        order.append(sig)
        while len(order) > max(1, int(self.max_workspace_shapes)):
            old = order.pop(0)
            if _CAPTURING[0]:
                raise RuntimeError("ResNet: a workspace eviction inside a HIP-graph capture (the graph would keep freed addresses)")
            if isinstance(sig[-1], torch.device) and sig[-1].type == "cuda":
                torch.cuda.synchronize(sig[-1])                      # nothing in flight may still read the evicted buffers
            self._drop_graphs(shape_sig=old)                         # a graph of that shape replays into the evicted buffers
            for k in [k for k in cache if isinstance(k, tuple) and len(k) > 1 and k[0] == "b" and k[1] == old]:
                del cache[k]
        self.__dict__["_ws_sig"] = sig

    @staticmethod
    def _cache_filled(device):
        """Cache entries (workspaces with zeroed borders, folded / split weights) are produced on whichever lane's stream
        asks first and then used by every lane: make them visible to all streams once, when they are made."""
        if _CAPTURING[0]:
            raise RuntimeError("ResNet: a cache entry was made inside a HIP-graph capture (the capture call must find everything in place)")
        torch.cuda.synchronize(device)

    @staticmethod
    def _fold(cb: "ConvBN"):
        """conv weight and bias with the eval-mode BatchNorm folded in, channels_last (for MIOpen's NHWC kernels)."""
        bn = cb.bn
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
        w = (cb.conv.weight.detach() * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
        return w, (bn.bias - bn.running_mean * scale).detach().contiguous()

    def _miopen_nhwc(self, key, cb: "ConvBN", x_cl):
        """conv + folded BN in MIOpen on a channels_last input; returns the dense NHWC f32 result (N,H,W,C) (a view of the
        channels_last output: no layout conversion on either side)."""
        cache = self.__dict__.setdefault("_split_cache", {})
        k = ("m",) + key + (x_cl.device,)
        if k not in cache:
            cache[k] = self._fold(cb)
            self._cache_filled(x_cl.device)
        w, b = cache[k]
        c = cb.conv
        y = F.conv2d(x_cl, w, b, c.stride, c.padding, c.dilation, c.groups)
        y = y.contiguous(memory_format=torch.channels_last)          # already is, for MIOpen's NHWC solvers
        return y.permute(0, 2, 3, 1)

    split_lanes = 2                # batch slices run on this many HIP streams at once (1 = everything on the caller's stream);
                                   # equal slices matter: 8 frames as 4+4 beat 2+3+3 once the persistent kernels came in

    def _stage_split(self, si: int, cur, call):
        """Run stage `si` for the batch slice [lo, hi) of an N-image batch.  `cur` = dict(split = padded split NHWC input,
        f32 = dense NHWC f32 of the same tensor, H, W, lo, hi, N, need_split, need_f32); returns the same for the stage output plus
        `full` = the whole-batch dense NHWC f32 buffer the slice was written into.  `call` = per-call state shared by the
        lanes: main stream, lane streams, the stages whose output the caller receives (`fresh`: those are allocated per
        call -- a cached workspace would be overwritten by the next forward) and the buffers allocated so far."""
        from .. import ops
        stage = getattr(self, self.res_layers[si])
        cache = self.__dict__.setdefault("_split_cache", {})
        dev = cur["split"].device
        lo, hi, N = cur["lo"], cur["hi"], cur["N"]
        from ..ops import ACT_BF16X2
        last_stage = call["last"]
        fmts = [self._block_formats(si, bi, last_stage) for bi in range(len(stage))]
        calib = self.__dict__.get("_calib")
        scales = self._scales(dev) or {}
        ovf = cache.get(("overflow", dev))
        wkey = ("w", si, dev, self.arith)
        if wkey not in cache:
            def prep_s1(w, bn, fmt):      # 64 -> 64 3x3: the register-resident-weights kernel (its own weight orders)
                if self.use_conv64 and tuple(w.shape) == (64, 64, 3, 3):
                    if fmt == ops.ACT_F16F8:
                        return ops.prepare_conv64_f16(w, bn) + ("conv64",)
                    assert fmt == ACT_BF16X2
                    return ops.prepare_conv64(w, bn) + (0, "conv64")
                if fmt == ACT_BF16X2:
                    return ops.prepare_conv_split(w, bn) + (0, "s1")
                return ops.prepare_conv_split_f16(w, bn, fmt) + ("s1",)

            def prep(cb, fmt):            # -> (weights, bias, log2 s_w) or None (MIOpen)
                st = cb.conv.stride
                if st == (1, 1):
                    return prep_s1(cb.conv.weight.detach(), cb.bn, fmt)
                if st == (2, 2):
                    return ops.prepare_conv_s2(cb.conv.weight.detach(), cb.bn) + (0, "s2")
                return None
            cache[wkey] = [dict(c1=prep(b.conv1, f[0]), c2=prep(b.conv2, f[1]),
                                ds=None if b.downsample is None else prep(b.downsample, f[0]))
                           for b, f in zip(stage, fmts)]
            self._cache_filled(dev)
        full = None
        for bi, (blk, wt) in enumerate(zip(stage, cache[wkey])):
            Cout = blk.conv2.conv.out_channels
            idt_split = None
            H, W = cur["H"], cur["W"]
            f_in, f_a, f_y = fmts[bi]
            s_in = cur.get("scale", 0)
            s_a = scales.get((si, bi, "a"), 0) if f_a != ACT_BF16X2 else 0
            s_y = scales.get((si, bi, "y"), 0) if f_y != ACT_BF16X2 else 0
            assert cur.get("fmt", ACT_BF16X2) == f_in, "split activation format mismatch between producer and consumer"

            def conv_s1(x_split, wb, in_fmt=ACT_BF16X2, in_scale=0, out_fmt=ACT_BF16X2, out_scale=0, **kw):
                """stride-1 convolution by whichever kernel the weights were laid out for"""
                if wb[3] == "conv64":
                    ops.conv64_split(x_split, wb[0], wb[1], H, W, in_fmt=in_fmt, in_scale_log2=in_scale + wb[2], out_fmt=out_fmt,
                                     out_scale_log2=out_scale, overflow=ovf, **kw)
                else:
                    ops.conv_split(x_split, wb[0], wb[1], H, W, in_fmt=in_fmt, in_scale_log2=in_scale + wb[2], out_fmt=out_fmt,
                                   out_scale_log2=out_scale, overflow=ovf, **kw)

            last_conv = bi == len(stage) - 1 and not cur["need_split"]        # nobody reads the split form of the trunk output
            proj = None
            x_in_split = cur["split"]
            if blk.conv1.conv.stride == (2, 2) and self.use_s2_conv:
                # stride-2 3x3 and stride-2 projection on the bf16 pipe as well (fgvc_conv_s2_split_f32)
                Hi, Wi = H, W
                H, W = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
                bufs = self._split_buffers((si, bi), N, Cout, H, W, dev, ("s_a", "s_y", "f_y", "f_idt"))
                buf = {k: v[lo:hi] for k, v in bufs.items()}
                ops.conv_s2_split(cur["split"], wt["ds"][0], wt["ds"][1], Hi, Wi, relu=False, out_f32=buf["f_idt"])
                idt = buf["f_idt"]
                ops.conv_s2_split(cur["split"], wt["c1"][0], wt["c1"][1], Hi, Wi, relu=True, out_split=buf["s_a"], out_fmt=f_a,
                                  out_scale_log2=s_a, overflow=ovf)
            elif blk.conv1.conv.stride != (1, 1):
                # other strides: strided 3x3 and strided projection in MIOpen, NHWC in and out (the dense f32 tensors ARE
                # channels_last tensors), then back onto the bf16 pipe: ReLU + split in one pass
                x_cl = cur["f32"].permute(0, 3, 1, 2)
                t1 = self._miopen_nhwc((si, bi, "c1"), blk.conv1, x_cl)
                idt = self._miopen_nhwc((si, bi, "ds"), blk.downsample, x_cl)
                _, H, W, _ = t1.shape
                bufs = self._split_buffers((si, bi), N, Cout, H, W, dev, ("s_a", "s_y", "f_y"))
                buf = {k: v[lo:hi] for k, v in bufs.items()}
                assert f_a == ACT_BF16X2                                   # (fgvc_nhwc_to_split_f32 writes the bf16 form)
                ops.nhwc_to_split(t1, buf["s_a"], relu=True)
            else:
                bufs = self._split_buffers((si, bi), N, Cout, H, W, dev, ("s_a", "s_y", "f_y", "f_idt"))
                buf = {k: v[lo:hi] for k, v in bufs.items()}
                if blk.downsample is not None:
                    proj = self._projection_fold(blk, wt, si, bi, dev, f_in, f_a, s_in, s_a, Cout, last_conv and si == call["last"] and call.get("bank_of") is not None)
                if proj is not None:
                    idt = None                                             # (w2, summed bias): the projection rides in conv2's sums
                elif blk.downsample is not None:
                    conv_s1(cur["split"], wt["ds"], f_in, s_in, relu=False, out_f32=buf["f_idt"])
                    idt = buf["f_idt"]
                elif self._identity_from_split(blk):
                    idt, idt_split = None, cur["split"]                    # hi + lo of the block's input: no f32 copy of it exists
                else:
                    idt = cur["f32"]
                    assert idt is not None
                conv_s1(cur["split"], wt["c1"], f_in, s_in, f_a, s_a, relu=True, out_split=buf["s_a"])
            if calib is not None:
                calib[(si, bi, "a")] = torch.maximum(calib.get((si, bi, "a"), torch.zeros((), device=dev)),
                                                     buf["s_a"].view(torch.bfloat16)[..., :32].abs().amax().float())
            full = bufs["f_y"]
            if bi == len(stage) - 1 and si in call["fresh"]:
                if si not in call["out"]:
                    with torch.cuda.stream(call["main"]):                # owned by the caller's stream, like any result
                        call["out"][si] = ops.alloc_nhwc(N, Cout, H, W, dev)
                    for st in call["streams"]:
                        if st is not call["main"]:
                            st.wait_stream(call["main"])                 # the block's previous life ended on that stream
                full = call["out"][si]
            f_y_ = full[lo:hi]
            # nobody reads the f32 form of this output: the stage's last block when the next stage takes the split form, any other
            # block whose successor takes its identity from the split form
            skip_f32 = (not cur["need_f32"]) if bi == len(stage) - 1 else self._identity_from_split(stage[bi + 1])
            if skip_f32:
                f_y_ = None
            kw_res = dict(residual_split=idt_split) if idt_split is not None else dict(residual=idt)
            bank = None
            if (last_conv and si == call["last"] and call.get("bank_of") is not None and wt["c2"][3] == "s1" and Cout == 256
                    and blk.conv2.conv.kernel_size == (3, 3) and idt_split is None and calib is None):
                bank = call["bank_of"](lo, hi, Cout, H, W)              # the caller's rows for these frames, or None
            if bank is not None:
                ops.conv_split_to_bank(buf["s_a"], wt["c2"][0], wt["c2"][1], H, W, True, bank, residual=idt, in_fmt=f_a,
                                       in_scale_log2=s_a + wt["c2"][2], normalize=call["bank_normalize"])
                cur = dict(cur, split=None, f32=None, H=H, W=W, fmt=f_y, scale=s_y, banked=True)
                continue
            if proj is not None:
                ops.conv_split(buf["s_a"], wt["c2"][0], proj[1], H, W, True, out_split=None if last_conv else buf["s_y"], out_f32=f_y_,
                               in_fmt=f_a, in_scale_log2=s_a + wt["c2"][2], out_fmt=f_y, out_scale_log2=s_y, overflow=ovf,
                               x2_split=x_in_split, w2=proj[0])
            else:
                conv_s1(buf["s_a"], wt["c2"], f_a, s_a, f_y, s_y, relu=True, out_split=None if last_conv else buf["s_y"], out_f32=f_y_, **kw_res)
            if calib is not None and not last_conv:
                calib[(si, bi, "y")] = torch.maximum(calib.get((si, bi, "y"), torch.zeros((), device=dev)),
                                                     buf["s_y"].view(torch.bfloat16)[..., :32].abs().amax().float())
            cur = dict(cur, split=buf["s_y"], f32=f_y_, H=H, W=W, fmt=f_y, scale=s_y)
        cur["full"] = full
        return cur

    def _trunk(self, x, last: int, fresh=(), post=None, bank_of=None, bank_normalize=True):
        """Stem and stages 0..last.  When every requested stage qualifies (and there is no pooling layer) the whole trunk
        stays in NHWC on the bf16 pipe: the 7x7 stride-2 stem in fgvc_stem7_split_f32 (any other stem: MIOpen on
        channels_last tensors with BatchNorm folded, ReLU fused into the split), the stages as _stage_split runs them.
        The batch is cut into `split_lanes` slices that run on separate HIP streams: a layer's launch covers the 256 CUs
        3.3 times at 8 frames, and the other lane's workgroups fill the tail of each launch.
        Stage outputs listed in `fresh` are new tensors, the others views of cached workspaces (valid until the next call).
        `post(y_slice, lo, hi, C, H, W)`, if given, runs at the end of every lane on that lane's stream with the lane's slice of the
        last stage's dense NHWC output (NHWC trunk only: check the returned flag).  `bank_of(lo, hi, C, H, W)`, if given, may return
        the caller's f16f6 feature rows for the lane's frames: the stage's last convolution then writes them itself
        (fgvc_conv_split_bank_f16f6p_f32: normalised iff `bank_normalize`), the stage's dense output is NOT produced and `post` is skipped.
        Returns (list of NCHW outputs of stages < last, last stage output, NHWC flag, H, W)."""
        from .. import ops
        stages = [getattr(self, nm) for nm in self.res_layers[:last + 1]]
        probe = x
        if (self.pool is None and x.is_cuda and x.dtype == torch.float32 and self.conv1.conv.out_channels % 32 == 0
                and all(self._split_stage_ok(st, probe) for st in stages)):
            N, dev = x.shape[0], x.device
            cache = self.__dict__.setdefault("_split_cache", {})
            if self.arith != "bf16x3" and self._scales(dev) is None and "_calib" not in self.__dict__:
                # first use of these weights: one pass in the bf16 form fixes the f16 scales -- of the canonical frames (a function of the
                # weights alone), or of this batch (`calibration = "first_batch"`, round 4's rule)
                self.calibrate(None if self.calibration == "canonical" else x, last, device=dev)
            self._touch_workspace_shape((N, x.shape[2], x.shape[3], dev))
            # (one or two frames: one lane -- two streams of half-empty launches gain nothing there and cost the fork / join:
            # 1 047 -> 1 375 frames/s on the 2 x 256 x 256 workload, profiles/r04_other_workloads.log)
            n_lanes = 1 if N <= 2 else max(1, min(int(self.split_lanes), N))
            main = torch.cuda.current_stream(dev)
            c1 = self.conv1.conv
            stem7 = None
            f_stem = self._format_plan(last)["stem"]
            s_stem = (self._scales(dev) or {}).get(("stem",), 0) if f_stem != ops.ACT_BF16X2 else 0
            calib = self.__dict__.get("_calib")
            if self._stem7_ok():
                if ("stem7", dev) not in cache:
                    cache[("stem7", dev)] = ops.prepare_stem7(c1.weight.detach(), self.conv1.bn)
                    self._cache_filled(dev)
                stem7 = cache[("stem7", dev)]
                x = x.contiguous()
            # on the caller's stream, BEFORE the lanes wait for it
            x_cl = None if stem7 else x.contiguous(memory_format=torch.channels_last)
            if n_lanes > 1:
                skey = ("streams", dev, n_lanes)
                if skey not in cache:
                    cache[skey] = [torch.cuda.Stream(dev, priority=int(getattr(self, "lane_priority", 0))) for _ in range(n_lanes)]
                streams = cache[skey]
                for s in streams:
                    s.wait_stream(main)
            else:
                streams = [main]
            # Layer 1 runs on fgvc_conv64_split_fmt_f32 -- one wave per SIMD owns the whole register file (the weights live there), so two
            # lanes' launches cannot share a CU: they take turns, and each pays its own prologue (147 KB of weights per workgroup) and its
            # own ragged last round of tiles.  With `layer1_whole_batch` the stem and layer 1 therefore run ONCE over the whole batch on
            # the caller's stream and the lanes fork behind them (measured: bench.py --no-layer1-whole-batch).
            whole = bool(n_lanes > 1 and self.layer1_whole_batch and last >= 1 and calib is None and 0 not in fresh
                         and all(self._is_conv64(b.conv1) and self._is_conv64(b.conv2) for b in stages[0]))
            lane_streams = [main] if whole else streams
            lanes = []
            for li, s in enumerate(lane_streams):
                lo, hi = (0, N) if whole else (li * N // n_lanes, (li + 1) * N // n_lanes)
                with torch.cuda.stream(s):
                    if stem7:                                                   # stem on the bf16 pipe, from the NCHW frames
                        H, W, C0 = (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1, 64
                        if self._identity_from_split(stages[0][0]):             # nobody reads an f32 copy of the stem's output
                            sb = self._split_buffers(("stem",), N, C0, H, W, dev, ("s_x",))
                            t = None
                        else:
                            sb = self._split_buffers(("stem",), N, C0, H, W, dev, ("s_x", "f_x"))
                            t = sb["f_x"][lo:hi]
                        ops.stem7_split(x[lo:hi], stem7[0], stem7[1], True, out_split=sb["s_x"][lo:hi], out_f32=t, out_fmt=f_stem,
                                        out_scale_log2=s_stem, overflow=cache.get(("overflow", dev)))
                    else:
                        assert f_stem == ops.ACT_BF16X2
                        t = self._miopen_nhwc(("stem",), self.conv1, x_cl[lo:hi])   # (n,H,W,64)
                        _, H, W, C0 = t.shape
                        sb = self._split_buffers(("stem",), N, C0, H, W, dev, ("s_x",))
                        ops.nhwc_to_split(t, sb["s_x"][lo:hi], relu=True)          # ReLU in place on t + split
                    if calib is not None:                                        # (the calibration pass runs in the bf16 form)
                        calib[("stem",)] = torch.maximum(calib.get(("stem",), torch.zeros((), device=dev)),
                                                         sb["s_x"][lo:hi].view(torch.bfloat16)[..., :32].abs().amax().float())
                lanes.append(dict(split=sb["s_x"][lo:hi], f32=t, H=H, W=W, lo=lo, hi=hi, N=N, need_split=True, fmt=f_stem, scale=s_stem))
            fulls = []
            call = dict(main=main, streams=streams, fresh=tuple(fresh), out={}, last=last, bank_of=bank_of, bank_normalize=bool(bank_normalize))
            for i in range(last + 1):
                need_f32 = True           # the stage output in f32: returned, or the next block's identity / MIOpen input
                if i < last and i not in call["fresh"]:
                    nb = stages[i + 1][0]
                    st = nb.conv1.conv.stride
                    need_f32 = nb.downsample is None or not (st == (1, 1) or (st == (2, 2) and self.use_s2_conv))
                if whole and i == 0:
                    one = self._stage_split(0, dict(lanes[0], need_split=True, need_f32=need_f32), call)     # the whole batch, on `main`
                    fulls.append(one["full"])
                    for s in streams:
                        s.wait_stream(main)
                    lanes = []
                    for li in range(n_lanes):                                    # ... and the lanes take their slices of its output
                        lo, hi = li * N // n_lanes, (li + 1) * N // n_lanes
                        lanes.append(dict(one, split=one["split"][lo:hi], f32=None if one["f32"] is None else one["f32"][lo:hi], lo=lo, hi=hi))
                    continue
                for li, s in enumerate(streams):                                 # lanes interleaved stage by stage
                    with torch.cuda.stream(s):
                        lanes[li] = self._stage_split(i, dict(lanes[li], need_split=i < last, need_f32=need_f32), call)
                fulls.append(lanes[0]["full"])
            if post is not None:
                for ln, s in zip(lanes, streams):
                    if ln.get("banked"):                                         # the last convolution wrote this lane's rows itself
                        continue
                    with torch.cuda.stream(s):
                        post(ln["f32"], ln["lo"], ln["hi"], ln["f32"].shape[-1], ln["H"], ln["W"])
            if n_lanes > 1:
                for s in streams:
                    main.wait_stream(s)
            outs = [f.permute(0, 3, 1, 2) for f in fulls]                        # channels_last NCHW views
            return outs[:-1], fulls[-1], True, lanes[0]["H"], lanes[0]["W"]
        x = self.conv1(x)
        if self.pool is not None:
            x = self.pool(x)
        outs = []
        for st in stages:
            x = st(x)
            outs.append(x)
        return outs[:-1], x, False, x.shape[-2], x.shape[-1]

    def forward(self, x, out_idx=None):
        want = tuple(out_idx) if out_idx is not None else self.out_indices
        last = max(want)                               # later stages cannot influence the outputs
        earlier, y, nhwc, H, W = self._trunk(x, last, fresh=want)
        if nhwc:
            y = y.permute(0, 3, 1, 2)                      # channels_last NCHW view of the dense NHWC buffer
        stage_out = earlier + [y]
        outs = [stage_out[i] for i in want]
        return outs[0] if len(outs) == 1 else tuple(outs)

    use_graph = "auto"             # True: forward_hwc replays a HIP graph of the whole NHWC trunk per input shape (captured on the second
                                   # call of a shape); "auto" (default, round 4): for inputs of at most `graph_auto_max_px` pixels
                                   # (N x h x w) only -- HOST-bound calls: a 2-frame 256 x 256 call is ~32 launches at ~25 us of Python
                                   # each against 0.3 ms of GPU work (1 047 -> 2 001 frames/s on that workload); an 8-frame 480p clip is
                                   # GPU-bound and gains nothing; False: never.
    graph_auto_max_px = 300_000

    def forward_hwc(self, x, normalize: bool = True, split_if=None, split_fmt: str = "bf16", out=None):
        """forward_hwc_eager, or -- with `use_graph` on the GPU in eval mode -- the same work replayed from a HIP graph (torch.cuda.CUDAGraph
        on ROCm).  The graph of a shape is captured on its second call (the first runs eagerly: calibration, workspaces, weight
        layouts), reads a static copy of the input and writes a static output: the returned tensor is overwritten by the next call
        with the same shape, so the caller gets a COPY (`out` when given, else a fresh tensor).  Graphs live in the split cache: new
        weights, a re-calibration, an overflow (`check_overflow`) and the eviction of the shape's workspaces drop them."""
        want = self.use_graph is True or (self.use_graph == "auto" and x.dim() == 4 and x.shape[0] * x.shape[2] * x.shape[3] <= self.graph_auto_max_px)
        if not (want and x.is_cuda and not self.training and x.dtype == torch.float32) or torch.cuda.is_current_stream_capturing():
            return self.forward_hwc_eager(x, normalize, split_if, split_fmt, out)
        cache = self.__dict__.setdefault("_split_cache", {})
        # (everything a captured pass bakes in besides the input: the class-level switches tests and A/B runs flip between calls)
        sig = (self.arith, self.split_lanes, self.use_conv64, self.use_stem7, self.use_s2_conv, self.conv64_f16f8, self.res_from_split,
               self.use_split_conv, self.fuse_bank, self.fold_projection, self.layer1_whole_batch, tuple(self.out_indices))
        key = ("graph", tuple(x.shape), x.device, bool(normalize), split_fmt, split_if is not None, sig)
        ent = cache.get(key)
        if ent is None:                                        # first call of this shape: eager (it may calibrate and allocate)
            cache[key] = "warm"
            return self.forward_hwc_eager(x, normalize, split_if, split_fmt, out)
        if ent == "warm":
            if self._scales(x.device) is None and self.arith != "bf16x3":
                return self.forward_hwc_eager(x, normalize, split_if, split_fmt, out)      # (scales were dropped: calibrate eagerly first)
            static_in = torch.empty_like(x)
            static_in.copy_(x)
            # one more eager pass right in front of the capture: this shape's workspaces exist and are the most recently used ones, so
            # the captured pass neither allocates nor evicts (both synchronise, and an evicted buffer's address would stay in the graph)
            self.forward_hwc_eager(static_in, normalize, split_if, split_fmt, None)
            torch.cuda.synchronize(x.device)
            g = torch.cuda.CUDAGraph()
            _CAPTURING[0] = True
            try:
                with torch.cuda.graph(g):
                    o, H, W = self.forward_hwc_eager(static_in, normalize, split_if, split_fmt, None)
            finally:
                _CAPTURING[0] = False
            ent = cache[key] = (g, static_in, o, H, W)
        g, static_in, o, H, W = ent
        static_in.copy_(x)
        g.replay()
        if out is not None and tuple(out.shape) == tuple(o.shape) and out.dtype == o.dtype:
            out.copy_(o)
            return out, H, W
        return o.clone(), H, W                                 # `o` is the graph's static output: the next replay overwrites it

    def forward_hwc_eager(self, x, normalize: bool = True, split_if=None, split_fmt: str = "bf16", out=None):
        """The tracker's fast path: features of the single requested stage as (N, H*W, C) f32 rows, L2-normalised if
        `normalize` -- straight from the dense NHWC buffer when the stage ran on the bf16 pipe (no NCHW round trip).
        `split_if(C, H, W) -> bool`: when given and true for the stage's shape, the rows come back as their (hi, lo) bf16
        split (N, H*W, 2, C) int16 instead (what the split pair kernel reads; `split_fmt` "bf16" = ops.split_bf16, "f16" =
        ops.split_f16x2), made in the same pass.  Returns (feats, H, W)."""
        from .. import ops
        assert len(self.out_indices) == 1
        box = {}

        def ensure_out(C, H, W):
            if "out" not in box:
                as_split = bool(split_if is not None and split_if(C, H, W))
                shape = (x.shape[0], H * W, 4 if split_fmt == "f16f6x" else 2, C) if as_split else (x.shape[0], H * W, C)
                dtype = torch.int16 if as_split else torch.float32
                if out is not None and tuple(out.shape) == shape and out.dtype == dtype and out.is_contiguous() and out.device == x.device:
                    box["out"] = out                                           # the caller's rows (a slice of its feature bank): no copy later
                else:
                    with torch.cuda.stream(main):                              # the result is the caller's: its stream owns it
                        box["out"] = torch.empty(shape, device=x.device, dtype=dtype)
                torch.cuda.current_stream(x.device).wait_stream(main)          # (the block's previous life ended on that stream)
                box["split"] = as_split
            elif torch.cuda.current_stream(x.device) != main:
                torch.cuda.current_stream(x.device).wait_stream(main)

        def post(y_slice, lo, hi, C, H, W):      # each lane normalises (and splits) its own frames under the other lane's tail
            ensure_out(C, H, W)
            ops.normalize_nhwc(y_slice, normalize, split=split_fmt if box["split"] else False, out=box["out"][lo:hi])

        def bank_of(lo, hi, C, H, W):            # ... or hands its rows to the trunk's last convolution (f16f6 rows of 256 channels)
            if not (self.fuse_bank and split_fmt in ("f16f6", "f16f6x") and C == 256 and split_if is not None and split_if(C, H, W)):
                return None
            ensure_out(C, H, W)
            return box["out"][lo:hi] if box["split"] else None

        main = torch.cuda.current_stream(x.device) if x.is_cuda else None
        _, y, nhwc, H, W = self._trunk(x, self.out_indices[0], post=post, bank_of=bank_of, bank_normalize=normalize)
        if nhwc:
            return box["out"], H, W
        C = y.shape[1]
        as_split = bool(split_if is not None and split_if(C, H, W))
        f = ops.normalize_to_hwc(y.float(), normalize, pad=True)
        splitter = {"f16": ops.split_f16x2, "f16f6": ops.split_f16f6p, "f16f6x": ops.split_f16f6x}.get(split_fmt, ops.split_bf16)
        return (splitter(f) if as_split and f.shape[-1] == C else f), H, W


def torchvision_key(name: str) -> str:
    """Own state_dict key -> the torchvision ResNet key it is filled from (resnet.py:540-556): a ConvModule `X` holds `X.conv.*`
    and `X.bn.*`; torchvision has `layerN.M.convK.weight` / `layerN.M.bnK.*` and `layerN.M.downsample.0.* / .1.*`."""
    mod, _, leaf = name.rpartition(".")               # e.g. "layer2.0.downsample.bn", "running_var"
    owner, _, part = mod.rpartition(".")              # "layer2.0.downsample", "bn"
    if part not in ("conv", "bn"):
        return name
    if "downsample" in owner:
        return f"{owner}.{0 if part == 'conv' else 1}.{leaf}"
    return f"{owner if part == 'conv' else owner.replace('conv', 'bn')}.{leaf}"


def load_torchvision_checkpoint(module: nn.Module, filename_or_state):
    """resnet.py:525-563: copy a torchvision ResNet checkpoint into the ConvModule-nested parameters; keys of the checkpoint that
    have no counterpart (fc.*, stages that were not built) are ignored, buffers absent from it (num_batches_tracked) keep their
    values.  Returns the checkpoint keys that were not used."""
    sd = torch.load(filename_or_state, map_location="cpu") if isinstance(filename_or_state, str) else filename_or_state
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    own = module.state_dict()
    used = set()
    with torch.no_grad():
        for k, v in own.items():
            tv = torchvision_key(k)
            if tv in sd:
                v.copy_(sd[tv])
                used.add(tv)
            elif not k.endswith("num_batches_tracked"):
                raise KeyError(f"torchvision checkpoint has no '{tv}' (for '{k}')")     # the reference indexes it unguarded
    if hasattr(module, "reset_split_cache"):
        module.reset_split_cache()
    return sorted(set(sd) - used)


_PREFIXES = (r"^module\.", r"^backbone\.", r"^encoder\.", r"^backbone_fine\.")


def load_checkpoint(module: nn.Module, filename_or_state, strict: bool = False, prefixes: Sequence[str] = _PREFIXES):
    """Minimal stand-in for mmcv.runner.load_checkpoint: accepts a path or a state dict, unwraps
    'state_dict', strips the prefixes the reference strips (resnet.py:580), loads non-strictly."""
    sd = torch.load(filename_or_state, map_location="cpu") if isinstance(filename_or_state, str) else filename_or_state
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    own = module.state_dict()
    out = {}
    for k, v in sd.items():
        kk = k
        if kk not in own:
            for p in prefixes:
                k2 = re.sub(p, "", kk)
                if k2 != kk and (k2 in own or not strict):
                    kk = k2
                    if kk in own:
                        break
        out[kk] = v
    return module.load_state_dict({k: v for k, v in out.items() if k in own} if not strict else out, strict=strict)
