"""Round 5's hand-laid instruction streams -- conv64p_kernel (layer 1, f16 + fp8) and conv256p_kernel (the 128 / 256-channel 3 x 3 layers,
f16 + FP6: main loop one generated assembly statement) -- against the kernels they replace (conv64_kernel<1>: option conv64_variant = 16;
conv_split_kernel: option conv_debug = 1024).  Same arithmetic AND the same accumulation order, so every output is BIT-identical: split
rows with their zero borders, f32 rows, feature-bank rows, overflow flags.  Sizes: partial tile columns, row counts that are no multiple
of a tile, fewer tiles than workgroups and several per workgroup, 1 / 2 / 4 / 8 input chunks.  (The replaced kernels are themselves held
to float64 in test_gpu_parity.py; the encoder-level tests there run on the new ones by default.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fgvc_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _pack_f16f8(x, sx, dev):
    from fgvc_amd import ops
    N, C, H, W = x.shape
    xs = ops.alloc_split_nhwc(N, C, H, W, dev)
    v = (x.permute(0, 2, 3, 1) * 2.0 ** sx).reshape(N, H, W, C // 32, 32).contiguous()
    hh = v.to(torch.float16)
    l8 = ((v - hh.float()) * 2.0 ** ops.F8_BX).to(torch.float8_e4m3fn).view(torch.uint8)
    h8 = (hh.float() * 2.0 ** -ops.F8_AX).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    xs[:, 1:H + 1, 1:W + 1] = torch.cat([hh.view(torch.uint8), l8, h8], -1).contiguous().view(torch.int16)
    return xs


def _pack_f16f6(x, sx, dev):
    from fgvc_amd import ops
    from oracle import fgvc_oracle as O
    N, C, H, W = x.shape
    v = x.permute(0, 2, 3, 1).float().reshape(-1, 32).cpu().numpy()
    row = torch.from_numpy(O.act_f16f6_rows(v, sx)).reshape(N, H, W, C // 32, 128)
    out = ops.alloc_split_nhwc(N, C, H, W, dev)
    out[:, 1:H + 1, 1:W + 1] = row.contiguous().view(torch.int16).to(dev)
    return out


@pytest.mark.parametrize("shape", [(1, 4, 32), (1, 3, 5), (2, 21, 50), (3, 64, 96), (2, 120, 214)])
def test_conv64p_is_conv64_bit_for_bit(dev, shape):
    from fgvc_amd import ops
    N, H, W = shape
    F8, BF = ops.ACT_F16F8, ops.ACT_BF16X2
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).to(dev)
    bn = torch.nn.BatchNorm2d(64).eval().to(dev)
    bn.bias.data = torch.randn(64, generator=g).to(dev) * 0.1
    w1, b1, sw = ops.prepare_conv64_f16(wt, bn)
    x = (torch.randn(N, 64, H, W, generator=g).abs() ** 1.5).to(dev)
    sx = ops.act_scale_log2(float(x.abs().max()))
    xs = _pack_f16f8(x, sx, dev)
    r_f = torch.randn(N, H, W, 64, generator=g).to(dev)
    try:
        for (res, f32, fmt) in [(False, False, F8), (True, True, F8), (True, False, F8), (True, False, BF)]:
            for relu, so in ((True, 3), (False, 3), (True, 16)):            # (scale 2^16: the overflow flag must rise in both)
                got = []
                for variant in (0, 16):
                    ops.set_option("conv64_variant", variant)
                    o_s = ops.alloc_split_nhwc(N, 64, H, W, dev)
                    o_f = ops.alloc_nhwc(N, 64, H, W, dev) if f32 else None
                    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
                    ops.conv64_split(xs, w1, b1, H, W, relu, residual=r_f if res else None, out_split=o_s, out_f32=o_f, in_fmt=F8, in_scale_log2=sx + sw,
                                     out_fmt=fmt, out_scale_log2=so, overflow=ovf)
                    got.append((o_s, o_f, int(ovf.item())))
                a, b = got
                assert torch.equal(a[0], b[0]) and a[2] == b[2], (shape, res, f32, fmt, relu, so)
                assert a[1] is None or torch.equal(a[1], b[1])
                if so == 16 and fmt == F8:
                    assert a[2] == 1
    finally:
        ops.set_option("conv64_variant", 0)


def _conv_operands(N, Cin, Cout, H, W, seed, dev, ks=3):
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(seed)
    wt = (torch.randn(Cout, Cin, ks, ks, generator=g) * (0.03 if ks == 3 else 0.08)).to(dev)
    bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
    bn.bias.data = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    x = (torch.randn(N, Cin, H, W, generator=g).abs() ** 1.3).to(dev)
    return wt, bn, x


@pytest.mark.parametrize("shape", [(1, 256, 256, 8, 32), (1, 32, 256, 5, 7), (2, 256, 256, 21, 50), (1, 128, 256, 33, 70), (1, 256, 256, 120, 214),
                                   (1, 128, 128, 8, 32), (2, 128, 128, 21, 50), (1, 64, 128, 33, 70), (1, 128, 128, 120, 214), (1, 128, 512, 16, 40)])
def test_conv256p_is_conv_split_bit_for_bit(dev, shape):
    from fgvc_amd import ops
    N, Cin, Cout, H, W = shape
    F6 = ops.ACT_F16F6
    wt, bn, x = _conv_operands(N, Cin, Cout, H, W, sum(shape), dev)
    wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, F6)
    sx = ops.act_scale_log2(float(x.abs().max()))
    xs = _pack_f16f6(x, sx, dev)
    res = torch.randn(N, H, W, Cout, device=dev)
    try:
        for (r, f32, fmt, so, relu) in [(None, False, F6, 4, True), (res, True, F6, 4, True), (res, True, ops.ACT_BF16X2, 0, False), (None, True, F6, 16, True),
                                        (None, True, ops.ACT_F16F8, 4, True)]:
            got = []
            for dbg in (0, 1024):
                ops.set_option("conv_debug", dbg)
                o_s = ops.alloc_split_nhwc(N, Cout, H, W, dev)
                o_f = ops.alloc_nhwc(N, Cout, H, W, dev) if f32 else None
                ovf = torch.zeros(1, dtype=torch.int32, device=dev)
                ops.conv_split(xs, wp, bias, H, W, relu, residual=r, out_split=o_s, out_f32=o_f, in_fmt=F6, in_scale_log2=sx + sw, out_fmt=fmt,
                               out_scale_log2=so, overflow=ovf)
                got.append((o_s, o_f, int(ovf.item())))
            a, b = got
            assert torch.equal(a[0], b[0]) and a[2] == b[2], (shape, f32, fmt, so)
            assert a[1] is None or torch.equal(a[1], b[1])
    finally:
        ops.set_option("conv_debug", 0)


@pytest.mark.parametrize("shape", [(1, 256, 128, 8, 32), (2, 256, 128, 21, 50), (1, 64, 32, 9, 40), (1, 256, 96, 33, 70), (1, 256, 128, 120, 214)])
def test_conv256p_second_input_and_bank(dev, shape):
    """the folded 1 x 1 projection (fgvc_conv_split_proj_fmt_f32) and the feature-bank epilogue (fgvc_conv_split_bank_f16f6x_f32) on the new main loop"""
    from fgvc_amd import ops
    N, Cin, Cin2, H, W = shape
    F6 = ops.ACT_F16F6
    wt, bn, x = _conv_operands(N, Cin, 256, H, W, sum(shape), dev)
    wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, F6)
    sx = ops.act_scale_log2(float(x.abs().max()))
    xs = _pack_f16f6(x, sx, dev)
    wt2, bn2, x2 = _conv_operands(N, Cin2, 256, H, W, sum(shape) + 1, dev, ks=1)
    x2 = x2 * 3.0
    sx2 = ops.act_scale_log2(float(x2.abs().max()))
    xs2 = _pack_f16f6(x2, sx2, dev)
    wp2, bias2, sw2 = ops.prepare_conv_split_f16(wt2, bn2, F6, force_exp=sx + sw - sx2)
    res = torch.randn(N, H, W, 256, device=dev)
    try:
        got = []
        for dbg in (0, 1024):
            ops.set_option("conv_debug", dbg)
            o_s, o_f = ops.alloc_split_nhwc(N, 256, H, W, dev), ops.alloc_nhwc(N, 256, H, W, dev)
            ovf = torch.zeros(1, dtype=torch.int32, device=dev)
            ops.conv_split(xs, wp, bias + bias2, H, W, True, out_split=o_s, out_f32=o_f, in_fmt=F6, in_scale_log2=sx + sw, out_fmt=F6, out_scale_log2=4,
                           overflow=ovf, x2_split=xs2, w2=wp2)
            bank = torch.zeros(N, H * W, 4, 256, dtype=torch.int16, device=dev)
            ops.conv_split_to_bank(xs, wp, bias, H, W, True, bank, residual=res, in_fmt=F6, in_scale_log2=sx + sw)
            got.append((o_s, o_f, int(ovf.item()), bank))
        a, b = got
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
        assert torch.equal(a[3], b[3])
    finally:
        ops.set_option("conv_debug", 0)


def test_stream_kernels_soak(dev):
    """200 launches of each stream kernel on the same operands, other work in flight on a second stream: every output bit-identical to the
    first launch's (their waits are hand-counted: a wait that is one short shows up as a rare wrong tile, not as a failing parity test)."""
    from fgvc_amd import ops
    F6, F8 = ops.ACT_F16F6, ops.ACT_F16F8
    N, H, W = 2, 120, 214
    wt, bn, x = _conv_operands(N, 256, 256, H, W, 77, dev)
    wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, F6)
    sx = ops.act_scale_log2(float(x.abs().max()))
    xs = _pack_f16f6(x, sx, dev)
    res = torch.randn(N, H, W, 256, device=dev)
    g = torch.Generator().manual_seed(5)
    wt64 = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).to(dev)
    w64, b64, sw64 = ops.prepare_conv64_f16(wt64, torch.nn.BatchNorm2d(64).eval().to(dev))
    x64 = (torch.randn(2, 64, 240, 427, generator=g).abs() ** 1.5).to(dev)
    sx64 = ops.act_scale_log2(float(x64.abs().max()))
    xs64 = _pack_f16f8(x64, sx64, dev)
    r64 = torch.randn(2, 240, 427, 64, generator=g).to(dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    noise = torch.randn(4096, 4096, device=dev)

    def launch():
        o_s, o_f = ops.alloc_split_nhwc(N, 256, H, W, dev), ops.alloc_nhwc(N, 256, H, W, dev)
        ops.conv_split(xs, wp, bias, H, W, True, residual=res, out_split=o_s, out_f32=o_f, in_fmt=F6, in_scale_log2=sx + sw, out_fmt=F6, out_scale_log2=4,
                       overflow=ovf)
        y_s, y_f = ops.alloc_split_nhwc(2, 64, 240, 427, dev), ops.alloc_nhwc(2, 64, 240, 427, dev)
        ops.conv64_split(xs64, w64, b64, 240, 427, True, residual=r64, out_split=y_s, out_f32=y_f, in_fmt=F8, in_scale_log2=sx64 + sw64, out_fmt=F8,
                         out_scale_log2=3, overflow=ovf)
        return o_s, o_f, y_s, y_f
    first = launch()
    torch.cuda.synchronize()
    for it in range(200):
        with torch.cuda.stream(side):
            noise = noise @ noise * 1e-4                    # memory and matrix traffic beside the kernels under test
        got = launch()
        if it % 20 == 19:
            torch.cuda.synchronize()
            for a, b in zip(first, got):
                assert torch.equal(a, b), it
    torch.cuda.synchronize()
    assert int(ovf.item()) == 0
