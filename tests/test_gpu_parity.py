"""GPU parity: the HIP path (through the C ABI) against the oracle and against the golden
vectors recorded from the reference.  Run on an MI355X with `pytest -m gpu`.

Bars (BASELINE.json north_star): affinity indices bit-exact (modulo the documented tie policy: exact on every
query whose top-(k+1) ranks are separated by more than 1e-5 in the float64 oracle, legitimate-within-tolerance
otherwise), float scores within 1e-3.
"""
import numpy as np
import pytest
import torch

from oracle import fgvc_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fgvc_amd import _lib
    _lib.load()          # the HIP library must be the thing under test: fail loudly if it is missing
    return torch.device("cuda:0")


def hwc(x):  # (C,H,W) or (n,C,H,W) cpu -> normalised channels-last on cpu (oracle-side layout twin)
    if x.dim() == 3:
        x = x.unsqueeze(0)
    x = O.l2_normalize(x, 1)
    return x.flatten(2).transpose(1, 2).contiguous()


def gpu_affinity(dev, q, key, topk, temperature, neighbor_range, mask_mode="circle", non_mask_len=0,
                 mode="softmax"):
    """query (C,H,W), key (C,T,H,W) -> idx/logit/weight (HW,k) via normalize -> pair_topk -> merge."""
    from fgvc_amd import ops
    C, H, W = q.shape
    Tn = key.shape[1]
    frames = torch.cat([q.unsqueeze(0), key.permute(1, 0, 2, 3)], 0).to(dev)
    feats = ops.normalize_to_hwc(frames, pad=True)
    mask = ops.MaskSpec.from_neighbor_range(neighbor_range, mask_mode)
    pairs = ops.make_pairs([(0, 1 + t, (t >= non_mask_len) and not mask.is_none) for t in range(Tn)], dev)
    pidx, pscore = ops.pair_topk(feats, feats, pairs, H, W, H, W, mask, topk)
    slot_pair = torch.arange(Tn, dtype=torch.int32, device=dev).view(1, Tn)
    idx, logit, weight = ops.merge_topk(pidx, pscore, slot_pair, H * W, topk, temperature, mode)
    return feats, idx[0], logit[0], weight[0]


def dense64(q, key, temperature, neighbor_range, mask_mode="circle", non_mask_len=0):
    C, H, W = q.shape
    Tn = key.shape[1]
    vol = O.corr_volume(q.double(), key.double(), temperature)
    if neighbor_range is not None:
        m = O.mask_slab(H, W, H, W, Tn, torch.arange(H * W), neighbor_range, mask_mode, non_mask_len)
        vol = vol.masked_fill(~m, float("-inf"))
    return vol


def test_normalize(dev):
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(1)
    for (n, C, H, W) in [(2, 256, 9, 13), (1, 64, 16, 16), (3, 40, 5, 7)]:
        x = torch.randn(n, C, H, W, generator=g)
        x[0, :, 0, 0] = 0          # zero vector -> eps path
        out = ops.normalize_to_hwc(x.to(dev)).cpu()
        ref = hwc(x)
        assert torch.allclose(out, ref, atol=1e-6, rtol=1e-5)
        raw = ops.normalize_to_hwc(x.to(dev), normalize=False).cpu()
        assert torch.equal(raw, x.flatten(2).transpose(1, 2))
        padded = ops.normalize_to_hwc(x.to(dev), pad=True).cpu()
        assert padded.shape[2] == ops.padded_channels(C)
        assert torch.equal(padded[..., :C], out) and float(padded[..., C:].abs().max() if padded.shape[2] > C else 0) == 0


MAE = ["mae_s8x12", "mae_s16x16", "mae_s32x32", "mae_s20x24_nml1", "mae_s12x20_cos", "mae_s16x24_c256"]


@pytest.mark.parametrize("name", MAE)
def test_affinity_topk_vs_reference_golden(dev, golden, name):
    from fgvc_amd import ops
    g = golden(name)
    q, key, v = T(g["query"])[0], T(g["key"])[0], T(g["value"])[0]
    nr, topk, nml, mode = int(g["nr"]), int(g["topk"]), int(g["non_mask_len"]), str(g["mode"])
    C, H, W = q.shape
    Tn = key.shape[1]
    feats, idx, logit, weight = gpu_affinity(dev, q, key, topk, 0.07, nr, non_mask_len=nml, mode=mode)
    # (1) against the float64 oracle slab: score tolerance, legitimacy, exactness on clear-gap queries
    stats = O.check_topk(dense64(q, key, 0.07, nr, non_mask_len=nml), idx.cpu().long(), logit.cpu(), topk, tol=TOL)
    assert stats["clear"] > 0.9 * stats["queries"] or name == "mae_s16x16"
    # (2) against what the reference's own topk returned (recorded by the golden generator)
    rv = T(g["ref_topk_val"])
    assert torch.allclose(logit.cpu(), rv.sort(1, descending=True)[0], atol=TOL)
    # (3) propagated labels against the reference's output tensor
    labels = v.permute(1, 2, 3, 0).reshape(Tn, H * W, -1).contiguous().to(dev)
    out = ops.propagate_topk(labels, torch.arange(Tn, dtype=torch.int32, device=dev), idx, weight, H, W, H, W)
    ref_out = T(g["out"])[0].flatten(1).t()
    assert torch.allclose(out.cpu(), ref_out, atol=TOL), float((out.cpu() - ref_out).abs().max())


def test_nomask_golden(dev, golden):
    g = golden("mae_nomask_10x14")
    q, key = T(g["query"])[0], T(g["key"])[0]
    feats, idx, logit, weight = gpu_affinity(dev, q, key, 10, 0.07, None)
    O.check_topk(dense64(q, key, 0.07, None), idx.cpu().long(), logit.cpu(), 10, tol=TOL)
    ri = T(g["ref_topk_idx"]).long()
    assert (idx.cpu().long() == ri).all(1).float().mean() > 0.99


@pytest.mark.parametrize("shape", [(256, 6, 30, 44, 30, "circle"), (128, 2, 17, 23, 9, "square"),
                                   (64, 3, 8, 8, 30, "circle"), (256, 1, 33, 70, 30, "circle"),
                                   (32, 2, 5, 3, 4, "circle")])
def test_affinity_topk_vs_oracle_seeded(dev, shape):
    """ragged sizes (not multiples of the 8x16 tile), both mask modes, tiny grids."""
    C, Tn, H, W, nr, mm = shape
    g = torch.Generator().manual_seed(sum(v for v in shape if isinstance(v, int)))
    q, key = torch.randn(C, H, W, generator=g), torch.randn(C, Tn, H, W, generator=g)
    topk = min(10, 5 if H * W < 20 else 10)
    feats, idx, logit, weight = gpu_affinity(dev, q, key, topk, 0.07, nr, mask_mode=mm)
    stats = O.check_topk(dense64(q, key, 0.07, nr, mm), idx.cpu().long(), logit.cpu(), topk, tol=TOL)
    assert stats["exact"] >= stats["clear"]
    # float32 oracle (the reference's arithmetic): indices identical wherever the f64 gaps are clear
    oi, ol = O.affinity_topk(q, key, topk, 0.07, neighbor_range=nr, mask_mode=mm)
    assert torch.allclose(logit.cpu(), ol, atol=TOL)
    w = O.topk_weights(ol)
    assert torch.allclose(weight.cpu(), w, atol=TOL)


def test_smooth_features_ties(dev):
    """spatially smooth features + duplicated key frame: many near-ties; validity must still hold."""
    g = torch.Generator().manual_seed(5)
    C, H, W = 64, 24, 24
    base = torch.randn(C, 6, 6, generator=g)
    q = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear")[0]
    k0 = q + 0.01 * torch.randn(C, H, W, generator=g)
    key = torch.stack([k0, k0, q], 1)                       # slot 0 == slot 1 (the reference's duplicate frame 0)
    feats, idx, logit, weight = gpu_affinity(dev, q, key, 10, 0.07, 14)
    O.check_topk(dense64(q, key, 0.07, 14), idx.cpu().long(), logit.cpu(), 10, tol=TOL)
    # exact duplicates: canonical order puts the lower slot first
    HW = H * W
    i = idx.cpu().long()
    dup = (i[:, 1:] - i[:, :-1] == HW) & (logit.cpu()[:, 1:] == logit.cpu()[:, :-1])
    assert dup.any()


def test_corr_volume(dev):
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(9)
    for (C, Hq, Wq, Hk, Wk) in [(256, 12, 20, 12, 20), (64, 9, 11, 7, 13), (128, 16, 16, 16, 16)]:
        q, k = torch.randn(C, Hq, Wq, generator=g), torch.randn(C, Hk, Wk, generator=g)
        ref64 = O.corr_volume(q.double(), k.double(), 0.07)
        qf, kf = ops.normalize_to_hwc(q[None].to(dev))[0], ops.normalize_to_hwc(k[None].to(dev))[0]
        v32 = ops.corr_volume(qf, kf, 0.07, "f32").cpu()
        assert v32.shape == ref64.shape
        e32 = float((v32.double() - ref64).abs().max())
        assert e32 < 2e-5, e32
        qs, ks = ops.split_bf16(qf), ops.split_bf16(kf)
        vx3 = ops.corr_volume(qs, ks, 0.07, "bf16x3").cpu()
        ex3 = float((vx3.double() - ref64).abs().max())
        assert ex3 < TOL, ex3
        vb = ops.corr_volume(qs, ks, 0.07, "bf16").cpu()
        eb = float((vb.double() - ref64).abs().max())
        assert eb < 3e-2, eb        # plain bf16 is NOT within the 1e-3 bar (reported, reduced precision; bound ~ 5 sigma of its rounding)
        e8 = -1.0
        if C == 256:               # f16 + block-scaled fp8 cross terms: parity-grade at two bf16-MFMA times per tile
            v8 = ops.corr_volume(ops.split_f16f8(qf), ops.split_f16f8(kf), 0.07, "f16f8").cpu()
            e8 = float((v8.double() - ref64).abs().max())
            assert e8 < TOL / 2, e8
            v6 = ops.corr_volume(ops.split_f16f6(qf), ops.split_f16f6(kf), 0.07, "f16f6").cpu()     # FP6 cross terms: 1.5 MFMA times
            e6 = float((v6.double() - ref64).abs().max())
            assert e6 < TOL / 2, e6
        print(f"C={C}: max |err| logits  f32 {e32:.2e}  bf16x3 {ex3:.2e}  f16f8 {e8:.2e}  bf16 {eb:.2e}")


def test_corr_volume_f16f8_ragged_shapes_and_adversarial_rows(dev):
    """fgvc_corr_volume_f16f8 and fgvc_corr_volume_f16f6 on grids whose HW is not a multiple of 32 (row classes of equal line phase: periods 1, 2, 4 and the
    unshifted fallback), non-square query/key grids, and on rows built to stress the split (one-hot, heavy-tailed, tiny and
    negative-zero components): every entry within 1e-3 of the float64 product."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(19)
    for (HWq, HWk, kind) in [(48 * 37 + 16, 1000, "gauss"), (2056, 777, "gauss"), (33 * 31, 33 * 31, "gauss"), (25 * 52, 64, "gauss"),
                             (4096, 4096, "onehot"), (1500, 1300, "heavy"), (31, 5, "gauss"), (300, 300, "relu")]:
        def rows(n):
            x = torch.randn(n, 256, generator=g)
            if kind == "onehot":
                x = x * (torch.rand(n, 256, generator=g) < 0.02) + 1e-4 * torch.randn(n, 256, generator=g)
                x[0] = 0.0
                x[0, 5] = -1.0
                x[1, 7] = -0.0
            elif kind == "heavy":
                x = x ** 3
            elif kind == "relu":
                x = torch.relu(torch.randn(1, 256, generator=g) + 0.3 * x)
            return torch.nn.functional.normalize(x, dim=1)
        q, k = rows(HWq), rows(HWk)
        ref = (k.double() @ q.double().t()) / 0.07
        for prec, split in (("f16f8", ops.split_f16f8), ("f16f6", ops.split_f16f6)):
            vol = ops.corr_volume(split(q.to(dev)), split(k.to(dev)), 0.07, prec).cpu()
            assert vol.shape == (HWk, HWq)
            err = float((vol.double() - ref).abs().max())
            assert err < TOL, (prec, HWq, HWk, kind, err)


def _decode_f16f6(sp):
    """fgvc_split_f16f6 rows (n, 1024) uint8 -> h, h6, l6 as (n, 256) float64 (h6 / l6 dequantised with their 2^(s-4) scales);
    the layout is documented in fgvc_amd/csrc/corr_volume_f6.hip"""
    import numpy as np
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
                sc = sp[:, 896 + 4 * g + (u if base == 512 else 2 + u)].astype(np.int64) - 127
                vals[:, 128 * u + 32 * g: 128 * u + 32 * g + 32] = lut[codes] * (2.0 ** sc)[:, None]
        outs.append(vals)
    return h, outs[0], outs[1]


def test_split_f16f6_format(dev):
    """The operand rows of fgvc_corr_volume_f16f6 decoded on the host: h is exactly f16(256 x); h6 / l6 (block-scaled e2m3, one
    E8M0 scale per 32 channels) reproduce h and its residual to half an e2m3 step of their block; the pad is zero."""
    import numpy as np
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(5)
    f = torch.nn.functional.normalize(torch.randn(2048, 256, generator=g), dim=1)
    f[:64] = 0
    f[:64, 3] = 1.0                                           # one-hot rows: all-zero blocks next to a block holding 256
    f[64:128] = torch.nn.functional.normalize(f[64:128] * (torch.rand(64, 256, generator=g) < 0.05) + 1e-6, dim=1)
    sp = ops.split_f16f6(f.to(dev))
    assert sp.shape == (2048, 1024) and sp.dtype == torch.uint8
    h, h6, l6 = _decode_f16f6(sp)
    x = f.double().numpy()
    h_ref = (f.numpy() * np.float32(256)).astype(np.float16).astype(np.float64)
    l_ref = (x * 256 - h_ref) * 256
    assert (h == h_ref).all()
    assert (sp[:, 912:] == 0).all()
    for got, ref in ((h6 * 16, h_ref), (l6 * 16, l_ref)):     # each stored scale carries 2^-4
        bm = np.abs(ref).reshape(-1, 8, 32).max(-1, keepdims=True).repeat(32, -1).reshape(ref.shape)
        bound = np.maximum(np.abs(ref) / 16, bm / 7.5 / 8) * 1.0001 + 1e-30
        assert (np.abs(got - ref) <= bound).all()
    # the three sums the kernel forms reproduce the float64 product to the parity bar
    k, q = slice(0, 1024), slice(1024, 2048)
    tot = h[k] @ h[q].T + (h6[k] @ l6[q].T + l6[k] @ h6[q].T)
    assert float(np.abs(tot / 65536.0 / 0.07 - (x[k] @ x[q].T) / 0.07).max()) < TOL


def test_dense_golden(dev, golden):
    from fgvc_amd import ops
    g = golden("dense_9x11")
    q, key = T(g["query"])[0], T(g["key"])[0]
    qf = ops.normalize_to_hwc(q[None].to(dev))[0]
    kf = ops.normalize_to_hwc(key.permute(1, 0, 2, 3).contiguous().to(dev))
    v0 = ops.corr_volume(qf, kf[0], 0.07, "f32").cpu()
    assert torch.allclose(v0, T(g["compute_affinity"]), atol=1e-4)
    v1 = ops.corr_volume(qf, kf[1], 0.07, "f32").cpu()
    att = T(g["non_local_att"])
    assert torch.allclose(v1.t(), att[1], atol=1e-4)


def test_local_corr_golden(dev, golden):
    from fgvc_amd import ops
    g = golden("localcorr_10x12")
    q, key, v = T(g["query"])[0], T(g["key"])[0], T(g["value"])[0]
    R, topk = int(g["radius"]), int(g["topk"])
    C, H, W = q.shape
    K = key.shape[1]
    qf = ops.normalize_to_hwc(q[None].to(dev))
    kf = ops.normalize_to_hwc(key.permute(1, 0, 2, 3).contiguous().to(dev))
    idx, logit, weight = ops.local_corr_topk(qf, kf, H, W, R, topk, 0.07)
    o_out, o_idx, o_logit = O.local_corr_topk(q, key.transpose(0, 1), v.transpose(0, 1), R, topk, 0.07)
    assert torch.allclose(logit.cpu(), o_logit, atol=TOL)
    assert (idx.cpu().long() == o_idx).all(1).float().mean() > 0.98
    labels = v.permute(1, 2, 3, 0).reshape(K, H * W, -1).contiguous().to(dev)
    out = ops.propagate_topk(labels, torch.arange(K, dtype=torch.int32, device=dev), idx, weight, H, W, H, W,
                             window_L=2 * R + 1)
    assert torch.allclose(out.cpu(), T(g["out"])[0].flatten(1).t(), atol=TOL)


def test_c2f_golden(dev, golden):
    from fgvc_amd import ops
    g = golden("c2f_8x10")
    q, key = T(g["query"])[0], T(g["key"])[0]
    qfine, kfine, v = T(g["query_fine"])[0], T(g["key_fine"])[0], T(g["value"])[0]
    nr, topk, Rf = int(g["nr"]), int(g["topk"]), int(g["radius_fine"])
    C, H, W = q.shape
    Tn = key.shape[1]
    scale = kfine.shape[2] // H
    # coarse stage: top-1 per key slot
    frames = torch.cat([q[None], key.permute(1, 0, 2, 3)], 0).to(dev)
    feats = ops.normalize_to_hwc(frames)
    pairs = ops.make_pairs([(0, 1 + t) for t in range(Tn)], dev)
    cidx, _ = ops.pair_topk(feats, feats, pairs, H, W, H, W, ops.MaskSpec.from_neighbor_range(nr), 1)
    coarse = cidx[:, :, 0].contiguous()
    o_out, o_arg, o_idx, o_logit = O.c2f_attention(q, key, qfine, kfine, v, topk, 0.07, neighbor_range=nr,
                                                   radius_fine=Rf)
    assert (coarse.cpu().long() == o_arg).float().mean() > 0.99
    qf = ops.normalize_to_hwc(qfine[None].to(dev))[0]
    kf = ops.normalize_to_hwc(kfine.permute(1, 0, 2, 3).contiguous().to(dev))
    vf = v.permute(1, 2, 3, 0).reshape(Tn, -1, v.shape[0]).contiguous().to(dev)
    out, idx, logit = ops.c2f_refine(o_arg.to(dev, torch.int32), qf, kf, vf, H, W, scale, Rf, topk, 0.07)
    assert torch.allclose(logit.cpu(), o_logit, atol=TOL)
    assert (idx.cpu().long() == o_idx).all(1).float().mean() > 0.98
    assert torch.allclose(out.cpu(), T(g["out"])[0].flatten(1).t(), atol=TOL)


def test_readout_golden(dev, golden):
    from fgvc_amd import ops
    g = golden("readout_small")
    pts = T(g["gauss_points"])
    lab = ops.gaussian_labels(pts.to(dev), 12, 16, int(g["stride"])).cpu()
    assert torch.allclose(lab.t().reshape(2, 12, 16), T(g["gauss_res"]), atol=1e-6)
    # top-5 soft-argmax with no upsampling (Hf==h): the reference's img2coord on the same maps
    maps = T(g["maps"])                                     # (T,P,h,w)
    Tn, P, h, w = maps.shape
    labels = maps.permute(0, 2, 3, 1).reshape(Tn, h * w, P).contiguous().to(dev)
    c = ops.softargmax_top5(labels, h, w, h, w).cpu()       # (T,P,2)
    ref = T(g["coords"]).permute(2, 1, 0)                   # (2,P,T) -> (T,P,2)
    assert torch.allclose(c, ref, atol=1e-5), (c - ref).abs().max()
    # analytic Gaussian frame 0 + bilinear upsample path against the oracle
    gen = torch.Generator().manual_seed(3)
    lab2 = torch.rand(2, 12 * 16, 2, generator=gen)
    c2 = ops.softargmax_top5(lab2.to(dev), 12, 16, 24, 32, gauss_points=pts.to(dev)).cpu()
    full, _ = O.gaussian_labels(pts, 24, 32, 2)
    up = O.upsample_bilinear(lab2[1].t().reshape(2, 12, 16), 24, 32)
    oc = T(O.img2coord(torch.stack([full, up], 0).numpy())).permute(2, 1, 0)
    assert torch.allclose(c2, oc, atol=1e-4), (c2 - oc).abs().max()


def _readout_both(ops, labels, Hf, Wf, h, w, **kw):
    try:
        ops.set_option("readout_prune", 0)
        full = ops.softargmax_top5(labels, Hf, Wf, h, w, **kw).cpu()
    finally:
        ops.set_option("readout_prune", 1)
    return ops.softargmax_top5(labels, Hf, Wf, h, w, **kw).cpu(), full


def test_split_feature_bank_equals_f32_bank(dev):
    """The bank as split_bf16() rows straight from the trunk output (one pass: normalize_nhwc(split=True), and
    VanillaTracker.get_feats_hwc(split=True)) is bit-identical to normalising first and splitting after, and engine.run_affinity
    gives the same lists from either form."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import engine, ops
    g = torch.Generator().manual_seed(41)
    torch.manual_seed(41)                                               # the encoder's random initialisation
    y = torch.randn(3, 9, 13, 256, generator=g).to(dev)
    f = ops.normalize_nhwc(y, True)
    assert torch.equal(ops.normalize_nhwc(y, True, split=True), ops.split_bf16(f))
    assert torch.equal(ops.normalize_nhwc(y, False, split=True), ops.split_bf16(ops.normalize_nhwc(y, False)))
    assert float((ops.unsplit_bf16(ops.split_bf16(f)) - f).abs().max()) < 1e-5
    assert torch.equal(ops.normalize_nhwc(y, True, split="f16"), ops.split_f16x2(f))       # the f16 (h, l) form of fgvc_pair_topk_f16x3
    assert float((ops.unsplit_f16x2(ops.split_f16x2(f)) - f).abs().max()) < 2e-7
    model = api.build_model(dict(type="VanillaTracker",
                                 backbone=dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none")),
                            train_cfg=None,
                            test_cfg=api.ConfigDict(precede_frames=3, topk=10, temperature=0.07, neighbor_range=12,
                                                    with_first=True, with_first_neighbor=True, batch_step=2)).to(dev).eval()
    frames = torch.randn(5, 3, 64, 96, generator=g).to(dev)
    bank_s, Hf, Wf = model.get_feats_hwc(frames, split=True)            # 3 encoder calls (2 + 2 + 1 frames), concatenated
    bank_f, Hf2, Wf2 = model.get_feats_hwc(frames)
    assert bank_s.dtype == torch.int16 and bank_s.shape == (5, Hf * Wf, 4, 256) and (Hf, Wf) == (Hf2, Wf2) == (16, 24)
    cfg = model.engine_config()
    assert cfg.pair_split_fmt == "f16f6" and cfg.bank_fmt == "f16f6x"  # an f16f8 / f16f6 trunk goes with the f16 + FP6 pair kernel (round 4) on 2 KiB rows: + the exact channels (round 5)
    assert torch.equal(ops.normalize_nhwc(y, True, split="f16f6"), ops.split_f16f6p(f))    # ... whose rows the same single pass writes
    assert float((ops.unsplit_f16f6p(ops.split_f16f6p(f)) - f).abs().max()) < 2.0 ** -12  # (their h part: 11 bits of 256 x, |x| <= 1)
    assert torch.equal(ops.split_f16f6x(bank_f), bank_s)                # the hand-written trunk is deterministic call to call
    assert torch.equal(ops.f32_of_f16f6x(bank_s), bank_f)               # ... and the bank's second KiB IS the f32 bank
    plan = engine.plan_clip(5, [0], cfg)
    a = engine.run_affinity(bank_s, Hf, Wf, plan, cfg)
    b = engine.run_affinity(bank_f, Hf, Wf, plan, cfg)                  # f32 form: split inside
    assert torch.equal(a.idx, b.idx) and torch.equal(a.weight, b.weight)
    # the 1e-7-grade form on request (and by itself under any other arithmetic of the trunk): same lists wherever the float64 ranks are clear
    model.test_cfg["pair_split_fmt"] = "f16"
    cfg3 = model.engine_config()
    bank_3 = model.get_feats_hwc(frames, split=True)[0]
    assert cfg3.pair_split_fmt == "f16" and torch.equal(ops.split_f16x2(bank_f), bank_3)
    a3 = engine.run_affinity(bank_3, Hf, Wf, plan, cfg3)
    assert float((a3.logit - a.logit).abs().max()) < 2e-4 and float((a3.idx == a.idx).all(-1).float().mean()) > 0.97
    del model.test_cfg["pair_split_fmt"]
    model.backbone.set_arith("f16x3")
    assert model.engine_config().pair_split_fmt == "f16"
    model.backbone.set_arith("f16f6")
    cfg32 = engine.TrackerConfig(**{**cfg.__dict__, "pair_precision": "f32"})
    with pytest.raises(ValueError):
        engine.run_affinity(bank_s, Hf, Wf, plan, cfg32)
    # the retired bf16 operand format is refused, not silently replaced
    model.test_cfg["pair_split_fmt"] = "bf16"
    with pytest.raises(ValueError):
        engine.run_affinity(bank_f, Hf, Wf, plan, model.engine_config())
    del model.test_cfg["pair_split_fmt"]
    assert not ops.pair_f16x3_timed_out()


def test_c2f_operator_256_channels_vs_oracle(dev):
    """masked_attention_efficient_c2f at the API level with 256 coarse channels: its coarse arg-max stage then runs on the
    bf16-pipe pair kernel (top-1); against the oracle's c2f_attention (local_attention.py:721-880)."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd.mmpt_api.common import masked_attention_efficient_c2f, spatial_neighbor
    g = torch.Generator().manual_seed(52)
    C, Cf, H, W, Tn, P, scale, topk, Rf, nr = 256, 32, 12, 16, 3, 3, 2, 5, 3, 8
    q = torch.randn(1, C, H, W, generator=g)
    key = torch.randn(1, C, Tn, H, W, generator=g)
    qfine = torch.randn(1, Cf, H * scale, W * scale, generator=g)
    kfine = torch.randn(1, Cf, Tn, H * scale, W * scale, generator=g)
    v = torch.rand(1, P, Tn, H * scale, W * scale, generator=g)
    mask = spatial_neighbor(1, H, W, neighbor_range=nr, device=dev, dtype=torch.bool, dim=1, mode="circle")
    out = masked_attention_efficient_c2f(q.to(dev), key.to(dev), qfine.to(dev), kfine.to(dev), v.to(dev), mask,
                                         temperature=0.07, topk=topk, normalize=True, radius_fine=Rf).cpu()
    o_out, o_arg, _, _ = O.c2f_attention(q[0], key[0], qfine[0], kfine[0], v[0], topk, 0.07, neighbor_range=nr, radius_fine=Rf)
    assert out.shape == (1, P, H, W)
    close = (out[0] - o_out).abs().amax(0) < 1e-4                 # per query pixel (a flipped arg-max changes the whole window)
    assert float(close.float().mean()) > 0.99, float(close.float().mean())


def test_run_propagation_async_equals_sync(dev):
    """engine.run_propagation_async (sweep + read-out on a side stream, the caller's stream free for the next clip) returns what
    run_propagation returns, also when the caller immediately reuses its stream and drops its references."""
    from fgvc_amd import engine, ops
    g = torch.Generator().manual_seed(33)
    Tn, C, Hf, Wf, h, w = 6, 64, 16, 20, 64, 80
    feats = ops.normalize_to_hwc(torch.randn(Tn, C, Hf, Wf, generator=g).to(dev))
    cfg = engine.TrackerConfig(neighbor_range=12)
    plan = engine.plan_clip(Tn, [0], cfg)
    pts = (torch.rand(5, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])).to(dev)
    side = torch.cuda.Stream(dev)
    want_l, want_c = engine.run_propagation(engine.run_affinity(feats, Hf, Wf, plan, cfg), 0, pts, Hf, Wf, h, w, cfg)
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        tk = engine.run_affinity(feats, Hf, Wf, plan, cfg) if rep % 2 else engine.run_pairs(feats, Hf, Wf, plan, cfg)
        labels, coords, done = engine.run_propagation_async(tk, 0, pts, Hf, Wf, h, w, cfg, side)    # merged lists | pair lists
        del tk                                                     # the allocator may hand these blocks out again right away
        junk = [torch.full((1 << 20,), float(rep), device=dev) for _ in range(8)]     # ... to this, on the caller's stream
        outs.append((labels, coords, done))
        del junk
    for labels, coords, done in outs:
        done.synchronize()
        assert torch.equal(coords, want_c) and torch.equal(labels, want_l)


def test_readout_pruned_equals_full_scan(dev):
    """The pruned read-out (only coarse cells whose corner maximum reaches the running 5th value are upsampled) returns the
    bits of the full scan -- on peaked maps (its fast path) and on everything it must hand back: flat and constant maps,
    negative labels, all-zero maps, equal twin peaks, plateaus of exact ties, up- and down-sampling, ragged scales."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(21)

    def blobs(Tn, Hf, Wf, P, sigma=1.5, n_peaks=1, equal=False):
        ys = torch.arange(Hf).view(1, Hf, 1, 1).float()
        xs = torch.arange(Wf).view(1, 1, Wf, 1).float()
        lab = torch.zeros(Tn, Hf, Wf, P)
        for k in range(n_peaks):
            cy = torch.rand(Tn, 1, 1, P, generator=g) * (Hf - 1)
            cx = torch.rand(Tn, 1, 1, P, generator=g) * (Wf - 1)
            amp = 1.0 if equal else float(0.5 + 0.5 * torch.rand(1, generator=g))
            lab = lab + amp * torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * sigma ** 2))
        return lab.reshape(Tn, Hf * Wf, P).contiguous()

    cases = []
    cases.append(("peaked x4", blobs(3, 30, 54, 16), 30, 54, 120, 214))
    cases.append(("ragged scale", blobs(2, 30, 54, 5), 30, 54, 119, 213))
    cases.append(("scale 8", blobs(2, 16, 20, 3, sigma=1.0), 16, 20, 128, 160))
    cases.append(("no upsampling", blobs(2, 24, 32, 4), 24, 32, 24, 32))
    cases.append(("downsampling", blobs(2, 24, 32, 4, sigma=3.0), 24, 32, 12, 16))
    cases.append(("three peaks", blobs(2, 30, 54, 8, n_peaks=3), 30, 54, 120, 216))
    twin = blobs(1, 30, 54, 4)
    twin = torch.maximum(twin, twin.reshape(1, 30, 54, 4).flip(1, 2).reshape(1, 30 * 54, 4))    # exact mirror twins
    cases.append(("equal twin peaks", twin, 30, 54, 120, 216))
    cases.append(("uniform noise", torch.rand(2, 30 * 54, 6, generator=g), 30, 54, 120, 216))
    cases.append(("constant", torch.full((1, 30 * 54, 3), 0.25), 30, 54, 120, 216))
    z = blobs(2, 30, 54, 4)
    z[0, :, 1] = 0.0
    z[1, :, 2] = 0.0
    cases.append(("all-zero maps among peaked ones", z, 30, 54, 120, 216))
    n = blobs(2, 30, 54, 4)
    n[0, 7, 0] = -0.5
    n[1, :, 3] -= 0.1
    cases.append(("negative labels", n, 30, 54, 120, 216))
    pl = torch.zeros(1, 30 * 54, 2)
    pl.view(1, 30, 54, 2)[0, 10:14, 20:26, :] = 1.0                           # plateau: hundreds of exactly equal pixels
    cases.append(("plateau", pl, 30, 54, 120, 216))
    one = torch.zeros(1, 30 * 54, 2)
    one[0, 0, 0] = 1.0                                                        # single hot corner cell
    one[0, 30 * 54 - 1, 1] = 1.0
    cases.append(("hot corners", one, 30, 54, 120, 216))
    for name, lab, Hf, Wf, h, w in cases:
        pruned, full = _readout_both(ops, lab.to(dev), Hf, Wf, h, w)
        assert torch.equal(pruned, full), (name, (pruned - full).abs().max())
    # first frame read out from the analytic Gaussian: centres inside, on the border, outside, far outside (-> -1)
    pts = torch.tensor([[50.3, 40.7], [0.0, 0.0], [215.0, 119.0], [-3.2, 60.1], [100.5, 140.0], [-400.0, 50.0],
                        [107.5, 59.5], [300.0, -200.0]])
    lab = blobs(2, 30, 54, pts.shape[0])
    pruned, full = _readout_both(ops, lab.to(dev), 30, 54, 120, 216, gauss_points=pts.to(dev))
    assert torch.equal(pruned, full), (pruned - full).abs().max()
    assert (pruned[0, 5] == -1).all() and (pruned[0, 7] == -1).all() and (pruned[0, 0] > 0).all()


def test_engine_vs_oracle_tracker(dev):
    """whole post-encoder path (plan -> pair top-k -> merge -> sweep -> read-out) vs the oracle driver."""
    from fgvc_amd import engine, ops
    g = torch.Generator().manual_seed(11)
    Tn, C, Hf, Wf, h, w = 9, 64, 16, 20, 32, 40
    feats = torch.randn(Tn, C, Hf, Wf, generator=g)
    qp = torch.tensor([[0., 10., 20.], [0., 30.2, 5.7], [3., 12.3, 9.1], [3., 25.0, 25.0], [6., 3.3, 30.1]])
    cfg = engine.TrackerConfig(neighbor_range=12, regroup=True)
    fh = ops.normalize_to_hwc(feats.to(dev))
    traj, order = engine.track_points(fh, Hf, Wf, h, w, qp, cfg)
    traj = traj.cpu()
    col = 0
    for s in sorted(set(qp[:, 0].int().tolist())):
        sel = (qp[:, 0].int() == s).nonzero().flatten()
        ref = O.forward_test_main(feats[s:], qp[sel, 1:], h, w, neighbor_range=12)[0]      # (T-s,P,2)
        got = traj[s:, col:col + sel.numel()]
        assert torch.allclose(got, ref, atol=5e-3), float((got - ref).abs().max())
        assert float(traj[:s, col:col + sel.numel()].abs().max() if s else 0.0) == 0.0
        col += sel.numel()
    assert order.tolist() == [0, 1, 2, 3, 4]


def test_pair_f32_kernel_is_prefix_consistent_and_deterministic(dev):
    """The exact-f32 pair kernel (the fallback of the 16-bit one and its reference in these tests): a shorter list is the prefix of
    a longer one (the list length is a compile-time register array: 1 / 5 / 10 / 16), and two launches give identical bits."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(77)
    for (C, H, W, nr) in [(256, 37, 53, 30), (64, 9, 70, 12), (128, 20, 20, None)]:
        f = ops.normalize_to_hwc(torch.randn(3, C, H, W, generator=g).to(dev))
        mask = ops.MaskSpec.from_neighbor_range(nr)
        pairs = ops.make_pairs([(2, 0, nr is not None), (2, 1, nr is not None), (1, 0, nr is not None)], dev)
        i10, s10 = ops.pair_topk(f, f, pairs, H, W, H, W, mask, 10)
        i10b, s10b = ops.pair_topk(f, f, pairs, H, W, H, W, mask, 10)
        assert torch.equal(i10, i10b) and torch.equal(s10, s10b)
        for k in (1, 4, 16):
            ik, sk = ops.pair_topk(f, f, pairs, H, W, H, W, mask, k)
            kk = min(k, 10)
            assert torch.equal(ik[..., :kk], i10[..., :kk]) and torch.equal(sk[..., :kk], s10[..., :kk])


def test_tracker_end_to_end_on_synthetic_tapvid(dev):
    """tools/test.py path: registry-built VanillaTracker -> 5-tuple -> TAP-Vid metrics.  The synthetic clip is a
    rigidly translating texture, so even a RANDOM-INIT ResNet's features match exactly between frames and label
    propagation must track the points (a smoke test of the whole drop-in path, not an accuracy claim)."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import apis, metrics
    from fgvc_amd.datasets import StridedLoader, SyntheticTapVid
    test_cfg = api.ConfigDict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, with_first=True,
                              with_first_neighbor=True)
    model = api.build_model(dict(type="VanillaTracker",
                                 backbone=dict(type="ResNet", depth=18, strides=(1, 1, 1, 4), out_indices=(2,),
                                               pool_type="none", zero_init_residual=False)),
                            train_cfg=None, test_cfg=test_cfg)
    torch.manual_seed(0)
    model.init_weights()
    model = model.to(dev).eval()
    ds = SyntheticTapVid(n_videos=2, frames=6, size=(128, 128), points=6, query_mode="strided", device=dev)
    outs = apis.single_gpu_test(model, StridedLoader(ds))
    assert len(outs) == 2 and outs[0][2].shape == (1, 6, 6, 2)
    traj, vis, pred, vpred, qp = outs[0]
    assert bool((qp[0, 1:, 0] >= qp[0, :-1, 0]).all())                # regrouped by query time
    s = metrics.tapvid_evaluate(outs, "strided")
    assert s["average_pts_within_thresh"] > 60.0, s


def test_encoder_fused_bn_act_matches_torch(dev):
    """A1: the ResNet with the fused BN(+residual)+ReLU kernel vs the same weights through plain torch modules
    (the oracle ResNet on the CPU), including non-trivial BN statistics."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(4)
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none"))
    ora = O.ResNet18((1, 2, 1, 1), 2, "none")
    sd = O.seeded_resnet_state(5, (1, 2, 1, 1), "none")
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        elif k.endswith("bn.weight"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        elif k.endswith("bn.bias"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    net.load_state_dict(sd)
    ora.load_state_dict(sd)
    x = torch.randn(2, 3, 64, 96, generator=g)
    net = net.to(dev).eval()
    with torch.no_grad():
        b = ora.eval()(x)
        for arith, atol in (("bf16x3", 2e-4), ("f16x3", 2e-4), ("f16f8", 6e-4), ("f16f6", 6e-4)):       # whole trunk; features up to ~20: 1e-5 / 3e-5 of the largest
            net.set_arith(arith)
            a = net(x.to(dev)).cpu()
            assert a.shape == b.shape == (2, 256, 16, 24)
            assert torch.allclose(a, b, atol=atol, rtol=1e-4), (arith, float((a - b).abs().max()), float(b.abs().max()))
    # odd spatial size -> scalar path of the kernel; residual + relu
    y = torch.randn(2, 8, 5, 7, generator=g)
    r = torch.randn(2, 8, 5, 7, generator=g)
    bn = torch.nn.BatchNorm2d(8).eval()
    bn.running_mean.normal_(generator=g); bn.running_var.uniform_(0.5, 2.0, generator=g)
    bn.weight.data.uniform_(0.5, 1.5, generator=g); bn.bias.data.normal_(generator=g)
    ref = torch.relu(bn(y) + r)
    got = ops.bn_act(y.to(dev), bn.to(dev), r.to(dev), True, inplace=False).cpu()
    assert torch.allclose(got, ref.detach(), atol=1e-5, rtol=1e-5)


def test_get_coord_vs_oracle(dev):
    """A7 forward-warping field (vanilla_tracker.py:445-488)."""
    import fgvc_amd.mmpt_api as api
    g = torch.Generator().manual_seed(8)
    C, H, W, R, scale = 64, 14, 18, 4, 4
    q, k = torch.randn(1, C, H, W, generator=g), torch.randn(1, C, H, W, generator=g)
    trk = api.build_model(dict(type="HRVanillaTracker",
                               backbone=dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,),
                                             pool_type="none")),
                          train_cfg=None, test_cfg=api.ConfigDict(neighbor_range=2 * R, topk=6, temperature=0.07))
    field = trk.get_coord(q.to(dev), k.to(dev), (H * scale, W * scale), scale).cpu()
    ref = O.get_coord(q[0], k[0], R, 6, 0.07, scale)
    assert field.shape == (1, 2, H, W)
    assert torch.allclose(field[0], ref, atol=2e-3), float((field[0] - ref).abs().max())


# ---------------------------------------------------------------------------------------------------------------
# fgvc_pair_topk_f16x3: the same operator on the f16 matrix pipe (h/l split features, fixed-point selection keys)
# ---------------------------------------------------------------------------------------------------------------
def split_affinity(dev, q, key, topk, temperature, neighbor_range, mask_mode="circle"):
    from fgvc_amd import ops
    C, H, W = q.shape
    Tn = key.shape[1]
    frames = torch.cat([q.unsqueeze(0), key.permute(1, 0, 2, 3)], 0).to(dev)
    feats = ops.normalize_to_hwc(frames)
    mask = ops.MaskSpec.from_neighbor_range(neighbor_range, mask_mode)
    pairs = ops.make_pairs([(0, 1 + t, not mask.is_none) for t in range(Tn)], dev)
    h16 = ops.split_f16x2(feats)                   # fgvc_pair_topk_f16x3 (the engine's default)
    pidx, pscore = ops.pair_topk_split(h16, h16, pairs, H, W, H, W, mask, topk)
    assert not ops.pair_f16x3_timed_out()
    fidx, fscore = ops.pair_topk(feats, feats, pairs, H, W, H, W, mask, topk)
    slot_pair = torch.arange(Tn, dtype=torch.int32, device=dev).view(1, Tn)
    idx, logit, weight = ops.merge_topk(pidx, pscore, slot_pair, H * W, topk, temperature, "softmax")
    return (pidx, pscore, fidx, fscore), idx[0], logit[0], weight[0]


@pytest.mark.parametrize("shape", [(3, 30, 44, 30, "circle", 10), (2, 17, 23, 9, "square", 10), (1, 33, 70, 30, "circle", 5),
                                   (2, 5, 3, 4, "circle", 5), (2, 20, 20, None, "circle", 10), (1, 9, 130, 12, "circle", 3),
                                   (6, 8, 8, 30, "circle", 10), (2, 37, 53, 30, "circle", 10), (1, 61, 47, 14, "square", 7)])
def test_split_pair_topk_vs_oracle(dev, shape):
    """Ragged grids, both mask modes, no mask, k in {3,5,7,10}; fgvc_pair_topk_f16x3: indices exact wherever the f64 ranks are
    clear, scores within the north_star bar (and within 1e-5 of the f32-MFMA kernel)."""
    Tn, H, W, nr, mm, topk = shape
    g = torch.Generator().manual_seed(sum(v for v in shape if isinstance(v, int)))
    q, key = torch.randn(256, H, W, generator=g), torch.randn(256, Tn, H, W, generator=g)
    (pidx, pscore, fidx, fscore), idx, logit, weight = split_affinity(dev, q, key, topk, 0.07, nr, mm)
    stats = O.check_topk(dense64(q, key, 0.07, nr, mm), idx.cpu().long(), logit.cpu(), topk, tol=TOL)
    assert stats["exact"] >= stats["clear"]
    assert stats["max_score_err"] < 5e-5                       # logit units (score / 0.07)
    fin = torch.isfinite(fscore)
    assert torch.equal(fin, torch.isfinite(pscore)) and torch.equal(pidx < 0, fidx < 0)
    assert float((pscore - fscore)[fin].abs().max()) < 1e-5
    assert float((pidx == fidx).all(-1).float().mean()) > 0.99
    oi, ol = O.affinity_topk(q, key, topk, 0.07, neighbor_range=nr, mask_mode=mm)
    assert torch.allclose(logit.cpu(), ol, atol=TOL) and torch.allclose(weight.cpu(), O.topk_weights(ol), atol=TOL)


def test_split_pair_topk_exact_ties_and_identical_frames(dev):
    """Key frame == query frame: every query's best match is itself with score 1 (fixed-point 2^30 exactly); a constant
    feature map makes every in-window candidate tie -- any k of them is legitimate, but scores must all be 1."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(4)
    H, W = 21, 37
    f = ops.normalize_to_hwc(torch.randn(1, 256, H, W, generator=g).to(dev))
    mask = ops.MaskSpec.from_neighbor_range(30)
    pairs = ops.make_pairs([(0, 0, True)], dev)
    const = torch.ones(1, 256, H, W)
    fc = ops.normalize_to_hwc(const.to(dev))
    qy, qx = torch.arange(H * W) // W, torch.arange(H * W) % W
    for fmt, split in (("f16", ops.split_f16x2),):
        hl = split(f)
        idx, score = ops.pair_topk_split(hl, hl, pairs, H, W, H, W, mask, 10, fmt=fmt)
        assert torch.equal(idx[0, :, 0].cpu(), torch.arange(H * W, dtype=torch.int32))
        self64 = (f[0].double() ** 2).sum(1)
        assert float((score[0, :, 0].double() - self64).abs().max()) < 2.5e-6
        idx, score = ops.pair_topk_split(split(fc), split(fc), pairs, H, W, H, W, mask, 10, fmt=fmt)
        assert float((score.double() - float((fc[0, 0].double() ** 2).sum())).abs().max()) < 2.5e-6
        assert float(score.max() - score.min()) == 0.0                 # identical operands -> identical fixed-point keys
        ii = idx[0].cpu().long()
        d2 = (ii // W - qy.view(-1, 1)) ** 2 + (ii % W - qx.view(-1, 1)) ** 2
        assert int(d2.max()) <= mask.r2max and all(len(set(r.tolist())) == 10 for r in ii[:: 37])
        # canonical order among exact ties: ascending pixel index
        assert bool((ii[:, 1:] > ii[:, :-1]).all())
    assert not ops.pair_f16x3_timed_out()


def test_pair_f16x3_runs_of_pairs_equal_pair_by_pair(dev):
    """fgvc_pair_topk_f16x3_runs (a query frame's pairs in one workgroup: one query prologue, the key-block ring never drains) gives
    bit-identical lists to one workgroup per pair, on a grid with edge tiles, for runs of 1-4 pairs, masked and not -- and both equal
    the f32-MFMA kernel wherever scores are not within rounding of each other.  (This is the configuration in which a consumer that
    released a ring slot before its block was staged let the producers overwrite a block another consumer was still reading.)"""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(77)
    H, W, T = 33, 70, 6
    f = ops.normalize_to_hwc(torch.randn(T, 256, H, W, generator=g).to(dev))
    h16 = ops.split_f16x2(f)
    rows = [(1, 0, True), (2, 0, True), (2, 1, True), (3, 0, False), (3, 1, True), (3, 2, True), (5, 0, True), (5, 1, True), (5, 3, True),
            (5, 4, True), (4, 4, True)]
    pairs = ops.make_pairs(rows, dev)
    assert ops.pair_runs(pairs).tolist() == [[6, 4], [1, 2], [4, 2], [0, 1], [3, 1], [10, 1]]     # longest first; a mask flag splits a run
    mask = ops.MaskSpec.from_neighbor_range(30)
    for _ in range(3):                                            # the race was timing dependent
        ia, sa = ops.pair_topk_split(h16, h16, pairs, H, W, H, W, mask, 10, fmt="f16")
        ib, sb = ops.pair_topk_split(h16, h16, pairs, H, W, H, W, mask, 10, fmt="f16", use_runs=False)
        assert torch.equal(ia, ib) and torch.equal(sa, sb)
    assert not ops.pair_f16x3_timed_out()
    i3, s3 = ops.pair_topk(f, f, pairs, H, W, H, W, mask, 10)
    fin = torch.isfinite(s3)
    assert torch.equal(fin, torch.isfinite(sa)) and float((sa - s3)[fin].abs().max()) < 2.5e-6   # f32 accumulation of the f32 kernel
    differ = ~(ia == i3).all(-1)
    assert float(differ.float().mean()) < 2e-3                    # ... and only where two scores are within rounding of each other
    if differ.any():
        assert float((torch.sort(sa[differ], dim=-1).values - torch.sort(s3[differ], dim=-1).values).abs().max()) < 2.5e-6


def test_conv256_forms_are_bit_identical(dev):
    """The two builds of the 256-channel-tile bf16x3 convolution -- hand-ordered assembly stage (default) and the compiler's
    schedule (conv_debug = 16) -- accumulate in the same order and must agree bit for bit, with residual + ReLU + both outputs, on a ragged grid (edge tiles) and two input widths."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(9)
    for Cin, H, W in ((128, 19, 45), (256, 24, 70)):
        wt = (torch.randn(256, Cin, 3, 3, generator=g) * 0.03).to(dev)
        bn = torch.nn.BatchNorm2d(256).eval().to(dev)
        bn.running_mean.copy_(torch.randn(256, generator=g).to(dev) * 0.1)
        bn.running_var.copy_(torch.rand(256, generator=g).to(dev) + 0.5)
        wp, bs = ops.prepare_conv_split(wt, bn)
        xs = ops.nchw_to_split_nhwc(torch.randn(3, Cin, H, W, generator=g).to(dev))
        res = ops.alloc_nhwc(3, 256, H, W, dev)
        res.copy_(torch.randn(res.shape, generator=g).to(dev))
        outs = []
        for dbg in (0, 16):
            ys, yf = ops.alloc_split_nhwc(3, 256, H, W, dev), ops.alloc_nhwc(3, 256, H, W, dev)
            ops.set_option("conv_debug", dbg)
            try:
                ops.conv_split(xs, wp, bs, H, W, True, out_split=ys, out_f32=yf, residual=res)
            finally:
                ops.set_option("conv_debug", 0)
            outs.append((ys, yf))
        for ys, yf in outs[1:]:
            assert torch.equal(ys, outs[0][0]) and torch.equal(yf, outs[0][1])


def test_pair_f16x3_three_roles_equal_two_roles_and_large_lists_fall_back(dev):
    """The default form of fgvc_pair_topk_f16x3 (consumer / selector / producer waves, accumulators handed over through the LDS) gives
    bit-identical lists to the two-role form (pair_f16_debug = 1024) -- same products, same keys, same networks -- on grids with
    edge tiles, K = 10 and K = 5, disc and rectangular windows, masked and unmasked pairs.  An unmasked pair on a grid whose block
    list exceeds the three-role form's 2048 entries (here 65 x 33 = 2145 blocks) runs on the two-role form: checked against the
    f32-MFMA kernel."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(5)
    for (H, W, T, topk, nr, rect) in [(33, 70, 4, 10, 30, None), (17, 41, 3, 5, 12, None), (24, 24, 3, 10, 14, (5, 3))]:
        f = ops.normalize_to_hwc(torch.randn(T, 256, H, W, generator=g).to(dev))
        h16 = ops.split_f16x2(f)
        rows = [(1, 0, True), (2, 0, True), (2, 1, True), (T - 1, 0, False)]
        pairs = ops.make_pairs(rows, dev)
        mask = ops.MaskSpec.from_neighbor_range(nr)
        if rect is not None:
            mask = ops.MaskSpec(r2max=mask.r2max, ry=rect[0], rx=rect[1])
        outs = []
        for dbg in (0, 1024):
            ops.set_option("pair_f16_debug", dbg)
            try:
                outs.append(ops.pair_topk_split(h16, h16, pairs, H, W, H, W, mask, topk, fmt="f16"))
            finally:
                ops.set_option("pair_f16_debug", 0)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (H, W, topk)
    assert not ops.pair_f16x3_timed_out()
    # block list beyond the three-role capacity: an unmasked pair scans the whole 260 x 260 key grid
    H = W = 260
    f = ops.normalize_to_hwc(torch.randn(2, 256, H, W, generator=g).to(dev))
    pairs = ops.make_pairs([(1, 0, False)], dev)
    mask = ops.MaskSpec.from_neighbor_range(30)
    i16, s16 = ops.pair_topk_split(ops.split_f16x2(f), ops.split_f16x2(f), pairs, H, W, H, W, mask, 10, fmt="f16")
    i32, s32 = ops.pair_topk(f, f, pairs, H, W, H, W, mask, 10)
    assert float((s16 - s32).abs().max()) < 2.5e-6
    assert float((i16 == i32).all(-1).float().mean()) > 0.995
    assert not ops.pair_f16x3_timed_out()


def test_split_pair_topk_rejects_what_it_cannot_do(dev):
    from fgvc_amd import ops, _lib
    f = ops.normalize_to_hwc(torch.randn(1, 128, 8, 8).to(dev))
    hl = ops.split_bf16(f)
    pairs = ops.make_pairs([(0, 0, True)], dev)
    with pytest.raises(_lib.FgvcHipError):
        ops.pair_topk_split(hl, hl, pairs, 8, 8, 8, 8, ops.MaskSpec.from_neighbor_range(6), 10)      # C != 256
    f = ops.normalize_to_hwc(torch.randn(1, 256, 8, 8).to(dev))
    hl = ops.split_bf16(f)
    with pytest.raises(_lib.FgvcHipError):
        ops.pair_topk_split(hl, hl, pairs, 8, 8, 8, 8, ops.MaskSpec.from_neighbor_range(6), 11)      # topk > 10
    with pytest.raises(_lib.FgvcHipError):
        ops.pair_topk_split(hl, hl, pairs, 8, 8, 8, 8, ops.MaskSpec.from_neighbor_range(6), 11, fmt="f16")
    assert not ops.split_path_ok(256, 8, 8, 10, normalized=False)
    assert not ops.split_path_ok(64, 8, 8, 10, normalized=True)
    assert ops.split_path_ok(256, 120, 214, 10, normalized=True)
    assert not ops.split_path_ok(256, 720, 1280, 10, normalized=True)         # more than 4096 key blocks per frame
    # auto falls back to the f32 kernel where the split path does not apply
    i, s = ops.pair_topk_auto(f, f, pairs, 8, 8, 8, 8, ops.MaskSpec.from_neighbor_range(6), 10, normalized=False)
    i2, s2 = ops.pair_topk(f, f, pairs, 8, 8, 8, 8, ops.MaskSpec.from_neighbor_range(6), 10)
    assert torch.equal(i, i2) and torch.equal(s, s2)


def test_engine_split_and_f32_paths_agree(dev):
    """The tracker's default pair kernel (split bf16) against the f32-MFMA kernel on a whole clip."""
    from fgvc_amd import engine, ops
    g = torch.Generator().manual_seed(12)
    T_, H, W = 7, 24, 40
    feats = ops.normalize_to_hwc(torch.randn(T_, 256, H, W, generator=g).to(dev))
    qp = torch.tensor([[0, 40.0, 30.0], [0, 100.0, 60.0], [2, 80.0, 20.0]])
    outs = {}
    for prec in ("f32", "split", "auto"):
        cfg = engine.TrackerConfig(pair_precision=prec, regroup=True)
        outs[prec] = engine.track_points(feats, H, W, H * 4, W * 4, qp, cfg)[0]
    assert torch.equal(outs["split"], outs["auto"])
    assert float((outs["split"] - outs["f32"]).abs().max()) < 1e-3


# ---------------------------------------------------------------------------------------------------------------
# encoder convolutions on the bf16 pipe (fgvc_conv_split_f32)
# ---------------------------------------------------------------------------------------------------------------
def _padded_to_nchw(t, H, W):
    return t[:, 1:H + 1, 1:W + 1, :].permute(0, 3, 1, 2).contiguous()


def _nhwc_to_nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _split_to_nchw(t, H, W):
    N, Hp, Wp, nch, _ = t.shape
    v = t.view(torch.bfloat16).float().view(N, Hp, Wp, nch, 2, 32)
    x = (v[..., 0, :] + v[..., 1, :]).reshape(N, Hp, Wp, nch * 32)
    return _padded_to_nchw(x, H, W)


@pytest.mark.parametrize("case", [(2, 128, 256, 3, 19, 45, True, True), (1, 256, 256, 3, 8, 32, False, True),
                                  (2, 128, 256, 1, 13, 70, False, False), (1, 64, 512, 3, 24, 33, True, True),
                                  (2, 64, 64, 3, 21, 50, True, True), (1, 64, 128, 3, 9, 40, False, True),
                                  (1, 128, 128, 3, 17, 31, True, True), (1, 64, 128, 1, 8, 8, False, False),
                                  (1, 32, 192, 3, 10, 34, True, False)])
def test_conv_split_vs_torch(dev, case):
    """conv -> BN(eval) [-> + identity] [-> ReLU] against torch in float64: ragged sizes, 3x3 and 1x1, every
    output-channel tiling (64 / 128 / 192 = 3 x 64 / 256 / 512 = 2 x 256)."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    N, Cin, Cout, KS, H, W, with_res, relu = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case))
    x = torch.randn(N, Cin, H, W, generator=g)
    wt = torch.randn(Cout, Cin, KS, KS, generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
    bn = torch.nn.BatchNorm2d(Cout).eval()
    bn.weight.data = torch.rand(Cout, generator=g) + 0.5
    bn.bias.data = torch.randn(Cout, generator=g) * 0.1
    bn.running_mean = torch.randn(Cout, generator=g) * 0.1
    bn.running_var = torch.rand(Cout, generator=g) + 0.5
    res = torch.randn(N, Cout, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), wt.double(), padding=KS // 2)
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().view(1, -1, 1, 1)
    ref = (ref - bn.running_mean.double().view(1, -1, 1, 1)) * sc + bn.bias.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    ref = ref.detach()

    wp, bias = ops.prepare_conv_split(wt.to(dev), bn.to(dev))
    xs = ops.nchw_to_split_nhwc(x.to(dev))
    assert float((_split_to_nchw(xs.cpu(), H, W) - x).abs().max()) < 1e-5 * float(x.abs().max())
    out_s = ops.alloc_split_nhwc(N, Cout, H, W, dev)
    out_f = ops.alloc_nhwc(N, Cout, H, W, dev)
    resp = res.permute(0, 2, 3, 1).contiguous().to(dev) if with_res else None
    ops.conv_split(xs, wp, bias, H, W, relu, residual=resp, out_split=out_s, out_f32=out_f)
    got_f = _nhwc_to_nchw(out_f.cpu()).double()
    got_s = _split_to_nchw(out_s.cpu(), H, W).double()
    scale = float(ref.detach().abs().max())
    assert float((got_f - ref).abs().max()) < 2e-5 * scale, float((got_f - ref).abs().max()) / scale
    assert float((got_s - ref).abs().max()) < 3e-5 * scale
    # nothing outside the interior of the padded split tensor was touched
    assert int(out_s[:, :, W + 1:].abs().max()) == 0 and int(out_s[:, H + 1:].abs().max()) == 0
    assert int(out_s[:, 0].abs().max()) == 0 and int(out_s[:, :, 0].abs().max()) == 0
    # dense NHWC -> split with the fused in-place ReLU
    t = torch.randn(N, H, W, Cout, generator=g).to(dev)
    want_t = t.clamp_min(0).cpu()
    ts = ops.nhwc_to_split(t, ops.alloc_split_nhwc(N, Cout, H, W, dev), relu=True)
    assert torch.equal(t.cpu(), want_t)
    assert float((_split_to_nchw(ts.cpu(), H, W) - _nhwc_to_nchw(want_t)).abs().max()) < 1e-5 * float(want_t.abs().max())
    # the normalised read-back
    nf = ops.normalize_nhwc(out_f)
    want = torch.nn.functional.normalize(got_f.float().permute(0, 2, 3, 1).reshape(N, H * W, Cout), dim=2)
    assert torch.allclose(nf.cpu(), want, atol=1e-6)


def _pack_act(x_nchw, fmt, scale_log2, dev):
    """f32 NCHW -> padded split NHWC in the f16 activation formats (test-side restatement of the kernels' epilogue: csrc/common.hpp
    split_f16_4): [h = f16(s x) | l8 = e4m3(8 l) | h8 = e4m3(h / 128)] (ACT_F16F8) or [h | l = f16(s x - h)] (ACT_F16X2)."""
    from fgvc_amd import ops
    N, C, H, W = x_nchw.shape
    xs = (x_nchw.permute(0, 2, 3, 1).float() * 2.0 ** scale_log2).reshape(N, H, W, C // 32, 32).contiguous()
    h = xs.to(torch.float16)
    l = xs - h.float()
    if fmt == ops.ACT_F16X2:
        row = torch.cat([h.view(torch.uint8), l.to(torch.float16).view(torch.uint8)], -1)
    elif fmt == ops.ACT_F16F6:            # the oracle's model of split_f16f6_chunk: FP6 blocks with their own scales
        from oracle import fgvc_oracle as O
        v = x_nchw.permute(0, 2, 3, 1).float().reshape(-1, 32).numpy()
        row = torch.from_numpy(O.act_f16f6_rows(v, scale_log2)).reshape(N, H, W, C // 32, 128)
    else:
        l8 = (l * 2.0 ** ops.F8_BX).to(torch.float8_e4m3fn).view(torch.uint8)
        h8 = (h.float() * 2.0 ** -ops.F8_AX).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        row = torch.cat([h.view(torch.uint8), l8, h8], -1)
    out = ops.alloc_split_nhwc(N, C, H, W, dev)
    out[:, 1:H + 1, 1:W + 1] = row.contiguous().view(torch.int16).to(dev)
    return out


@pytest.mark.parametrize("arith", ["f16f8", "f16f6", "f16x3"])
@pytest.mark.parametrize("case", [(2, 128, 256, 3, 19, 45, True, True), (1, 256, 256, 3, 8, 32, False, True),
                                  (2, 128, 256, 1, 13, 70, False, False), (1, 64, 128, 3, 9, 40, False, True),
                                  (1, 128, 128, 3, 17, 31, True, True), (1, 32, 192, 3, 10, 34, True, False)])
def test_conv_split_f16_forms_vs_torch(dev, case, arith):
    """fgvc_conv_split_fmt_f32 in the three f16 arithmetics (input, weights AND split output in the f16 formats) against torch in
    float64 of the EXACT f32 operands: the bound therefore covers the formats' own quantisation of input and weights.
    ReLU-like heavy-tailed activations (the case the static per-tensor scale has to survive).  Bounds: f16x3 5e-6 of max|y|,
    f16f8 3e-5 (simulated: 1.0-1.7e-5 on real activations, tools/sim_conv_formats.py; bf16x3's bound is 2e-5)."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    N, Cin, Cout, KS, H, W, with_res, relu = case
    fmt = ops.ACT_FMT[arith]
    g = torch.Generator().manual_seed(sum(int(v) for v in case) + fmt)
    x = torch.randn(N, Cin, H, W, generator=g).abs() ** 1.5 * (torch.rand(N, Cin, H, W, generator=g) > 0.4)     # 40 % zeros, a long tail
    if Cin >= 64:
        x[:, 32:64] = 0                                                      # a whole 32-channel chunk of zeros: all-zero FP6 blocks on the way in
    wt = torch.randn(Cout, Cin, KS, KS, generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
    bn = torch.nn.BatchNorm2d(Cout).eval()
    bn.weight.data = torch.rand(Cout, generator=g) * 1.5 + 0.2
    bn.bias.data = torch.randn(Cout, generator=g) * 0.1
    if relu and not with_res:
        bn.bias.data[32:64] = -1e3                                           # ... and one the ReLU zeroes on the way out (the scale byte of an empty block)
    bn.running_mean = torch.randn(Cout, generator=g) * 0.1
    bn.running_var = torch.rand(Cout, generator=g) + 0.5
    res = torch.randn(N, Cout, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), wt.double(), padding=KS // 2)
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().view(1, -1, 1, 1)
    ref = (ref - bn.running_mean.double().view(1, -1, 1, 1)) * sc + bn.bias.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    ref = ref.detach()
    scale = float(ref.abs().max())
    sx, so = ops.act_scale_log2(float(x.abs().max())), ops.act_scale_log2(scale)
    wp, bias, sw = ops.prepare_conv_split_f16(wt.to(dev), bn.to(dev), fmt)
    xs = _pack_act(x, fmt, sx, dev)
    assert float((_padded_to_nchw(ops.unsplit_act(xs.cpu(), fmt, sx), H, W) - x).abs().max()) < (3e-7 if fmt == ops.ACT_F16X2 else 4e-5) * float(x.abs().max())
    out_s = ops.alloc_split_nhwc(N, Cout, H, W, dev)
    out_f = ops.alloc_nhwc(N, Cout, H, W, dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    resp = res.permute(0, 2, 3, 1).contiguous().to(dev) if with_res else None
    ops.conv_split(xs, wp, bias, H, W, relu, residual=resp, out_split=out_s, out_f32=out_f, in_fmt=fmt, in_scale_log2=sx + sw, out_fmt=fmt,
                   out_scale_log2=so, overflow=ovf)
    got_f = _nhwc_to_nchw(out_f.cpu()).double()
    got_s = _padded_to_nchw(ops.unsplit_act(out_s.cpu(), fmt, so), H, W).double()
    tol = 5e-6 if arith == "f16x3" else 3e-5
    err = float((got_f - ref).abs().max()) / scale
    assert err < tol, err
    if arith == "f16f6":      # the split output, byte for byte what the oracle's model makes of the kernel's own f32 output: h, both FP6
        from oracle import fgvc_oracle as O          # blocks as the matrix instruction decodes them, and the scale bytes
        want = O.act_f16f6_rows(out_f.cpu().reshape(-1, 32).numpy(), so)
        got = out_s[:, 1:H + 1, 1:W + 1].contiguous().cpu().view(torch.uint8).reshape(-1, 128).numpy()
        for a_, b_ in zip(O.act_f16f6_decode(got), O.act_f16f6_decode(want)):
            assert np.array_equal(a_, b_)
        assert np.array_equal(got[:, 104], want[:, 104]) and np.array_equal(got[:, 120], want[:, 120])
        assert int(got[:, 105:112].max()) == 0 and int(got[:, 121:128].max()) == 0
    assert float((got_s - got_f).abs().max()) < (1e-6 if arith == "f16x3" else 4e-5) * scale          # the output format's own rounding
    assert int(ovf.item()) == 0
    assert int(out_s[:, :, W + 1:].abs().max()) == 0 and int(out_s[:, H + 1:].abs().max()) == 0
    assert int(out_s[:, 0].abs().max()) == 0 and int(out_s[:, :, 0].abs().max()) == 0
    # the bf16 form of the SAME layer from the same kernel family agrees (cross-check of layouts and scales, three ways)
    wp0, bias0 = ops.prepare_conv_split(wt.to(dev), bn.to(dev))
    out0 = ops.alloc_nhwc(N, Cout, H, W, dev)
    ops.conv_split(ops.nchw_to_split_nhwc(x.to(dev)), wp0, bias0, H, W, relu, residual=resp, out_f32=out0)
    assert float((out0 - out_f).abs().max()) < (tol + 2e-5) * scale
    # a mixed layer: bf16 input, f16-format output (the first convolution behind layer 1) -- and the overflow flag
    ops.conv_split(ops.nchw_to_split_nhwc(x.to(dev)), wp0, bias0, H, W, relu, residual=resp, out_split=out_s, out_fmt=fmt, out_scale_log2=so,
                   overflow=ovf)
    assert float((_padded_to_nchw(ops.unsplit_act(out_s.cpu(), fmt, so), H, W).double() - ref).abs().max()) < 6e-5 * scale and int(ovf.item()) == 0
    ops.conv_split(xs, wp, bias, H, W, relu, residual=resp, out_split=out_s, in_fmt=fmt, in_scale_log2=sx + sw, out_fmt=fmt,
                   out_scale_log2=so + 9, overflow=ovf)                                                  # 2^17 times the largest value: beyond f16
    assert int(ovf.item()) == 1


@pytest.mark.parametrize("case", [(2, 64, 128, 3, 24, 43, True), (1, 64, 128, 1, 24, 43, False), (2, 128, 256, 3, 17, 66, True),
                                  (1, 128, 256, 1, 17, 66, False), (1, 32, 96, 3, 9, 130, True), (1, 64, 64, 3, 5, 5, False),
                                  (1, 64, 32, 1, 8, 7, False)])
def test_conv_s2_split_vs_torch(dev, case):
    """stride-2 3x3 (pad 1) and 1x1 (pad 0) conv -> BN(eval) [-> ReLU] against torch in float64: odd and even sizes (the last
    input row / column is or is not read), one and several chunk groups (Cin 32 / 64 / 128), ragged output tiles, Cout
    that does not fill the workgroup's 128 channels, inputs whose padded buffer is smaller than the tiles' reach."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    N, Cin, Cout, KS, H, W, relu = case
    g = torch.Generator().manual_seed(100 + sum(int(v) for v in case))
    x = torch.randn(N, Cin, H, W, generator=g)
    wt = torch.randn(Cout, Cin, KS, KS, generator=g) * (2.0 / (Cin * KS * KS)) ** 0.5
    bn = torch.nn.BatchNorm2d(Cout).eval()
    bn.weight.data = torch.rand(Cout, generator=g) + 0.5
    bn.bias.data = torch.randn(Cout, generator=g) * 0.1
    bn.running_mean = torch.randn(Cout, generator=g) * 0.1
    bn.running_var = torch.rand(Cout, generator=g) + 0.5
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=KS // 2)
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().view(1, -1, 1, 1)
    ref = (ref - bn.running_mean.double().view(1, -1, 1, 1)) * sc + bn.bias.double().view(1, -1, 1, 1)
    if relu:
        ref = ref.clamp_min(0)
    ref = ref.detach()
    Ho, Wo = ref.shape[-2:]
    assert (Ho, Wo) == ((H - 1) // 2 + 1, (W - 1) // 2 + 1)
    wp, bias = ops.prepare_conv_s2(wt.to(dev), bn.to(dev))
    xs = ops.nchw_to_split_nhwc(x.to(dev))
    out_s = ops.alloc_split_nhwc(N, Cout, Ho, Wo, dev)
    out_f = ops.alloc_nhwc(N, Cout, Ho, Wo, dev)
    ops.conv_s2_split(xs, wp, bias, H, W, relu, out_split=out_s, out_f32=out_f)
    got_f = _nhwc_to_nchw(out_f.cpu()).double()
    got_s = _split_to_nchw(out_s.cpu(), Ho, Wo).double()
    scale = float(ref.abs().max())
    assert float((got_f - ref).abs().max()) < 2e-5 * scale, float((got_f - ref).abs().max()) / scale
    assert float((got_s - ref).abs().max()) < 3e-5 * scale
    assert int(out_s[:, :, Wo + 1:].abs().max()) == 0 and int(out_s[:, Ho + 1:].abs().max()) == 0
    assert int(out_s[:, 0].abs().max()) == 0 and int(out_s[:, :, 0].abs().max()) == 0
    # either output alone
    only_f = ops.alloc_nhwc(N, Cout, Ho, Wo, dev)
    ops.conv_s2_split(xs, wp, bias, H, W, relu, out_f32=only_f)
    only_s = ops.alloc_split_nhwc(N, Cout, Ho, Wo, dev)
    ops.conv_s2_split(xs, wp, bias, H, W, relu, out_split=only_s)
    assert torch.equal(only_f, out_f) and torch.equal(only_s, out_s)
    # the split output in the f16 forms the stride-1 kernel behind it reads (what the encoder asks of the first block of a stage)
    so = ops.act_scale_log2(scale)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    for fmt, tol in ((ops.ACT_F16F8, 4e-5), (ops.ACT_F16F6, 4e-5), (ops.ACT_F16X2, 1e-6)):
        o = ops.alloc_split_nhwc(N, Cout, Ho, Wo, dev)
        ops.conv_s2_split(xs, wp, bias, H, W, relu, out_split=o, out_fmt=fmt, out_scale_log2=so, overflow=ovf)
        assert float((_padded_to_nchw(ops.unsplit_act(o.cpu(), fmt, so), Ho, Wo).double() - got_f).abs().max()) < tol * scale
        assert int(ovf.item()) == 0
        assert int(o[:, :, Wo + 1:].abs().max()) == 0 and int(o[:, Ho + 1:].abs().max()) == 0
        if fmt == ops.ACT_F16F6:       # ... as the oracle's model of the format makes them from the kernel's f32 output
            from oracle import fgvc_oracle as O
            want = O.act_f16f6_rows(out_f.cpu().reshape(-1, 32).numpy(), so)
            got = o[:, 1:Ho + 1, 1:Wo + 1].contiguous().cpu().view(torch.uint8).reshape(-1, 128).numpy()
            for a_, b_ in zip(O.act_f16f6_decode(got), O.act_f16f6_decode(want)):
                assert np.array_equal(a_, b_)
            assert np.array_equal(got[:, 104], want[:, 104]) and np.array_equal(got[:, 120], want[:, 120])


@pytest.mark.parametrize("case", [(2, 21, 50, True, True), (1, 4, 32, False, True), (1, 3, 5, True, False), (3, 61, 100, True, True),
                                  (2, 240, 427, False, True)])
def test_conv64_split_vs_torch(dev, case):
    """64 -> 64 3x3 conv -> BN(eval) [-> + identity] [-> ReLU] with register-resident weights against torch in float64:
    ragged tiles, fewer tiles than workgroups, many tiles per persistent workgroup (both LDS buffers, prefetch chain)."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    N, H, W, with_res, relu = case
    g = torch.Generator().manual_seed(300 + N + H + W)
    x = torch.randn(N, 64, H, W, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    bn = torch.nn.BatchNorm2d(64).eval()
    bn.weight.data = torch.rand(64, generator=g) + 0.5
    bn.bias.data = torch.randn(64, generator=g) * 0.1
    bn.running_mean = torch.randn(64, generator=g) * 0.1
    bn.running_var = torch.rand(64, generator=g) + 0.5
    res = torch.randn(N, 64, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().view(1, -1, 1, 1)
    ref = (ref - bn.running_mean.double().view(1, -1, 1, 1)) * sc + bn.bias.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    ref = ref.detach()
    wp, bias = ops.prepare_conv64(wt.to(dev), bn.to(dev))
    xs = ops.nchw_to_split_nhwc(x.to(dev))
    out_s = ops.alloc_split_nhwc(N, 64, H, W, dev)
    out_f = ops.alloc_nhwc(N, 64, H, W, dev)
    resp = res.permute(0, 2, 3, 1).contiguous().to(dev) if with_res else None
    ops.conv64_split(xs, wp, bias, H, W, relu, residual=resp, out_split=out_s, out_f32=out_f)
    got_f = _nhwc_to_nchw(out_f.cpu()).double()
    got_s = _split_to_nchw(out_s.cpu(), H, W).double()
    scale = float(ref.abs().max())
    assert float((got_f - ref).abs().max()) < 2e-5 * scale, float((got_f - ref).abs().max()) / scale
    assert float((got_s - ref).abs().max()) < 3e-5 * scale
    assert int(out_s[:, :, W + 1:].abs().max()) == 0 and int(out_s[:, H + 1:].abs().max()) == 0
    assert int(out_s[:, 0].abs().max()) == 0 and int(out_s[:, :, 0].abs().max()) == 0
    # same bits as the generic kernel (same products in the same order)
    wg, bg = ops.prepare_conv_split(wt.to(dev), bn.to(dev))
    gen_f = ops.alloc_nhwc(N, 64, H, W, dev)
    ops.conv_split(xs, wg, bg, H, W, relu, residual=resp, out_f32=gen_f)
    assert float((gen_f - out_f).abs().max()) < 1e-5 * scale
    only_s = ops.alloc_split_nhwc(N, 64, H, W, dev)
    ops.conv64_split(xs, wp, bias, H, W, relu, residual=resp, out_split=only_s)
    assert torch.equal(only_s, out_s)
    if with_res:
        # the identity given as a split tensor (what the encoder does in layer 1): bit-identical to the f32 identity hi + lo
        rs = ops.nchw_to_split_nhwc(res.to(dev))
        r32 = ops.unsplit_act(rs, ops.ACT_BF16X2)[:, 1:H + 1, 1:W + 1].contiguous()
        assert float((r32 - resp).abs().max()) <= 2.0 ** -16 * float(resp.abs().max())
        a_s, a_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
        b_s, b_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
        ops.conv64_split(xs, wp, bias, H, W, relu, residual=r32, out_split=a_s, out_f32=a_f)
        ops.conv64_split(xs, wp, bias, H, W, relu, residual_split=rs, out_split=b_s, out_f32=b_f)
        assert torch.equal(a_s, b_s) and torch.equal(a_f, b_f)
        with pytest.raises(Exception):
            ops.conv64_split(xs, wp, bias, H, W, relu, residual=r32, residual_split=rs, out_split=b_s)


@pytest.mark.parametrize("case", [(2, 21, 50, True, True), (1, 4, 32, False, True), (1, 3, 5, True, False), (2, 120, 214, True, True)])
def test_conv64_f16f8_vs_torch(dev, case):
    """fgvc_conv64_split_fmt_f32 in the f16 + fp8 arithmetic (input, register-resident weights AND split output in the f16f8 format)
    against torch in float64 of the exact f32 operands, heavy-tailed ReLU-like input: the bound (3e-5 of max|y|, the wide layers' bound)
    covers the format's quantisation of input and weights.  Also: bf16 split output from f16f8 input (what layer 1's last block hands
    to the stride-2 convolutions), agreement with the generic kernel in the same arithmetic, the overflow flag."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    N, H, W, with_res, relu = case
    fmt = ops.ACT_F16F8
    g = torch.Generator().manual_seed(640 + N + H + W)
    x = torch.randn(N, 64, H, W, generator=g).abs() ** 1.5 * (torch.rand(N, 64, H, W, generator=g) > 0.4)
    wt = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    bn = torch.nn.BatchNorm2d(64).eval()
    bn.weight.data = torch.rand(64, generator=g) * 1.5 + 0.2
    bn.bias.data = torch.randn(64, generator=g) * 0.1
    bn.running_mean = torch.randn(64, generator=g) * 0.1
    bn.running_var = torch.rand(64, generator=g) + 0.5
    res = torch.randn(N, 64, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().view(1, -1, 1, 1)
    ref = (ref - bn.running_mean.double().view(1, -1, 1, 1)) * sc + bn.bias.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    ref = ref.detach()
    scale = float(ref.abs().max())
    sx, so = ops.act_scale_log2(float(x.abs().max())), ops.act_scale_log2(scale)
    wp, bias, sw = ops.prepare_conv64_f16(wt.to(dev), bn.to(dev))
    xs = _pack_act(x, fmt, sx, dev)
    out_s, out_f = ops.alloc_split_nhwc(N, 64, H, W, dev), ops.alloc_nhwc(N, 64, H, W, dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    resp = res.permute(0, 2, 3, 1).contiguous().to(dev) if with_res else None
    ops.conv64_split(xs, wp, bias, H, W, relu, residual=resp, out_split=out_s, out_f32=out_f, in_fmt=fmt, in_scale_log2=sx + sw, out_fmt=fmt,
                     out_scale_log2=so, overflow=ovf)
    got_f = _nhwc_to_nchw(out_f.cpu()).double()
    got_s = _padded_to_nchw(ops.unsplit_act(out_s.cpu(), fmt, so), H, W).double()
    err = float((got_f - ref).abs().max()) / scale
    assert err < 3e-5, err
    assert float((got_s - got_f).abs().max()) < 4e-5 * scale and int(ovf.item()) == 0
    assert int(out_s[:, :, W + 1:].abs().max()) == 0 and int(out_s[:, H + 1:].abs().max()) == 0
    assert int(out_s[:, 0].abs().max()) == 0 and int(out_s[:, :, 0].abs().max()) == 0
    # the generic kernel in the same arithmetic on the same operands: same products, another accumulation order
    wg, bg, swg = ops.prepare_conv_split_f16(wt.to(dev), bn.to(dev), fmt)
    gen_f = ops.alloc_nhwc(N, 64, H, W, dev)
    ops.conv_split(xs, wg, bg, H, W, relu, residual=resp, out_f32=gen_f, in_fmt=fmt, in_scale_log2=sx + swg)
    assert swg == sw and float((gen_f - out_f).abs().max()) < 2e-6 * scale
    # bf16 split output from the f16f8 input
    out_b = ops.alloc_split_nhwc(N, 64, H, W, dev)
    ops.conv64_split(xs, wp, bias, H, W, relu, residual=resp, out_split=out_b, in_fmt=fmt, in_scale_log2=sx + sw)
    got_b = _split_to_nchw(out_b.cpu(), H, W).double()
    assert float((got_b - got_f).abs().max()) < 1e-5 * scale
    # the overflow flag
    ops.conv64_split(xs, wp, bias, H, W, relu, residual=resp, out_split=out_s, in_fmt=fmt, in_scale_log2=sx + sw, out_fmt=fmt,
                     out_scale_log2=so + 9, overflow=ovf)
    assert int(ovf.item()) == 1


@pytest.mark.parametrize("case", [(2, 64, 96, True), (1, 37, 131, True), (3, 9, 5, False), (1, 480, 854, True)])
def test_stem7_split_vs_torch(dev, case):
    """7x7 / stride 2 / pad 3 stem -> BN(eval) [-> ReLU] against torch in float64: even and odd sizes, images smaller than
    a tile, several tiles per persistent workgroup (the 480p frame: 840 tiles on 512 workgroups)."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    N, H, W, relu = case
    g = torch.Generator().manual_seed(200 + H + W)
    x = torch.randn(N, 3, H, W, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5
    bn = torch.nn.BatchNorm2d(64).eval()
    bn.weight.data = torch.rand(64, generator=g) + 0.5
    bn.bias.data = torch.randn(64, generator=g) * 0.1
    bn.running_mean = torch.randn(64, generator=g) * 0.1
    bn.running_var = torch.rand(64, generator=g) + 0.5
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=3)
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().view(1, -1, 1, 1)
    ref = (ref - bn.running_mean.double().view(1, -1, 1, 1)) * sc + bn.bias.double().view(1, -1, 1, 1)
    if relu:
        ref = ref.clamp_min(0)
    ref = ref.detach()
    Ho, Wo = ref.shape[-2:]
    wp, bias = ops.prepare_stem7(wt.to(dev), bn.to(dev))
    out_s = ops.alloc_split_nhwc(N, 64, Ho, Wo, dev)
    out_f = ops.alloc_nhwc(N, 64, Ho, Wo, dev)
    ops.stem7_split(x.to(dev), wp, bias, relu, out_split=out_s, out_f32=out_f)
    got_f = _nhwc_to_nchw(out_f.cpu()).double()
    got_s = _split_to_nchw(out_s.cpu(), Ho, Wo).double()
    scale = float(ref.abs().max())
    assert float((got_f - ref).abs().max()) < 2e-5 * scale, float((got_f - ref).abs().max()) / scale
    assert float((got_s - ref).abs().max()) < 3e-5 * scale
    assert int(out_s[:, :, Wo + 1:].abs().max()) == 0 and int(out_s[:, Ho + 1:].abs().max()) == 0
    assert int(out_s[:, 0].abs().max()) == 0 and int(out_s[:, :, 0].abs().max()) == 0
    only_f = ops.alloc_nhwc(N, 64, Ho, Wo, dev)
    ops.stem7_split(x.to(dev), wp, bias, relu, out_f32=only_f)
    assert torch.equal(only_f, out_f)
    # the split output in the f16 + fp8 form (what layer 1 reads when the trunk computes in f16f8), and its overflow flag
    so = ops.act_scale_log2(scale)
    out_8 = ops.alloc_split_nhwc(N, 64, Ho, Wo, dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.stem7_split(x.to(dev), wp, bias, relu, out_split=out_8, out_fmt=ops.ACT_F16F8, out_scale_log2=so, overflow=ovf)
    got_8 = _padded_to_nchw(ops.unsplit_act(out_8.cpu(), ops.ACT_F16F8, so), Ho, Wo).double()
    assert float((got_8 - got_f).abs().max()) < 4e-5 * scale and int(ovf.item()) == 0
    assert int(out_8[:, :, Wo + 1:].abs().max()) == 0 and int(out_8[:, 0].abs().max()) == 0
    ops.stem7_split(x.to(dev), wp, bias, relu, out_split=out_8, out_fmt=ops.ACT_F16F8, out_scale_log2=so + 9, overflow=ovf)
    assert int(ovf.item()) == 1


def test_encoder_identities_from_the_split_form(dev):
    """ResNet.res_from_split (off by default: measured slower): layer 1 adds its identities as hi + lo of the split tensor its
    first convolution reads, and neither the stem nor the first block writes an f32 copy.  Same trunk within the split's 2^-17."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd.mmpt_api.backbones import ResNet
    torch.manual_seed(22)
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none")).to(dev).eval()
    x = torch.randn(3, 3, 72, 100, device=dev)
    with torch.no_grad():
        for arith in net.supported_arith():
            net.set_arith(arith)
            try:
                ResNet.conv64_f16f8 = False                      # (the option is for layer 1 in the bf16 form)
                net.reset_split_cache()
                ref = net(x).clone()
                ResNet.res_from_split = True
                net.reset_split_cache()
                got = net(x).clone()
            finally:
                ResNet.res_from_split = False
                ResNet.conv64_f16f8 = True
                net.reset_split_cache()
            net.check_overflow()
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (arith, float((got - ref).abs().max()), float(ref.abs().max()))
            assert not torch.equal(got, ref)                       # (the option did change the arithmetic)


def test_encoder_layer1_in_f16f8(dev):
    """ResNet.conv64_f16f8 (off by default: no gain end to end): with the trunk in f16f8, layer 1 runs fgvc_conv64_split_fmt_f32 in the
    f16 + fp8 form and the stem writes that form (calibrated scales, overflow word).  Same trunk within the arithmetic's bound."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import ops
    from fgvc_amd.mmpt_api.backbones import ResNet
    torch.manual_seed(23)
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none")).to(dev).eval()
    x = torch.randn(3, 3, 72, 100, device=dev)
    with torch.no_grad():
        try:
            ResNet.use_split_conv = False
            ref = net(x).clone()                                    # MIOpen f32
        finally:
            ResNet.use_split_conv = True
        scale = float(ref.abs().max())
        for arith in ("f16f8", "f16f6"):                             # (layer 1 itself has one f16 form: f16 + fp8, under either trunk)
            net.set_arith(arith)
            try:
                ResNet.conv64_f16f8 = False
                net.reset_split_cache()
                assert net._format_plan(2)["stem"] == ops.ACT_BF16X2
                base = net(x).clone()
            finally:
                ResNet.conv64_f16f8 = True
                net.reset_split_cache()
            assert ResNet.conv64_f16f8 is True                       # the default since round 4 (+1.8 % of the step)
            plan = net._format_plan(2)
            assert plan["stem"] == ops.ACT_F16F8 and plan[(0, 0)] == (ops.ACT_F16F8,) * 3 and plan[(0, 1)][:2] == (ops.ACT_F16F8,) * 2
            assert plan[(0, 1)][2] == ops.ACT_BF16X2                 # the stride-2 convolutions of layer 2 read the bf16 form
            got = net(x).clone()
            net.check_overflow()
            assert float((got - ref).abs().max()) <= 5e-5 * scale and float((base - ref).abs().max()) <= 5e-5 * scale, (arith, float((got - ref).abs().max()) / scale)
            assert not torch.equal(got, base)
            print(f"layer 1 in f16 + fp8 under {arith}: trunk error {float((got - ref).abs().max()) / scale:.2e} of the largest feature (bf16 layer 1: {float((base - ref).abs().max()) / scale:.2e})")


def test_encoder_hip_graph_replay(dev):
    """ResNet.use_graph: forward_hwc replays a HIP graph of the whole trunk (two stream lanes, ~32 launches, the normalise + split pass)
    captured on the second call of an input shape.  Same bits as the eager path for new inputs, through `out=`, for a second shape;
    new weights drop the graphs."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import ops
    from fgvc_amd.mmpt_api.backbones import ResNet
    torch.manual_seed(24)
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none")).to(dev).eval()
    split_if = lambda C, H, W: True
    xs = [torch.randn(3, 3, 72, 100, device=dev) for _ in range(4)] + [torch.randn(2, 3, 64, 64, device=dev)]
    default_mode = ResNet.use_graph
    with torch.no_grad():
        ResNet.use_graph = False                               # the eager baseline
        want = [net.forward_hwc(x, True, split_if=split_if, split_fmt="f16")[0].clone() for x in xs]
        try:
            ResNet.use_graph = True
            got = []
            for x in xs + xs[:2]:                                  # first call of a shape eager, second captures, later ones replay
                f, H, W = net.forward_hwc(x, True, split_if=split_if, split_fmt="f16")
                got.append(f.clone())
            cache = net.__dict__["_split_cache"]
            graphs = [k for k, v in cache.items() if isinstance(k, tuple) and k and k[0] == "graph" and isinstance(v, tuple)]
            assert len(graphs) == 1                                # (the 2-frame shape was seen once: still "warm")
            for a, b in zip(got, want + want[:2]):
                assert torch.equal(a, b)
            out = torch.empty_like(want[0])
            f, H, W = net.forward_hwc(xs[3], True, split_if=split_if, split_fmt="f16", out=out)
            assert f is out and torch.equal(out, want[3]) and (H, W) == (18, 25)
            net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})       # any load drops the derived state, graphs included
            assert not any(isinstance(k, tuple) and k and k[0] == "graph" for k in net.__dict__.get("_split_cache", {}))
            f, _, _ = net.forward_hwc(xs[0], True, split_if=split_if, split_fmt="f16")
            assert torch.equal(f, want[0])
        finally:
            ResNet.use_graph = default_mode
    assert not net.check_overflow()


@pytest.mark.gpu
def test_encoder_hip_graph_in_the_tracker_and_invalidation(dev):
    """(i) a replayed graph writes ONE static output; VanillaTracker.get_feats_hwc keeps the chunks of a clip in a list and
    concatenates at the end, so with T > 2 batch_step every full chunk must come back as its own tensor (from the second video on, when
    the chunk shape replays); (ii) a graph bakes in the calibrated scales and the addresses of its shape's workspaces: calibrate(),
    an overflow and the eviction of the shape's workspaces must drop it."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd.mmpt_api.backbones import ResNet
    torch.manual_seed(31)
    test_cfg = api.ConfigDict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, step=512, with_first=True,
                              with_first_neighbor=True, batch_step=3)
    model = api.build_model(dict(type="VanillaTracker", backbone=dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,),
                                                                      pool_type="none", zero_init_residual=False)),
                            train_cfg=None, test_cfg=test_cfg).to(dev).eval()
    net = model.backbone
    vids = [torch.randn(8, 3, 64, 96, device=dev) for _ in range(3)]          # 8 frames at batch_step 3: chunks of 3, 3, 2
    default_mode = ResNet.use_graph
    with torch.no_grad():
        ResNet.use_graph = False                               # the eager baseline
        want = [model.get_feats_hwc(v, split=True)[0].clone() for v in vids]
        try:
            ResNet.use_graph = True
            for rnd in range(2):                                               # round 0 warms / captures, round 1 replays every full chunk
                for v, w in zip(vids, want):
                    got = model.get_feats_hwc(v, split=True)[0]
                    assert torch.equal(got, w), f"round {rnd}: a chunk was overwritten by a later replay"
            graphs = lambda: [k for k, e in net.__dict__["_split_cache"].items() if isinstance(k, tuple) and k and k[0] == "graph" and isinstance(e, tuple)]
            assert len(graphs()) >= 1
            # (ii-a) calibrate() changes the scales the kernels take as arguments: the graphs go
            net.calibrate(vids[0][:3])
            assert not graphs()
            for v, w in zip(vids[:2], want[:2]):
                assert torch.equal(model.get_feats_hwc(v, split=True)[0], w)
            assert graphs()
            # (ii-b) an overflow drops the scales AND the graphs (the retry must not replay the old ones)
            ovf = [e for k, e in net.__dict__["_split_cache"].items() if isinstance(k, tuple) and k and k[0] == "overflow"]
            assert ovf
            ovf[0].fill_(1)
            assert net.check_overflow() and not graphs()
            # (the retry after an overflow runs 2^4 further below the f16 top -- ResNet.check_overflow under the canonical calibration --:
            # the same features up to the fixed-scale e4m3 cross terms of layer 1; end_overflow_retry() returns to the canonical scales)
            from fgvc_amd import ops
            for v, w in zip(vids[:2], want[:2]):
                got = model.get_feats_hwc(v, split=True)[0]
                assert float((ops.f32_of_f16f6x(got) - ops.f32_of_f16f6x(w)).abs().max()) < 2e-4
            assert graphs()
            net.end_overflow_retry()
            assert not graphs() and "_headroom_extra" not in net.__dict__
            for v, w in zip(vids[:2], want[:2]):
                assert torch.equal(model.get_feats_hwc(v, split=True)[0], w)
            assert graphs()
            # (ii-c) more input shapes than max_workspace_shapes: the evicted shape's graph goes with its workspaces, and the shape still
            # computes the same features when it comes back
            g0 = set(graphs())
            for hw in ((48, 64), (40, 72), (56, 56)):
                for _ in range(2):
                    net.forward_hwc(torch.randn(3, 3, *hw, device=dev), True, split_if=lambda C, H, W: True, split_fmt="f16")
            assert not (g0 & set(graphs()))
            for v, w in zip(vids, want):
                assert torch.equal(model.get_feats_hwc(v, split=True)[0], w)
        finally:
            ResNet.use_graph = default_mode
    torch.cuda.synchronize()
    assert not net.check_overflow()


@pytest.mark.gpu
def test_encoder_zero_initialised_residual_branch(dev):
    """init_weights() of the reference's ResNet zeroes bn2.weight of every block (zero_init_residual, mmpt/models/backbones/resnet.py
    :596-601): folded conv2 weights are all zero, so is a block's pre-residual tensor.  The f16 forms take a scale from the largest
    magnitude of a tensor: an all-zero one must get a finite scale and the trunk must equal the MIOpen path."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd.mmpt_api.backbones import ResNet
    torch.manual_seed(21)
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none"))
    net.init_weights()
    assert all(float(b.conv2.bn.weight.abs().max()) == 0.0 for b in net.layer3)
    net = net.to(dev).eval()
    x = torch.randn(2, 3, 64, 96, device=dev)
    with torch.no_grad():
        try:
            ResNet.use_split_conv = False
            ref = net(x).cpu()
        finally:
            ResNet.use_split_conv = True
        for arith in net.supported_arith():
            net.set_arith(arith)
            a = net(x).cpu()
            net.check_overflow()
            assert torch.isfinite(a).all()
            assert float((a - ref).abs().max()) <= 5e-5 * float(ref.abs().max()) + 1e-6, (arith, float((a - ref).abs().max()), float(ref.abs().max()))


@pytest.mark.parametrize("arith", ["f16f8", "f16f6", "bf16x3", "f16x3"])
def test_encoder_split_conv_stage_vs_miopen_and_oracle(dev, arith):
    """A1: the ResNet-18 trunk on the hand-written kernels (the default on the GPU), in each arithmetic of the wide layers, against
    the same network with every convolution in MIOpen (f32), against the CPU oracle network, and through the tracker's forward_hwc
    fast path.  Whole-trunk bounds relative to the largest feature: bf16x3 / f16x3 2e-5 (round 2's bound), f16f8 (two pipe units
    instead of three) 5e-5; normalised features 8e-6 / 1.5e-5 (measured against a float64 network: 4-6e-6 / 1.0-1.3e-5,
    tools/experiments/res_split_precision.py; layer 1 adds its identities from the split form, which changes none of these)."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd.mmpt_api.backbones import ResNet
    g = torch.Generator().manual_seed(14)
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none"))
    ora = O.ResNet18((1, 2, 1, 1), 2, "none")
    sd = O.seeded_resnet_state(9, (1, 2, 1, 1), "none")
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        elif k.endswith("bn.weight"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    net.load_state_dict(sd)
    ora.load_state_dict(sd)
    net = net.to(dev).eval()
    net.set_arith(arith)
    tol, tol_n = (5e-5, 1.5e-5) if arith in ("f16f8", "f16f6") else (2e-5, 8e-6)
    x = torch.randn(3, 3, 76, 132, generator=g)              # features 19 x 33: ragged against the 8 x 32 tiles
    with torch.no_grad():
        assert net._split_stage_ok(net.layer3, torch.empty(1, 128, 4, 4, device=dev))
        a = net(x.to(dev)).cpu()
        hw, Hf, Wf = net.forward_hwc(x.to(dev), True)
        try:
            ResNet.use_split_conv = False
            b = net(x.to(dev)).cpu()
        finally:
            ResNet.use_split_conv = True
        c = ora.eval()(x)
    assert a.shape == b.shape == c.shape == (3, 256, 19, 33) and (Hf, Wf) == (19, 33)
    scale = float(c.abs().max())
    print(f"trunk [{arith}]: vs MIOpen f32 {float((a - b).abs().max()) / scale:.2e}, vs oracle {float((a - c).abs().max()) / scale:.2e} of max |feature|")
    assert float((a - b).abs().max()) < tol * scale and float((a - c).abs().max()) < 1e-4 * scale
    want = torch.nn.functional.normalize(c, dim=1).flatten(2).transpose(1, 2)
    assert hw.shape == (3, 19 * 33, 256) and torch.allclose(hw.cpu(), want, atol=tol_n)
    assert not net.check_overflow()
    if arith != "bf16x3":                                  # the calibrated scales: every f16-format tensor sits at 2^7..2^8 of 2^16
        sc_ = net._scales(dev)
        assert sc_ and all(-40 < v < 40 for v in sc_.values())
    # a second call reuses the cached workspaces (their zero borders must have stayed zero); single-stream and multi-stream
    # trunks agree (MIOpen picks other, not bit-reproducible solvers for the stem / strided convolutions of a 1-image slice,
    # hence a tolerance across lane counts; bit-equality on one stream)
    lanes0 = ResNet.split_lanes
    try:
        with torch.no_grad():
            ResNet.split_lanes = 1
            s1 = net(x.to(dev)).cpu()
            s2 = net(x.to(dev)).cpu()
            ResNet.split_lanes = 2
            m1 = net(x.to(dev)).cpu()
            m2 = net(x.to(dev)).cpu()
    finally:
        ResNet.split_lanes = lanes0
    assert torch.equal(s1, s2)
    for y in (s1, m1, m2):                                # every lane count: as close to the all-MIOpen network as `a` is
        assert float((y - b).abs().max()) < tol * scale and float((a - y).abs().max()) < 3e-5 * scale
    # what forward returns is the caller's: the next call must not write into it (the trunk's workspaces are cached)
    with torch.no_grad():
        r1 = net(x.to(dev))
        keep = r1.clone()
        r2 = net(torch.randn(3, 3, 76, 132, generator=g).to(dev))
        torch.cuda.synchronize()
    assert torch.equal(r1, keep) and not torch.equal(r1, r2)


def test_sharded_tracker_hip_backend_single_rank(dev):
    """fgvc_amd.dist.track_points_sharded with the product backend (HipBackend) under a real RCCL process group of one
    rank: must reproduce the unsharded tracker (to 1e-3 px: MIOpen's stem convolution is not bit-reproducible from call
    to call, 1e-6 on the features; the 2-rank choreography is covered on CPU with gloo)."""
    import os
    import torch.distributed as dist
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import dist as fdist, engine
    g = torch.Generator().manual_seed(21)
    model = api.build_model(dict(type="VanillaTracker",
                                 backbone=dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,),
                                               pool_type="none")),
                            train_cfg=None,
                            test_cfg=api.ConfigDict(precede_frames=3, topk=10, temperature=0.07, neighbor_range=12,
                                                    with_first=True, with_first_neighbor=True)).to(dev).eval()
    rgbs = torch.randn(6, 3, 64, 96, generator=g)
    qp = torch.tensor([[0, 20.0, 12.0], [0, 70.0, 40.0], [2, 33.0, 50.0]])
    cfg = model.engine_config()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    from fgvc_amd import ops
    try:
        be = fdist.HipBackend(model)
        traj_s, order_s = fdist.track_points_sharded(be, rgbs, qp, cfg, device=dev, check=True)
        # check=True: a failure flag of the backend (here the pair kernel's bounded wait, injected) is agreed on over the group
        # (one MAX all_reduce) and raised -- the results of that video never leave the function
        ops.set_option("pair_f16_debug", 4096)
        try:
            with pytest.raises(RuntimeError, match="timed out"):
                fdist.track_points_sharded(be, rgbs, qp, cfg, device=dev, check=True)
        finally:
            ops.set_option("pair_f16_debug", 0)
        assert be.failure_flags() == (False, False)                      # read and cleared by the check
        t2, _ = fdist.track_points_sharded(be, rgbs, qp, cfg, device=dev, check=True)
        assert torch.equal(t2, traj_s)
    finally:
        if created:
            dist.destroy_process_group()
    feats, Hf, Wf = model.get_feats_hwc(rgbs.to(dev))
    traj, order = engine.track_points(feats, Hf, Wf, 64, 96, qp, cfg)
    assert torch.equal(order_s, order)
    diff = float((traj_s.cpu() - traj.cpu()).abs().max())
    assert diff < 1e-3, diff


def test_local_corr_split_path_vs_oracle(dev):
    """A7 local window on the f16 pipe (C = 256, normalised): against the oracle and against the f32-MFMA path."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(31)
    C, K, H, W, R, topk = 256, 3, 14, 19, 4, 10
    q = torch.randn(C, H, W, generator=g)
    key = torch.randn(C, K, H, W, generator=g)
    v = torch.rand(5, K, H, W, generator=g)
    qf = ops.normalize_to_hwc(q[None].to(dev))
    kf = ops.normalize_to_hwc(key.permute(1, 0, 2, 3).contiguous().to(dev))
    i1, l1, w1 = ops.local_corr_topk(qf, kf, H, W, R, topk, 0.07, normalized=True)                       # fgvc_local_corr_topk_f16x3
    with pytest.raises(ValueError):
        ops.local_corr_topk(qf, kf, H, W, R, topk, 0.07, normalized=True, split_fmt="bf16")             # retired operand format
    i0, l0, w0 = ops.local_corr_topk(qf, kf, H, W, R, topk, 0.07)
    o_out, o_idx, o_logit = O.local_corr_topk(q, key.transpose(0, 1), v.transpose(0, 1), R, topk, 0.07)
    assert torch.allclose(l1.cpu(), o_logit, atol=TOL) and torch.allclose(l1, l0, atol=1e-4)
    assert (i1.cpu().long() == o_idx).all(1).float().mean() > 0.98 and (i1 == i0).all(1).float().mean() > 0.98


@pytest.mark.parametrize("arith", ["f16f6", "f16f8", "bf16x3", "f16x3"])
@pytest.mark.parametrize("case", [(2, 256, 19, 45, True, True), (1, 128, 8, 32, False, True), (1, 256, 17, 70, True, False), (3, 64, 5, 33, True, True)])
def test_conv_split_writes_the_feature_bank_itself(dev, case, arith):
    """fgvc_conv_split_bank_f16f6p_f32 (round 4): the trunk's last convolution (3 x 3, 256 output channels) normalises its pixels and
    writes the pair kernel's f16f6 rows in its own epilogue.  Byte for byte what the two-kernel route makes -- the same convolution's
    dense f32 output through fgvc_normalize_split_f16f6p_nhwc_f32 -- in every arithmetic, with and without the identity, ragged tiles,
    rows beyond the image; and un-normalised rows (normalize = 0) likewise."""
    from fgvc_amd import ops
    N, Cin, H, W, with_res, normalize = case
    fmt = ops.ACT_FMT[arith]
    g = torch.Generator().manual_seed(7 + sum(int(v) for v in case) + fmt)
    wt = (torch.randn(256, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dev)
    bn = torch.nn.BatchNorm2d(256).eval()
    bn.weight.data = torch.rand(256, generator=g) + 0.5
    bn.bias.data = torch.randn(256, generator=g) * 0.3
    bn.running_mean = torch.randn(256, generator=g) * 0.1
    bn.running_var = torch.rand(256, generator=g) + 0.5
    bn = bn.to(dev)
    x = torch.randn(N, Cin, H, W, generator=g).abs() * (torch.rand(N, Cin, H, W, generator=g) > 0.3)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    if fmt == ops.ACT_BF16X2:
        wp, bias = ops.prepare_conv_split(wt, bn)
        sw, sx = 0, 0
        xs = ops.nchw_to_split_nhwc(x.to(dev))
    else:
        wp, bias, sw = ops.prepare_conv_split_f16(wt, bn, fmt)
        sx = ops.act_scale_log2(float(x.abs().max()))
        xs = _pack_act(x, fmt, sx, dev)
    res = torch.randn(N, H, W, 256, generator=g).to(dev) if with_res else None
    out_f = ops.alloc_nhwc(N, 256, H, W, dev)
    ops.conv_split(xs, wp, bias, H, W, True, residual=res, out_f32=out_f, in_fmt=fmt, in_scale_log2=sx + sw)
    want = ops.normalize_nhwc(out_f, normalize, split="f16f6")
    bank = torch.full((N, H * W, 2, 256), 0x5a5a, dtype=torch.int16, device=dev)
    ops.conv_split_to_bank(xs, wp, bias, H, W, True, bank, residual=res, in_fmt=fmt, in_scale_log2=sx + sw, normalize=normalize)
    same = bank == want
    assert bool(same.all()), (arith, case, int((~same).sum()), (~same).nonzero()[:4].tolist())
    # ... and what the rows decode to is the normalised output (a check of the reference route itself)
    f = torch.nn.functional.normalize(out_f.reshape(N, H * W, 256), dim=2) if normalize else out_f.reshape(N, H * W, 256)
    if normalize:                                                            # (unsplit_f16f6p: the 11-bit h part alone)
        assert float((ops.unsplit_f16f6p(bank) - f).abs().max()) < 2.5e-4
    with pytest.raises(AssertionError):                                      # 1 x 1 kernels and other widths stay on the plain entry point
        ops.conv_split_to_bank(xs, wp[:1], bias, H, W, True, bank)


def test_encoder_bank_from_the_last_convolution(dev):
    """ResNet.fuse_bank (default on): the f16f6 rows forward_hwc returns come out of the trunk's last convolution and equal the rows of
    the two-kernel route bit for bit -- in every arithmetic, eagerly and from a HIP graph, into the caller's bank slice too."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import ops
    from fgvc_amd.mmpt_api.backbones import ResNet
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none"))
    net.load_state_dict(O.seeded_resnet_state(5, (1, 2, 1, 1), "none"))
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 3, 76, 132, generator=g).to(dev)
    yes = lambda C, H, W: True
    with torch.no_grad():
        for arith in net.supported_arith():
            net.set_arith(arith)
            try:
                ResNet.fuse_bank = False
                ref, Hf, Wf = net.forward_hwc(x, True, split_if=yes, split_fmt="f16f6")
                ref = ref.clone()
            finally:
                ResNet.fuse_bank = True
            net.reset_split_cache()
            got, H2, W2 = net.forward_hwc(x, True, split_if=yes, split_fmt="f16f6")
            assert (Hf, Wf) == (H2, W2) == (19, 33) and got.shape == (3, 19 * 33, 2, 256) and got.dtype == torch.int16
            assert torch.equal(got, ref), (arith, int((got != ref).sum()))
            bank = torch.zeros(5, 19 * 33, 2, 256, dtype=torch.int16, device=dev)
            o2, _, _ = net.forward_hwc(x, True, split_if=yes, split_fmt="f16f6", out=bank[1:4])
            assert o2.data_ptr() == bank[1:4].data_ptr() and torch.equal(bank[1:4], ref) and int(bank[0].abs().max()) == 0 and int(bank[4].abs().max()) == 0
            net.check_overflow()
        # small input: the graph route (use_graph = "auto") replays the same kernels
        xs = x[:2, :, :48, :64].contiguous()
        a, _, _ = net.forward_hwc(xs, True, split_if=yes, split_fmt="f16f6")
        b, _, _ = net.forward_hwc(xs, True, split_if=yes, split_fmt="f16f6")
        c, _, _ = net.forward_hwc(xs, True, split_if=yes, split_fmt="f16f6")
        assert torch.equal(a, b) and torch.equal(b, c)
        # other formats / widths keep the two-kernel route
        f32, _, _ = net.forward_hwc(x, True)
        assert f32.dtype == torch.float32 and float((ops.unsplit_f16f6p(ref) - f32).abs().max()) < 2.5e-4       # (the 11-bit h part alone)


@pytest.mark.parametrize("arith", ["f16f6", "bf16x3", "f16x3"])        # (not f16f8: its e4m3 forms have FIXED scales that assume weights at 2^9-2^10,
@pytest.mark.parametrize("case", [(2, 256, 128, 19, 45, True), (1, 256, 64, 8, 32, False), (1, 128, 32, 17, 70, True)])   # a forced scale breaks that)
def test_conv_split_with_the_projection_folded_in(dev, case, arith):
    """fgvc_conv_split_proj_fmt_f32 (round 4): y = conv3x3(x, w) + conv1x1(x2, w2) + bias [ReLU] in ONE accumulation -- a BasicBlock's
    second convolution with the block's 1 x 1 projection shortcut folded in -- against torch in float64 of the exact f32 operands, and
    against the three-launch route (projection -> dense f32 identity -> convolution with `residual`) within the arithmetic's bound."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    N, Cin, Cin2, H, W, relu = case
    fmt = ops.ACT_FMT[arith]
    g = torch.Generator().manual_seed(11 + sum(int(v) for v in case) + fmt)
    mk = lambda c: torch.randn(N, c, H, W, generator=g).abs() ** 1.3 * (torch.rand(N, c, H, W, generator=g) > 0.4)
    x, x2 = mk(Cin), mk(Cin2) * 3.0                                           # different magnitudes: different activation scales
    wt = torch.randn(256, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    wt2 = torch.randn(256, Cin2, 1, 1, generator=g) * (1.0 / Cin2) ** 0.5

    def mkbn():
        bn = torch.nn.BatchNorm2d(256).eval()
        bn.weight.data = torch.rand(256, generator=g) + 0.5
        bn.bias.data = torch.randn(256, generator=g) * 0.1
        bn.running_mean = torch.randn(256, generator=g) * 0.1
        bn.running_var = torch.rand(256, generator=g) + 0.5
        return bn
    bn, bn2 = mkbn(), mkbn()

    def ref_of(xx, ww, b, pad):
        r = F.conv2d(xx.double(), ww.double(), padding=pad)
        sc = (b.weight / torch.sqrt(b.running_var + b.eps)).double().view(1, -1, 1, 1)
        return (r - b.running_mean.double().view(1, -1, 1, 1)) * sc + b.bias.double().view(1, -1, 1, 1)
    ref = (ref_of(x, wt, bn, 1) + ref_of(x2, wt2, bn2, 0)).detach()
    if relu:
        ref = ref.clamp_min(0)
    scale = float(ref.abs().max())
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    bn, bn2 = bn.to(dev), bn2.to(dev)
    if fmt == ops.ACT_BF16X2:
        wp, bias = ops.prepare_conv_split(wt.to(dev), bn)
        wp2, bias2 = ops.prepare_conv_split(wt2.to(dev), bn2)
        sx = sx2 = sw = sw2 = 0
        xs, xs2 = ops.nchw_to_split_nhwc(x.to(dev)), ops.nchw_to_split_nhwc(x2.to(dev))
    else:
        sx, sx2 = ops.act_scale_log2(float(x.abs().max())), ops.act_scale_log2(float(x2.abs().max()))
        wp, bias, sw = ops.prepare_conv_split_f16(wt.to(dev), bn, fmt)
        wp2, bias2, sw2 = ops.prepare_conv_split_f16(wt2.to(dev), bn2, fmt, force_exp=sx + sw - sx2)       # s_x2 s_w2 = s_x s_w
        assert sw2 == sx + sw - sx2
        xs, xs2 = _pack_act(x, fmt, sx, dev), _pack_act(x2, fmt, sx2, dev)
    so = ops.act_scale_log2(scale) if fmt != ops.ACT_BF16X2 else 0
    out_s, out_f = ops.alloc_split_nhwc(N, 256, H, W, dev), ops.alloc_nhwc(N, 256, H, W, dev)
    ops.conv_split(xs, wp, bias + bias2, H, W, relu, out_split=out_s, out_f32=out_f, in_fmt=fmt, in_scale_log2=sx + sw, out_fmt=fmt,
                   out_scale_log2=so, overflow=ovf, x2_split=xs2, w2=wp2)
    got = _nhwc_to_nchw(out_f.cpu()).double()
    tol = {"f16x3": 5e-6, "bf16x3": 2e-5}.get(arith, 3e-5)
    err = float((got - ref).abs().max()) / scale
    assert err < tol, (arith, err)
    assert int(ovf.item()) == 0
    assert float((_padded_to_nchw(ops.unsplit_act(out_s.cpu(), fmt, so), H, W).double() - got).abs().max()) < 4e-5 * scale
    # the three-launch route: the projection's own launch writes the identity in f32, the convolution adds it
    idt = ops.alloc_nhwc(N, 256, H, W, dev)
    if fmt == ops.ACT_BF16X2:
        ops.conv_split(xs2, wp2, bias2, H, W, False, out_f32=idt)
    else:
        wp2n, bias2n, sw2n = ops.prepare_conv_split_f16(wt2.to(dev), bn2, fmt)
        ops.conv_split(xs2, wp2n, bias2n, H, W, False, out_f32=idt, in_fmt=fmt, in_scale_log2=sx2 + sw2n)
    out3 = ops.alloc_nhwc(N, 256, H, W, dev)
    ops.conv_split(xs, wp, bias, H, W, relu, residual=idt, out_f32=out3, in_fmt=fmt, in_scale_log2=sx + sw)
    assert float((out3 - out_f).abs().max()) < 2 * tol * scale
    with pytest.raises(AssertionError):
        ops.conv_split(xs, wp[:1], bias, H, W, relu, out_f32=out3, x2_split=xs2, w2=wp2)       # 1 x 1 main convolution: not this entry point


def test_encoder_projection_rides_in_the_second_convolution(dev):
    """ResNet.fold_projection (default on): layer 3's 1 x 1 stride-1 projection shortcut is folded into the block's second convolution
    (fgvc_conv_split_proj_fmt_f32) in the f16f6 / bf16x3 / f16x3 arithmetics -- not in f16f8 -- and the trunk stays within the
    arithmetic's bound of the route with the projection's own launch and of the MIOpen f32 network."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd.mmpt_api.backbones import ResNet
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none"))
    net.load_state_dict(O.seeded_resnet_state(6, (1, 2, 1, 1), "none"))
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(78)
    x = torch.randn(3, 3, 76, 132, generator=g).to(dev)
    with torch.no_grad():
        try:
            ResNet.use_split_conv = False
            ref = net(x).clone()
        finally:
            ResNet.use_split_conv = True
        top = float(ref.abs().max())
        for arith in net.supported_arith():
            net.set_arith(arith)
            try:
                ResNet.fold_projection = False
                net.reset_split_cache()
                sep = net(x).clone()
            finally:
                ResNet.fold_projection = True
            net.reset_split_cache()
            got = net(x).clone()
            net.check_overflow()
            folded = [k for k, v in net._split_cache.items() if isinstance(k, tuple) and k and k[0] == "wproj" and k[4] == arith and v is not None]
            assert (len(folded) == 1) == (arith != "f16f8"), (arith, folded)
            tol = 5e-5 if arith in ("f16f8", "f16f6") else 2e-5
            assert float((got - ref).abs().max()) <= tol * top + 1e-6, (arith, float((got - ref).abs().max()), top)
            assert float((got - sep).abs().max()) <= tol * top + 1e-6
            assert (arith == "f16f8") == bool(torch.equal(got, sep))          # (the fold did change the arithmetic where it applies)


@pytest.mark.gpu
def test_encoder_layer1_whole_batch_equals_lanes(dev):
    """ResNet.layer1_whole_batch (round 5): the stem and layer 1 once over the whole batch on the caller's stream, the stream lanes forking
    behind them (layer 1's register-resident kernel cannot share a CU between lanes) -- the same bank bit for bit as with every stage in
    lanes, in every arithmetic, for an odd batch (lanes of 2 and 3 frames)."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd.mmpt_api.backbones import ResNet
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none"))
    net.load_state_dict(O.seeded_resnet_state(9, (1, 2, 1, 1), "none"))
    net = net.to(dev).eval()
    x = torch.randn(5, 3, 72, 104, generator=torch.Generator().manual_seed(4)).to(dev)
    yes = lambda C, H, W: True
    assert ResNet.layer1_whole_batch
    with torch.no_grad():
        for arith in net.supported_arith():
            net.set_arith(arith)
            fmt = "f16f6x" if arith in ("f16f6", "f16f8") else "f16"
            a = net.forward_hwc(x, True, split_if=yes, split_fmt=fmt)[0].clone()
            try:
                ResNet.layer1_whole_batch = False
                b = net.forward_hwc(x, True, split_if=yes, split_fmt=fmt)[0].clone()
            finally:
                ResNet.layer1_whole_batch = True
            assert torch.equal(a, b), arith
            f32 = net.forward_hwc(x, True)[0]
            assert f32.dtype == torch.float32 and bool(torch.isfinite(f32).all())
    assert not net.check_overflow()
