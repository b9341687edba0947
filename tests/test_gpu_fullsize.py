"""Full-size (BASELINE.json configs[1]: 8-frame 480p clip -> 120x214x256 features) checks of the HIP path through
size-independent properties, where the CPU oracle would take minutes per frame:

  * top-k lists: every index inside the disc, (score desc, index asc) order, scores equal to re-computed dot
    products, and -- on a random sample of queries -- the exact top-k of the full masked row;
  * cross-kernel: unmasked pair top-k == top-k of the columns of the dense volume kernel's output;
  * dense volume: sum of all entries == <sum k, sum q>/tau (linearity), sampled entries vs dot products,
    bf16x3 within the 1e-3 score bar of f32;
  * merge: weights sum to one, merged list = best k of the slot lists;
  * propagation: linear in the labels, constant labels are reproduced (weights sum to 1);
  * read-out: a Gaussian bump is read back at its centre;
  * encoder kernels (stem, 64-channel, stride-2, 256-channel convolutions on the clip's tensors): exact homogeneity
    (2x in -> 2x out, bit for bit), batch independence, sampled outputs against float64 receptive-field sums.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
H, W, C, T_CLIP, K = 120, 214, 256, 8, 10
HW = H * W
TAU = 0.07


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fgvc_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def clip(dev):
    """Normalised channels-last features of an 8-frame clip with spatial structure (smooth field + noise), so
    neighbouring pixels correlate like real feature maps do."""
    from fgvc_amd import ops
    g = torch.Generator(device="cpu").manual_seed(2024)
    base = torch.randn(1, C, H // 8 + 1, W // 8 + 1, generator=g)
    smooth = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    frames = smooth + 0.6 * torch.randn(T_CLIP, C, H, W, generator=g)
    return ops.normalize_to_hwc(frames.to(dev))


@pytest.fixture(scope="module")
def affinity(dev, clip):
    from fgvc_amd import engine
    cfg = engine.TrackerConfig()
    plan = engine.plan_clip(T_CLIP, [0], cfg)
    return cfg, plan, engine.run_affinity(clip, H, W, plan, cfg)


def test_normalised_features(clip):
    assert clip.shape == (T_CLIP, HW, C)
    assert torch.allclose(clip.norm(dim=2), torch.ones_like(clip[..., 0]), atol=1e-5)


@pytest.mark.parametrize("precision", ["split", "f32"])
def test_pair_lists_properties(dev, clip, precision):
    """Both pair kernels: the f16-pipe kernel on split features (the engine's default at C = 256) and the f32-MFMA one."""
    from fgvc_amd import engine, ops
    cfg = engine.TrackerConfig()
    plan = engine.plan_clip(T_CLIP, [0], cfg)
    assert len(plan.pairs) == 27                         # BASELINE cfg2: 27 unique (query, key) frame pairs
    pairs = ops.make_pairs(plan.pairs, dev)
    idx, score = ops.pair_topk_auto(clip, clip, pairs, H, W, H, W, cfg.mask, K, normalized=True, precision=precision)
    assert idx.shape == (27, HW, K) and int(idx.min()) >= 0 and int(idx.max()) < HW      # disc always holds >= k pixels
    # (a) inside the disc
    qy = (torch.arange(HW, device=dev) // W).view(1, HW, 1)
    qx = (torch.arange(HW, device=dev) % W).view(1, HW, 1)
    d2 = (idx // W - qy) ** 2 + (idx % W - qx) ** 2
    assert int(d2.max()) <= cfg.mask.r2max
    # (b) canonical order
    ds = score[..., 1:] - score[..., :-1]
    assert float(ds.max()) <= 0.0
    tie = ds == 0
    assert bool((idx[..., 1:][tie] > idx[..., :-1][tie]).all())
    # (c) scores are the dot products of the rows they point at
    for p in (0, 13, 26):
        qf, kf = int(pairs[p, 0]), int(pairs[p, 1])
        dots = torch.einsum("qc,qkc->qk", clip[qf], clip[kf][idx[p].long()])
        assert torch.allclose(dots, score[p], atol=4e-6)
    # (d) exact top-k of the full masked row on a sample of queries (f64 scores; ranks separated by > 1e-6)
    g = torch.Generator().manual_seed(5)
    sample = torch.cat([torch.tensor([0, W - 1, HW - W, HW - 1, 60 * W + 107]), torch.randint(0, HW, (251,), generator=g)])
    sample = sample.to(dev)
    ky = (torch.arange(HW, device=dev) // W).view(-1, 1)
    kx = (torch.arange(HW, device=dev) % W).view(-1, 1)
    inside = ((ky - (sample // W).view(1, -1)) ** 2 + (kx - (sample % W).view(1, -1)) ** 2) <= cfg.mask.r2max
    for p in (1, 20):
        qf, kf = int(pairs[p, 0]), int(pairs[p, 1])
        full = (clip[kf].double() @ clip[qf][sample].double().t()).masked_fill(~inside, float("-inf"))   # (HW, n)
        tv, ti = full.topk(K + 1, dim=0)
        clear = ((tv[:-1] - tv[1:]).min(0).values > 1e-6)
        assert int(clear.sum()) > 200
        got = idx[p][sample].t().long()
        assert torch.equal(got[:, clear], ti[:K][:, clear])
        assert torch.allclose(score[p][sample].t().double(), tv[:K], atol=1e-5)


def test_pair_kernels_agree_full_size(dev, clip):
    """The 16-bit pair kernel against the exact-f32 one at full size: scores within f32 summation noise, index rows equal except
    where two scores are that close."""
    from fgvc_amd import engine, ops
    cfg = engine.TrackerConfig()
    pairs = ops.make_pairs([(3, 0, True), (3, 2, True)], dev)
    i32, s32 = ops.pair_topk(clip, clip, pairs, H, W, H, W, cfg.mask, K)
    h16 = ops.split_f16x2(clip)
    i16, s16 = ops.pair_topk_split(h16, h16, pairs, H, W, H, W, cfg.mask, K, all_masked=True)
    assert not ops.pair_f16x3_timed_out()
    assert float((s16 - s32).abs().max()) < 2.5e-6
    differ = ~(i16 == i32).all(-1)
    assert float(differ.float().mean()) < 1e-2
    if differ.any():
        assert float((torch.sort(s16[differ], dim=-1).values - torch.sort(s32[differ], dim=-1).values).abs().max()) < 2.5e-6


@pytest.mark.parametrize("precision", ["split", "f32"])
def test_unmasked_pair_topk_equals_topk_of_dense_volume(dev, clip, precision):
    from fgvc_amd import ops
    pairs = ops.make_pairs([(1, 0, False)], dev)
    idx, score = ops.pair_topk_auto(clip, clip, pairs, H, W, H, W, ops.MaskSpec.none(), K, normalized=True,
                                    precision=precision)
    vol = ops.corr_volume(clip[1], clip[0], 1.0, "f32")                  # (HWk, HWq)
    tv, ti = vol.topk(K + 1, dim=0)
    clear = (tv[:-1] - tv[1:]).min(0).values > 2e-6
    assert float(clear.float().mean()) > 0.9
    assert torch.equal(idx[0].t().long()[:, clear], ti[:K][:, clear])
    assert torch.allclose(score[0].t(), tv[:K], atol=4e-6)


def test_dense_volume_linearity_and_samples(dev, clip):
    from fgvc_amd import ops
    q, k = clip[1], clip[0]
    hl = ops.split_bf16(clip[:2])
    vols = {"f32": ops.corr_volume(q, k, TAU, "f32")}
    expect_sum = float((k.double().sum(0) * q.double().sum(0)).sum() / TAU)
    g = torch.Generator().manual_seed(11)
    kk = torch.randint(0, HW, (4096,), generator=g).to(dev)
    qq = torch.randint(0, HW, (4096,), generator=g).to(dev)
    ref = (k[kk].double() * q[qq].double()).sum(1) / TAU
    sp = {"f16f8": ops.split_f16f8(clip[:2]), "f16f6": ops.split_f16f6(clip[:2])}
    for prec, tol in (("f32", 2e-5), ("bf16x3", 1e-3), ("f16f8", 1e-3), ("f16f6", 1e-3)):
        vol = vols.get(prec)
        if vol is None:
            vol = ops.corr_volume(sp[prec][1], sp[prec][0], TAU, prec) if prec in sp else ops.corr_volume(hl[1], hl[0], TAU, prec)
        assert vol.shape == (HW, HW)
        assert abs(float(vol.double().sum()) - expect_sum) <= 1e-6 * HW * HW * 0.2 + 1e-3 * abs(expect_sum)
        assert float((vol[kk, qq].double() - ref).abs().max()) < tol
        if prec != "f32":
            assert float((vol - vols["f32"]).abs().max()) < 1e-3           # the north_star score bar, every entry
        # first/last rows and columns (ragged tile edges)
        for j in (0, HW - 1):
            assert torch.allclose(vol[j].double(), (q.double() @ k[j].double()) / TAU, atol=tol)
            assert torch.allclose(vol[:, j].double(), (k.double() @ q[j].double()) / TAU, atol=tol)
        if prec != "f32":
            del vol
    del vols


def test_merged_lists(affinity):
    cfg, plan, tk = affinity
    assert tk.idx.shape == (T_CLIP - 1, HW, K)
    assert torch.allclose(tk.weight.sum(-1), torch.ones_like(tk.weight[..., 0]), atol=1e-5)
    assert float((tk.logit[..., 1:] - tk.logit[..., :-1]).max()) <= 0.0
    n_slots = tk.slot_frame.shape[1]
    assert int(tk.idx.min()) >= 0 and int(tk.idx.max()) < n_slots * HW
    # softmax of the logits
    assert torch.allclose(torch.softmax(tk.logit, -1), tk.weight, atol=1e-5)


def test_merge_is_best_k_of_slot_lists(dev, clip, affinity):
    """Output frame 7 has six key slots [0, 2..6]: its list must be the best k of the union of its pair lists."""
    from fgvc_amd import ops
    cfg, plan, tk = affinity
    row = plan.out_rows[(0, 7)]
    slots = [int(s) for s in tk.slot_frame[row].tolist() if s >= 0]
    pairs = ops.make_pairs([(7, s, True) for s in slots], dev)
    pidx, pscore = ops.pair_topk_auto(clip, clip, pairs, H, W, H, W, cfg.mask, K, normalized=True)   # as the engine does
    gid = pidx.long() + (torch.arange(len(slots), device=dev) * HW).view(-1, 1, 1)
    allv = pscore.permute(1, 0, 2).reshape(HW, -1)
    alli = gid.permute(1, 0, 2).reshape(HW, -1)
    # canonical order: score desc then global index asc
    order = torch.argsort(alli, dim=1, stable=True)
    allv, alli = allv.gather(1, order), alli.gather(1, order)
    order = torch.argsort(allv, dim=1, descending=True, stable=True)[:, :K]
    assert torch.equal(alli.gather(1, order), tk.idx[row].long())
    assert torch.allclose(allv.gather(1, order) / cfg.temperature, tk.logit[row], atol=1e-5)


def test_propagation_linearity_and_constants(dev, affinity):
    from fgvc_amd import ops
    cfg, plan, tk = affinity
    row = plan.out_rows[(0, 6)]
    n_slots = int((tk.slot_frame[row] >= 0).sum())
    g = torch.Generator().manual_seed(3)
    P = 16
    A = torch.rand(T_CLIP, HW, P, generator=g).to(dev)
    B = torch.rand(T_CLIP, HW, P, generator=g).to(dev)
    f = lambda L: ops.propagate_topk(L, tk.slot_frame[row], tk.idx[row], tk.weight[row], H, W, H, W)
    assert torch.allclose(f(2.0 * A - 3.0 * B), 2.0 * f(A) - 3.0 * f(B), atol=1e-5)
    assert torch.allclose(f(torch.full_like(A, 0.25)), torch.full((HW, P), 0.25, device=dev), atol=1e-6)
    # against the definition
    sf = tk.slot_frame[row].long()
    slot, pix = tk.idx[row].long() // HW, tk.idx[row].long() % HW
    want = (A[sf[slot], pix] * tk.weight[row].unsqueeze(-1)).sum(1)
    assert torch.allclose(f(A), want, atol=1e-5)
    assert n_slots == 6


def test_readout_recovers_gaussian_centres(dev):
    from fgvc_amd import ops
    h, w = 480, 854
    pts = torch.tensor([[100.0, 200.0], [4.0, 4.0], [848.0, 472.0], [427.0, 240.0]], device=dev)   # (x, y), multiples of 4
    labels = torch.stack([ops.gaussian_labels(pts, H, W, 4, 6.0)])                                   # (1, HW, P)
    coords = ops.softargmax_top5(labels, H, W, h, w)
    assert coords.shape == (1, 4, 2)
    # bilinear upsampling of a sampled Gaussian peaks within a pixel or two of the centre
    assert float((coords[0].float() - pts).abs().max()) < 2.5


def test_track_points_full_size(dev, clip):
    """End to end at cfg2 size: stationary features (every frame identical) must keep every point where it started."""
    from fgvc_amd import engine
    cfg = engine.TrackerConfig()
    still = clip[:1].expand(T_CLIP, HW, C).contiguous()
    g = torch.Generator().manual_seed(9)
    xy = torch.stack([torch.randint(8, 846, (12,), generator=g), torch.randint(8, 472, (12,), generator=g)], 1).float()
    qp = torch.cat([torch.zeros(12, 1), xy], 1)
    traj, order = engine.track_points(still, H, W, 480, 854, qp, cfg)
    assert traj.shape == (T_CLIP, 12, 2)
    drift = (traj.float() - xy[order].to(dev).unsqueeze(0)).abs().max()
    assert float(drift) < 3.0, float(drift)


# ---------------------------------------------------------------------------------------------------------------
# encoder kernels at the 480p clip's sizes: properties that need no oracle
# ---------------------------------------------------------------------------------------------------------------
def _bn_identity_bias0(C, dev, g):
    """eval-mode BatchNorm whose folded bias is exactly zero (so that the layer is homogeneous of degree one)."""
    bn = torch.nn.BatchNorm2d(C).eval()
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data.zero_()
    bn.running_mean.zero_()
    bn.running_var = torch.rand(C, generator=g) + 0.5
    return bn.to(dev)


def _sample_check(out_nhwc, x_nchw, wt, bn, stride, pad, relu, n=48, seed=0):
    """`n` random output positions against a float64 convolution of the receptive field alone."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    N, Ho, Wo, Co = out_nhwc.shape
    KS = wt.shape[-1]
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().double().cpu()
    bias = (bn.bias - bn.running_mean * (bn.weight / torch.sqrt(bn.running_var + bn.eps))).detach().double().cpu()
    xp = F.pad(x_nchw.double().cpu(), (pad, pad, pad, pad))
    worst = 0.0
    for _ in range(n):
        i, y, x = int(torch.randint(N, (1,), generator=g)), int(torch.randint(Ho, (1,), generator=g)), int(torch.randint(Wo, (1,), generator=g))
        if _ % 4 == 0:
            y, x = (0, Ho - 1)[_ % 8 == 0], (0, Wo - 1)[_ % 3 == 0]               # corners and edges
        patch = xp[i, :, y * stride: y * stride + KS, x * stride: x * stride + KS]
        ref = (wt.double().cpu() * patch[None]).sum((1, 2, 3)) * sc + bias
        if relu:
            ref = ref.clamp_min(0)
        got = out_nhwc[i, y, x].double().cpu()
        worst = max(worst, float((got - ref).abs().max()) / max(1.0, float(ref.abs().max())))
    return worst


def test_encoder_kernels_full_size_properties(dev):
    """fgvc_stem7_split_f32, fgvc_conv64_split_f32, fgvc_conv_s2_split_f32 and fgvc_conv_split_f32 on the 8 x 480 x 854 clip's
    tensors: (i) exact homogeneity -- doubling the input doubles every output bit for bit (power-of-two scaling commutes with the
    bf16 split, the products and the f32 accumulation; bias folded to zero); (ii) batch independence -- an image alone gives the
    bits it gives inside the batch (persistent tile loops, lanes of tiles); (iii) sampled outputs, borders included, against a
    float64 convolution of their receptive fields; (iv) the register-resident 64-channel kernel against the generic one."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(77)
    N, h, w = 8, 480, 854
    frames = torch.randn(N, 3, h, w, generator=g).to(dev)
    # stem
    w0 = (torch.randn(64, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5).to(dev)
    bn0 = _bn_identity_bias0(64, dev, g)
    sw, sb = ops.prepare_stem7(w0, bn0)
    H1, W1 = 240, 427

    def stem(x):
        s, f = ops.alloc_split_nhwc(x.shape[0], 64, H1, W1, dev), ops.alloc_nhwc(x.shape[0], 64, H1, W1, dev)
        ops.stem7_split(x, sw, sb, True, out_split=s, out_f32=f)
        return s, f

    s1, f1 = stem(frames)
    s1d, f1d = stem(frames * 2)
    assert torch.equal(f1d, f1 * 2)
    assert torch.equal(stem(frames[5:6])[1][0], f1[5])
    assert _sample_check(f1, frames, w0, bn0, 2, 3, True) < 2e-5
    # layer 1: 64 -> 64 3x3 with residual, register-resident weights vs the generic kernel
    w1 = (torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5).to(dev)
    bn1 = _bn_identity_bias0(64, dev, g)
    c6, b6 = ops.prepare_conv64(w1, bn1)
    cg, bg = ops.prepare_conv_split(w1, bn1)

    def layer1(s_in, res, fn, wb):
        s, f = ops.alloc_split_nhwc(s_in.shape[0], 64, H1, W1, dev), ops.alloc_nhwc(s_in.shape[0], 64, H1, W1, dev)
        fn(s_in, wb[0], wb[1], H1, W1, True, residual=res, out_split=s, out_f32=f)
        return s, f

    s2, f2 = layer1(s1, f1, ops.conv64_split, (c6, b6))
    _, f2d = layer1(s1d, f1d, ops.conv64_split, (c6, b6))
    assert torch.equal(f2d, f2 * 2)
    assert torch.equal(layer1(s1[2:3].contiguous(), f1[2:3].contiguous(), ops.conv64_split, (c6, b6))[1][0], f2[2])
    _, f2g = layer1(s1, f1, ops.conv_split, (cg, bg))
    assert float((f2g - f2).abs().max()) < 1e-5 * float(f2.abs().max())
    x1_nchw = f1.permute(0, 3, 1, 2)
    no_res = ops.alloc_nhwc(N, 64, H1, W1, dev)
    ops.conv64_split(s1, c6, b6, H1, W1, False, out_f32=no_res)
    assert _sample_check(no_res, x1_nchw, w1, bn1, 1, 1, False, seed=1) < 2e-5
    # layer 2 entry: stride-2 3x3 (64 -> 128) and its 1x1 projection
    H2, W2 = 120, 214
    for KS in (3, 1):
        w2 = (torch.randn(128, 64, KS, KS, generator=g) * (2.0 / (64 * KS * KS)) ** 0.5).to(dev)
        bn2 = _bn_identity_bias0(128, dev, g)
        c2, b2 = ops.prepare_conv_s2(w2, bn2)

        def s2conv(s_in):
            f = ops.alloc_nhwc(s_in.shape[0], 128, H2, W2, dev)
            ops.conv_s2_split(s_in, c2, b2, H1, W1, KS == 3, out_f32=f)
            return f

        f3 = s2conv(s2)
        assert torch.equal(s2conv(layer1(s1d, f1d, ops.conv64_split, (c6, b6))[0]), f3 * 2)
        assert torch.equal(s2conv(s2[7:8].contiguous())[0], f3[7])
        assert _sample_check(f3, f2.permute(0, 3, 1, 2), w2, bn2, 2, KS // 2, KS == 3, seed=2 + KS) < 2e-5
    # layer 3: 256 -> 256 3x3 on the 120 x 214 grid
    x3 = torch.randn(N, 256, H2, W2, generator=g).to(dev)
    w3 = (torch.randn(256, 256, 3, 3, generator=g) * (2.0 / 2304) ** 0.5).to(dev)
    bn3 = _bn_identity_bias0(256, dev, g)
    c3, b3 = ops.prepare_conv_split(w3, bn3)

    def layer3(x):
        f = ops.alloc_nhwc(x.shape[0], 256, H2, W2, dev)
        ops.conv_split(ops.nchw_to_split_nhwc(x), c3, b3, H2, W2, True, out_f32=f)
        return f

    f4 = layer3(x3)
    assert torch.equal(layer3(x3 * 2), f4 * 2)
    assert torch.equal(layer3(x3[3:4].contiguous())[0], f4[3])
    assert _sample_check(f4, x3, w3, bn3, 1, 1, True, seed=9) < 2e-5
