"""GPU parity of fgvc_pair_topk_f16f6 (round 4): the windowed correlation + top-k in the f16 + FP6 arithmetic (csrc/pair_topk_v7.hpp).

Bars: the operand format is byte work -> BIT-EXACT against the checker's restatement (oracle.f16f6p_encode); scores within 1e-3 logit
of float64 (north_star; observed <= 7e-5) and within 1e-6 cosine of the checker's model of the kernel's own arithmetic
(oracle.f16f6_cosines: what the matrix instructions are fed, in float64); indices equal to the float64 top-k on every query whose
ranks are further apart than the arithmetic's error (3e-4 logit), legitimate-within-tolerance otherwise; lists independent of how
the pairs are grouped into runs, bit-identical from launch to launch, poison after an injected protocol fault.
Reference: local_attention.py:318-371 (masked_attention_efficient), vanilla_tracker.py:345-394."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import fgvc_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TAU = 0.07
GAP = 3e-4            # logit gap above which the f16 + FP6 scores must rank like float64 (the arithmetic's error is ~6e-5)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fgvc_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rows(feats):
    return feats.contiguous().view(torch.uint8).reshape(-1, 1024).cpu().numpy()


def test_split_f16f6p_bytes_equal_the_checker(dev):
    """fgvc_split_f16f6p against oracle.f16f6p_encode, byte for byte: Gaussian, ReLU-sparse, one-hot, all-zero and tiny rows (block
    scales from 2^-40 to 2^6, FP6 ties, saturation at 7.5), and the fused normalise + split pass against the two-pass form."""
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(700, 256, generator=g)
    x[100:200] = torch.relu(x[100:200]) * (torch.rand(100, 256, generator=g) < 0.2)
    x[200:232] = torch.eye(256)[::8]                                    # one-hot rows: h = 256 exactly, residual 0
    x[232] = 0
    x[233:300] *= 1e-3                                                  # tiny blocks inside a normalised row
    x[233:300, 0] = 1.0
    x[300:400] = torch.randn(100, 256, generator=g).pow(3)               # heavy tails
    f = torch.nn.functional.normalize(x, dim=1)
    f[232] = 0
    got = _rows(ops.split_f16f6p(f.to(dev)))
    want = O.f16f6p_encode(f.numpy())
    assert got.shape == want.shape == (700, 1024)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"{len(bad)} bytes differ, first at row {bad[0][0]} byte {bad[0][1]}: {got[tuple(bad[0])]} != {want[tuple(bad[0])]}"
    # the h part is the 11-bit form of 256 x; the FP6 forms carry 4 bits of h and of the residual under their block scales
    h, h6, l6 = O.f16f6p_decode(got)
    xs = f.double().numpy() * 256.0
    assert np.abs(h - xs).max() <= 2.0 ** -4 and np.abs(h6 - h).max() <= np.abs(h).max() / 16 + 1e-9
    # ... and the one pass from the trunk's NHWC output writes the same bytes
    y = torch.randn(2, 9, 13, 256, generator=g).to(dev) * torch.rand(2, 9, 13, 1, generator=g).to(dev)
    assert torch.equal(ops.normalize_nhwc(y, True, split="f16f6"), ops.split_f16f6p(ops.normalize_nhwc(y, True)))
    assert torch.equal(ops.normalize_nhwc(y, False, split="f16f6"), ops.split_f16f6p(ops.normalize_nhwc(y, False)))


def _pairs_case(dev, H, W, Tn, seed, kind="gauss"):
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(Tn, 256, H, W, generator=g)
    if kind == "relu":
        x = torch.relu(x)
    elif kind == "smooth":
        x = torch.nn.functional.avg_pool2d(torch.randn(Tn, 256, H + 6, W + 6, generator=g), 7, 1) + 0.05 * x
    return ops.normalize_to_hwc(x.to(dev))


@pytest.mark.parametrize("case", [(33, 70, 4, 30, "circle", 10, "gauss"), (37, 53, 3, 30, "circle", 10, "relu"),
                                  (24, 40, 4, 30, "circle", 10, "smooth"), (17, 23, 3, 9, "square", 5, "gauss"),
                                  (8, 8, 3, 30, "circle", 10, "gauss"), (5, 3, 2, 4, "circle", 5, "relu"),
                                  (16, 24, 3, 12, "circle", 10, "relu")])
def test_pair_f16f6_vs_float64_and_model(dev, case):
    """ragged grids (edge tiles, frames smaller than a tile), disc and rectangular windows, k = 10 and 5: scores against float64
    (1e-3 logit; observed <= 7e-5) and against the model of the kernel's own arithmetic (2e-6 cosine: f32 accumulation + the 2^-20
    quantisation of a key), indices exact where float64 ranks are GAP apart, every index legitimately in the tolerance top-k."""
    from fgvc_amd import engine, ops
    H, W, Tn, nr, mm, k, kind = case
    feats = _pairs_case(dev, H, W, Tn, seed=H * 100 + W, kind=kind)
    mask = ops.MaskSpec.from_neighbor_range(nr, mm)
    assert ops.pair_f16f6_ok(256, H, W, k, True, None, mask, True)
    rows = [(q, kk, True) for q in range(1, Tn) for kk in range(q)]
    pairs = ops.make_pairs(rows, dev)
    sp = ops.split_f16f6p(feats)
    idx, score = ops.pair_topk_split(sp, sp, pairs, H, W, H, W, mask, k, all_masked=True, fmt="f16f6")
    torch.cuda.synchronize()
    assert not ops.pair_f16x3_timed_out()
    rb = _rows(sp).reshape(Tn, H * W, 1024)
    worst, worst_m = 0.0, 0.0
    for pi, (q, kk, _) in enumerate(rows):
        fq, fk = feats[q].double().cpu(), feats[kk].double().cpu()
        dense = (fk @ fq.T) / TAU
        m = O.mask_slab(H, W, H, W, 1, torch.arange(H * W), nr, mm)
        dm = dense.masked_fill(~m, float("-inf"))
        st = O.check_topk(dm, idx[pi].cpu().long(), score[pi].cpu() / TAU, k, tol=1e-3, gap=GAP)
        assert st["exact"] >= st["clear"]
        worst = max(worst, st["max_score_err"])
        if pi < 2:                                                     # the arithmetic model (python loops over blocks: two pairs per case)
            model = T(O.f16f6_cosines(rb[q], rb[kk]))
            got = model.t().gather(1, idx[pi].cpu().long().clamp_min(0))
            fin = idx[pi].cpu() >= 0
            worst_m = max(worst_m, float((got - score[pi].cpu().double())[fin].abs().max()))
    assert worst < 2e-4 and worst_m < 2e-6, (worst, worst_m)
    # empty entries (a window smaller than k) are -1 / -inf, at the tail of a list
    emp = idx < 0
    assert bool((score[emp] == float("-inf")).all()) and bool((emp[..., 1:] >= emp[..., :-1]).all())


def test_pair_f16f6_vs_reference_golden(dev, golden):
    """The reference's own fixture with 256 channels (masked_attention_efficient, mae_s16x24_c256): merged top-k logits against what the
    reference's topk returned, propagated labels against its output tensor -- through fgvc_pair_topk_f16f6."""
    from fgvc_amd import ops
    g = golden("mae_s16x24_c256")
    q, key, v = T(g["query"])[0], T(g["key"])[0], T(g["value"])[0]
    nr, topk, nml, mode = int(g["nr"]), int(g["topk"]), int(g["non_mask_len"]), str(g["mode"])
    C, H, W = q.shape
    Tn = key.shape[1]
    assert C == 256 and nml == 0
    frames = torch.cat([q.unsqueeze(0), key.permute(1, 0, 2, 3)], 0).to(dev)
    sp = ops.split_f16f6p(ops.normalize_to_hwc(frames))
    mask = ops.MaskSpec.from_neighbor_range(nr, "circle")
    pairs = ops.make_pairs([(0, 1 + t, True) for t in range(Tn)], dev)
    pidx, pscore = ops.pair_topk_split(sp, sp, pairs, H, W, H, W, mask, topk, all_masked=True, fmt="f16f6")
    slot_pair = torch.arange(Tn, dtype=torch.int32, device=dev).view(1, Tn)
    idx, logit, weight = ops.merge_topk(pidx, pscore, slot_pair, H * W, topk, TAU, mode)
    rv = T(g["ref_topk_val"])
    assert float((logit[0].cpu() - rv.sort(1, descending=True)[0]).abs().max()) < 2e-4
    labels = v.permute(1, 2, 3, 0).reshape(Tn, H * W, -1).contiguous().to(dev)
    out = ops.propagate_topk(labels, torch.arange(Tn, dtype=torch.int32, device=dev), idx[0], weight[0], H, W, H, W)
    ref_out = T(g["out"])[0].flatten(1).t()
    assert float((out.cpu() - ref_out).abs().max()) < 1e-3
    vol = O.corr_volume(q.double(), key.double(), TAU)
    m = O.mask_slab(H, W, H, W, Tn, torch.arange(H * W), nr, "circle", 0)
    st = O.check_topk(vol.masked_fill(~m, float("-inf")), idx[0].cpu().long(), logit[0].cpu(), topk, tol=1e-3, gap=GAP)
    assert st["exact"] >= st["clear"] > 0.8 * st["queries"]


def test_pair_f16f6_runs_subsets_and_the_f16x3_kernel(dev):
    """Lists do not depend on how pairs are grouped (runs of 1-4 pairs of a query frame against one workgroup per pair against a
    subset of the pairs): bit-identical, three times over; and they differ from fgvc_pair_topk_f16x3's only where that kernel's own
    scores are within the FP6 error of each other."""
    from fgvc_amd import ops
    H, W, Tn = 33, 70, 6
    f = _pairs_case(dev, H, W, Tn, seed=77)
    sp6, sp3 = ops.split_f16f6p(f), ops.split_f16x2(f)
    rows = [(1, 0, True), (2, 0, True), (2, 1, True), (3, 0, True), (3, 1, True), (3, 2, True), (5, 0, True), (5, 1, True), (5, 3, True),
            (5, 4, True), (4, 4, True)]
    pairs = ops.make_pairs(rows, dev)
    mask = ops.MaskSpec.from_neighbor_range(30)
    ia, sa = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16f6")
    for _ in range(3):
        ib, sb = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16f6", use_runs=False)
        assert torch.equal(ia, ib) and torch.equal(sa, sb)
        sel = torch.tensor([0, 2, 5, 9, 10], device=dev)
        ic, sc = ops.pair_topk_split(sp6, sp6, pairs[sel].contiguous(), H, W, H, W, mask, 10, all_masked=True, fmt="f16f6")
        assert torch.equal(ic, ia[sel]) and torch.equal(sc, sa[sel])
    i3, s3 = ops.pair_topk_split(sp3, sp3, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16")
    assert not ops.pair_f16x3_timed_out()
    assert float((sa - s3).abs().max()) < 1.2e-5                         # cosine: 1.7e-4 logit
    differ = ~(ia == i3).all(-1)
    assert float(differ.float().mean()) < 5e-3
    if differ.any():                                                     # ... the same multiset of scores up to the error, in another order
        assert float((torch.sort(sa[differ], dim=-1).values - torch.sort(s3[differ], dim=-1).values).abs().max()) < 1.2e-5
    # top-5 lists are the head of the top-10 lists
    i5, s5 = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, mask, 5, all_masked=True, fmt="f16f6")
    assert torch.equal(i5, ia[..., :5]) and torch.equal(s5, sa[..., :5])


def test_pair_f16f6_refuses_what_it_cannot_do(dev):
    """All-masked pairs under a mask of at most 64 key blocks, C = 256, k <= 10 -- anything else is an error (Python) / FGVC_ERR_UNSUPPORTED
    (C ABI), never a silent other kernel."""
    from fgvc_amd import _lib, ops
    H, W = 24, 40
    f = _pairs_case(dev, H, W, 2, seed=3)
    sp = ops.split_f16f6p(f)
    pairs = ops.make_pairs([(1, 0, True)], dev)
    assert ops.pair_blocks_reached(ops.MaskSpec.from_neighbor_range(30)) == 56
    wide = ops.MaskSpec.from_neighbor_range(48)
    assert ops.pair_blocks_reached(wide) > 64 and not ops.pair_f16f6_ok(256, H, W, 10, True, None, wide, True)
    with pytest.raises(ValueError):
        ops.pair_topk_split(sp, sp, pairs, H, W, H, W, wide, 10, all_masked=True, fmt="f16f6")
    with pytest.raises(ValueError):
        ops.pair_topk_split(sp, sp, pairs, H, W, H, W, ops.MaskSpec.from_neighbor_range(30), 10, all_masked=False, fmt="f16f6")
    lib = _lib.load()
    idx = torch.empty((1, H * W, 10), dtype=torch.int32, device=dev)
    sc = torch.empty((1, H * W, 10), dtype=torch.float32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def call(r2max, topk, all_masked, C=256):
        return lib.fgvc_pair_topk_f16f6(P(sp), P(sp), P(pairs), 1, C, H, W, H, W, r2max, _lib.NO_LIMIT, _lib.NO_LIMIT, topk, all_masked, P(idx), P(sc), None)
    assert call(224, 10, 1) == 0
    assert call(wide.r2max, 10, 1) == _lib.ERR_UNSUPPORTED
    assert call(224, 10, 0) == _lib.ERR_UNSUPPORTED
    assert call(224, 11, 1) == _lib.ERR_UNSUPPORTED
    assert call(224, 10, 1, C=128) == _lib.ERR_UNSUPPORTED
    assert call(_lib.NO_LIMIT, 10, 1) == _lib.ERR_UNSUPPORTED
    torch.cuda.synchronize()
    assert not ops.pair_f16x3_timed_out()


@pytest.mark.parametrize("shape", [(120, 214), (128, 128)])
def test_pair_f16f6_soak_and_fail_closed(dev, shape):
    """300 launches of the 27-pair plan at the cfg2 and cfg4 grid shapes through the engine (bank in the f16f6 format): bit-identical,
    no bounded wait gives up; with the workgroup flag forced every list is poison (NaN weights after the merge) and the device flag
    reports once."""
    from fgvc_amd import engine, ops
    H, W = shape
    cfg = engine.TrackerConfig(pair_split_fmt="f16f6")
    plan = engine.plan_clip(8, [0], cfg)
    g = torch.Generator(device=dev).manual_seed(H)
    base = torch.randn(1, 256, H // 8 + 1, W // 8 + 1, generator=g, device=dev)
    smooth = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    clip = ops.split_f16f6p(torch.cat([ops.normalize_to_hwc(smooth + 0.6 * torch.randn(1, 256, H, W, generator=g, device=dev)) for _ in range(8)], 0))
    ref = engine.run_pairs(clip, H, W, plan, cfg)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(300):
        pl = engine.run_pairs(clip, H, W, plan, cfg)
        bad += (pl.idx != ref.idx).sum() + (pl.score != ref.score).sum()
        if i % 100 == 99:
            assert int(bad) == 0, f"launch {i - 99}..{i}: {int(bad)} differing entries"
    assert int(bad) == 0 and not ops.pair_f16x3_timed_out()
    # size-independent properties of the full-size lists: inside the disc, descending, ties by ascending pixel
    qy = (torch.arange(H * W, device=dev) // W).view(1, -1, 1)
    qx = (torch.arange(H * W, device=dev) % W).view(1, -1, 1)
    d2 = (ref.idx // W - qy) ** 2 + (ref.idx % W - qx) ** 2
    assert int(ref.idx.min()) >= 0 and int(d2.max()) <= cfg.mask.r2max
    ds = ref.score[..., 1:] - ref.score[..., :-1]
    assert float(ds.max()) <= 0.0 and bool((ref.idx[..., 1:][ds == 0] > ref.idx[..., :-1][ds == 0]).all())
    ops.set_option("pair_f16_debug", 4096)
    try:
        pl = engine.run_pairs(clip, H, W, plan, cfg)
        tk = engine.merge_pairs(pl, cfg)
        assert bool(torch.isinf(pl.score).all()) and bool(torch.isnan(tk.weight).all())
        assert ops.pair_f16x3_timed_out() and not ops.pair_f16x3_timed_out()
    finally:
        ops.set_option("pair_f16_debug", 0)
    pl = engine.run_pairs(clip, H, W, plan, cfg)
    assert torch.equal(pl.idx, ref.idx) and not ops.pair_f16x3_timed_out()


@pytest.mark.parametrize("shape", [(16, 40, 150), (9, 17, 70), (24, 24, 66)])
def test_pair_f16f6_work_order_over_many_runs(dev, shape):
    """The kernel's work order (XCD-contiguous ranges of the runs of one length, column strips, aligned walks) on launches with MORE THAN
    64 runs -- the stretch of equal-length runs around a workgroup's own is found 64 runs at a time -- of mixed lengths, odd numbers of
    tile columns (a last strip one tile wide) and fewer workgroups than XCDs' shares: every (pair, query) list
    equals what one workgroup per pair, the first build's order (debug 512 + 1024) and fgvc_pair_topk_f16x3 give."""
    from fgvc_amd import ops
    H, W, Tn = shape
    f = _pairs_case(dev, H, W, Tn, seed=H + W + Tn)
    sp6, sp3 = ops.split_f16f6p(f), ops.split_f16x2(f)
    rows = []
    for q in range(1, Tn):                                  # runs of 1 .. 4 pairs per query frame, in no particular order of length
        for kk in range(max(0, q - 1 - (q * 7) % 4), q):
            rows.append((q, kk, True))
    pairs = ops.make_pairs(rows, dev)
    runs = ops.pair_runs(pairs)
    assert runs.shape[0] == Tn - 1 > 64 and len(set(runs[:, 1].tolist())) >= 3
    mask = ops.MaskSpec.from_neighbor_range(12)
    ia, sa = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16f6")
    ib, sb = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16f6", use_runs=False)
    assert torch.equal(ia, ib) and torch.equal(sa, sb)
    ops.set_option("pair_f16_debug", 512 + 1024)
    try:
        ic, sc = ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16f6")
    finally:
        ops.set_option("pair_f16_debug", 0)
    assert torch.equal(ia, ic) and torch.equal(sa, sc)
    i3, s3 = ops.pair_topk_split(sp3, sp3, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16")
    assert not ops.pair_f16x3_timed_out()
    fin = torch.isfinite(s3)
    assert bool((torch.isfinite(sa) == fin).all()) and float((sa[fin] - s3[fin]).abs().max()) < 1.2e-5
    assert float((~(ia == i3).all(-1)).float().mean()) < 2e-2


V8 = 4194304          # pair_f16_debug bit: pair_topk_kernel_v8 (csrc/pair_topk_v8.hpp) takes the launch instead of pair_topk_kernel_v7


@pytest.mark.parametrize("case", [(33, 70, 4, 30, "circle", 10), (37, 53, 3, 30, "circle", 5), (17, 23, 3, 9, "square", 5),
                                  (8, 8, 3, 30, "circle", 10), (5, 3, 2, 4, "circle", 5), (64, 48, 7, 30, "circle", 10),
                                  (40, 72, 3, 32, "square", 10)])
def test_pair_v8_equals_v7(dev, case):
    """The one-role kernel of round 6 (a key block staged once for eight query blocks, one kind of wave; opt-in, not the default) against
    the three-role kernel: the same instructions on the same operands in the same order -> scores BIT-IDENTICAL; indices equal except
    where a score ties across the K-th place (v8's selection keys are canonical: the lower pixel index wins; v7's resolve by the order its
    blocks were visited in); ragged grids, frames smaller than a tile, disc and square windows (reach 2 .. 16), runs of 1 .. 6 pairs,
    2 KiB rows; no timed-out wait; then against float64 like every other pair kernel."""
    from fgvc_amd import ops
    H, W, Tn, nr, mm, k = case
    feats = _pairs_case(dev, H, W, Tn, seed=H * 131 + W, kind="smooth" if H == 64 else "gauss")
    mask = ops.MaskSpec.from_neighbor_range(nr, mm)
    rows = [(q, kk, True) for q in range(1, Tn) for kk in range(max(0, q - 6), q)]
    pairs = ops.make_pairs(rows, dev)
    for fmt, sp in (("f16f6", ops.split_f16f6p(feats)), ("f16f6x", ops.split_f16f6x(feats))):
        i7, s7 = ops.pair_topk_split(sp, sp, pairs, H, W, H, W, mask, k, all_masked=True, fmt=fmt)
        ops.set_option("pair_f16_debug", V8)
        try:
            i8, s8 = ops.pair_topk_split(sp, sp, pairs, H, W, H, W, mask, k, all_masked=True, fmt=fmt)
            torch.cuda.synchronize()
        finally:
            ops.set_option("pair_f16_debug", 0)
        assert not ops.pair_f16x3_timed_out()
        assert torch.equal(s7, s8), f"{fmt}: {int((s7 != s8).sum())} scores differ"
        differ = (i7 != i8).any(-1)
        for a, b, s in zip(i7[differ].tolist(), i8[differ].tolist(), s7[differ].tolist()):
            gone = [x for x in a if x not in b]
            assert (all(s[a.index(x)] == s[-1] for x in gone) if gone else len(set(s)) < len(s)), (a, b, s)
    for pi, (q, kk, _) in enumerate(rows[:3]):
        dense = (feats[kk].double().cpu() @ feats[q].double().cpu().T) / TAU
        m = O.mask_slab(H, W, H, W, 1, torch.arange(H * W), nr, mm)
        st = O.check_topk(dense.masked_fill(~m, float("-inf")), i8[pi].cpu().long(), s8[pi].cpu() / TAU, k, tol=1e-3, gap=GAP)
        assert st["exact"] >= st["clear"] and st["max_score_err"] < 2e-4


def test_pair_v8_fail_closed(dev):
    """the injected protocol fault (pair_f16_debug & 4096) poisons v8's lists as it does v7's, and raises the flag"""
    from fgvc_amd import ops
    H, W, Tn = 24, 40, 3
    feats = _pairs_case(dev, H, W, Tn, seed=11)
    sp = ops.split_f16f6p(feats)
    pairs = ops.make_pairs([(1, 0, True), (2, 0, True), (2, 1, True)], dev)
    mask = ops.MaskSpec.from_neighbor_range(30)
    ops.set_option("pair_f16_debug", V8 + 4096)
    try:
        idx, score = ops.pair_topk_split(sp, sp, pairs, H, W, H, W, mask, 10, all_masked=True, fmt="f16f6")
        torch.cuda.synchronize()
    finally:
        ops.set_option("pair_f16_debug", 0)
    assert ops.pair_f16x3_timed_out()
    assert bool((idx == 0).all()) and bool(torch.isinf(score).all()) and bool((score > 0).all())
    assert not ops.pair_f16x3_timed_out()
