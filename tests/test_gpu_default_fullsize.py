"""The SHIPPED default -- split_f16f6x() bank -> fgvc_pair_topk_f16f6x -> fgvc_merge_refine_topk_f32 -- held to float64 at the BASELINE
sizes (until round 5 the full-size float64 checks ran on the three-product kernel only):

  cfg2  8 x 120 x 214 (27 pairs), cfg4  64 x 128 x 128 (363 pairs in 63 runs), cfg5  24 x 180 x 320 (123 pairs)

per pair (sampled pairs, sampled queries): every index inside the disc, lists in canonical order, every score within half of
ops.REFINE_EPS of the float64 product of the rows it names (the bound the refining merge assumes), the exact float64 top-k wherever its
ranks are further apart than 2 REFINE_EPS; per merged row (first, a middle and the last frame): the float64 top-k of the union of the
key slots EXACTLY -- index for index, in order -- wherever the float64 ranks 1..k+1 are 1e-5 logit apart (SURVEY section 7's policy),
scores within 1e-3 logit (north_star), weights = softmax of the logits.
Reference: local_attention.py:321-356 (fp32 einsum + topk over T * HW), mask affinity_utils.py:98-109."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TAU, K, C = 0.07, 10, 256


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fgvc_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _structured(dev, n, H, W, seed, noise=0.6):
    from fgvc_amd import ops
    g = torch.Generator(device=dev).manual_seed(seed)
    base = torch.randn(1, C, H // 8 + 1, W // 8 + 1, generator=g, device=dev)
    smooth = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    return torch.cat([ops.normalize_to_hwc(smooth + noise * torch.randn(1, C, H, W, generator=g, device=dev)) for _ in range(n)], 0)


@pytest.mark.parametrize("case", [("cfg2", 120, 214, 8, 27, (0, 13, 26), (1, 4, 7)),
                                  ("cfg4", 128, 128, 64, 363, (0, 14, 180, 362), (1, 30, 63)),
                                  ("cfg5", 180, 320, 24, 123, (0, 61, 122), (2, 23))])
def test_default_route_vs_float64(dev, case):
    from fgvc_amd import engine, ops
    name, H, W, T, n_pairs, sample_pairs, sample_frames = case
    HW = H * W
    cfg = engine.TrackerConfig(pair_split_fmt="f16f6")                  # what VanillaTracker.engine_config() picks behind its default encoder
    assert cfg.bank_fmt == "f16f6x" and cfg.pair_refine
    plan = engine.plan_clip(T, [0], cfg)
    assert len(plan.pairs) == n_pairs
    clip = _structured(dev, T, H, W, seed=1000 + T)
    bank = ops.split_f16f6x(clip)
    pl = engine.run_pairs(bank, H, W, plan, cfg)
    assert not ops.pair_f16x3_timed_out() and pl.exact is bank and pl.idx.shape == (n_pairs, HW, K)
    idx, score = pl.idx, pl.score
    assert int(idx.min()) >= 0 and int(idx.max()) < HW
    qy = (torch.arange(HW, device=dev) // W).view(1, HW, 1)
    qx = (torch.arange(HW, device=dev) % W).view(1, HW, 1)
    for c0 in range(0, n_pairs, 32):                                     # chunks: int64 temporaries
        sl = slice(c0, c0 + 32)
        d2 = (idx[sl] // W - qy) ** 2 + (idx[sl] % W - qx) ** 2
        assert int(d2.max()) <= cfg.mask.r2max                           # inside the disc
        ds = score[sl][..., 1:] - score[sl][..., :-1]
        assert float(ds.max()) <= 0.0                                    # descending
        tie = ds == 0
        assert bool((idx[sl][..., 1:][tie] > idx[sl][..., :-1][tie]).all())     # canonical order among equal (quantised) scores
    g = torch.Generator().manual_seed(7)
    sample = torch.cat([torch.tensor([0, W - 1, HW - W, HW - 1]), torch.randint(0, HW, (508,), generator=g)]).to(dev)
    ky = (torch.arange(HW, device=dev) // W).view(-1, 1)
    kx = (torch.arange(HW, device=dev) % W).view(-1, 1)
    inside = ((ky - (sample // W).view(1, -1)) ** 2 + (kx - (sample % W).view(1, -1)) ** 2) <= cfg.mask.r2max
    err, n_clear = 0.0, 0
    for p in sample_pairs:
        qf, kf, _ = plan.pairs[p]
        dots = torch.einsum("qc,qkc->qk", clip[qf][sample].double(), clip[kf][idx[p][sample].long()].double())
        err = max(err, float((dots - score[p][sample].double()).abs().max()))
        full = (clip[kf].double() @ clip[qf][sample].double().t()).masked_fill(~inside, float("-inf"))
        tv, ti = full.topk(K + 1, dim=0)
        clear = (tv[:-1] - tv[1:]).min(0).values > 2 * ops.REFINE_EPS  # the pair kernel alone: exact where the ranks are beyond its error
        n_clear += int(clear.sum())
        assert torch.equal(idx[p][sample].t().long()[:, clear], ti[:K][:, clear])
    assert err < 0.5 * ops.REFINE_EPS, err
    assert n_clear > 0.5 * len(sample_pairs) * sample.numel()
    # the merged + refined lists against the float64 top-k over the row's key slots
    tk = engine.merge_pairs(pl, cfg)
    stats = tk.refine_stats.cpu().tolist()
    stats[0] -= stats[7]                                     # (word 0 counts queued items: the unbiased sample's queries, word 7, are not re-scored queries)
    assert torch.allclose(tk.weight.sum(-1), torch.ones_like(tk.weight[..., 0]), atol=1e-5)
    rows_clear = rows_all = 0
    worst = 0.0
    f64 = None
    for fr in sample_frames:
        row = plan.out_rows[(0, fr)]
        ks = [kf for kf, pid in zip(plan.slot_frame[row], plan.slot_pair[row]) if pid >= 0]
        q64 = clip[fr][sample].double()
        dense = torch.cat([(clip[kf].double() @ q64.t()).masked_fill(~inside, float("-inf")) for kf in ks], 0) / TAU      # (T' HW, S)
        dv, di = dense.topk(K + 1, dim=0)
        # (torch.topk leaves ties unordered: a frame in two slots ties every entry with its twin -> canonical order by hand)
        key = torch.stack([-dv, di.double()], -1)                        # sort by (score desc, index asc)
        order = torch.argsort(key[..., 1], dim=0, stable=True)
        dv, di = dv.gather(0, order), di.gather(0, order)
        order = torch.argsort(dv, dim=0, descending=True, stable=True)
        dv, di = dv.gather(0, order), di.gather(0, order)
        gaps = (dv[:-1] - dv[1:])
        twin = (di[:-1] % HW == di[1:] % HW) & (gaps == 0)               # exact twins are ordered by slot, not by score
        clear = (gaps.masked_fill(twin, float("inf")).min(0).values > 1e-5)
        got_i, got_l = tk.idx[row][sample].t().long(), tk.logit[row][sample].t().double()
        assert torch.equal(got_i[:, clear], di[:K][:, clear]), (name, fr, int((got_i[:, clear] != di[:K][:, clear]).any(0).sum()))
        worst = max(worst, float((got_l - dv[:K]).abs().max()))
        rows_clear += int(clear.sum())
        rows_all += sample.numel()
        w = torch.softmax(tk.logit[row][sample].double(), 1)
        assert float((w - tk.weight[row][sample].double()).abs().max()) < 1e-6
    assert worst < 1e-3 and rows_clear > 0.9 * rows_all, (worst, rows_clear, rows_all)
    n_q = len(plan.slot_pair) * HW
    print(f"{name}: pair score error {err:.2e} (eps {ops.REFINE_EPS:.0e}); merged rows: {rows_clear} of {rows_all} sampled queries clear at 1e-5, all exact; "
          f"max logit error {worst:.2e}; re-scored {stats[0]} of {n_q} queries ({100.0 * stats[0] / n_q:.1f} %), from scratch {stats[1]}, beyond the queue {stats[3]}")
