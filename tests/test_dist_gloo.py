"""CPU, world_size 2, gloo: the clip-sharding choreography of fgvc_amd/dist.py (frame ranges, halo,
first-frame broadcast, all_gather order, replicated sweep) with an ORACLE-backed compute backend,
checked against the un-sharded oracle driver.  The HIP backend runs the same code on the GPUs."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fgvc_oracle as O


class OracleBackend:
    """Same interface as fgvc_amd.dist.HipBackend, torch CPU arithmetic (test infrastructure)."""

    def encode(self, frames, out=None):             # "frames" are already feature maps (n,C,Hf,Wf) in this test
        n, C, Hf, Wf = frames.shape
        f = O.l2_normalize(frames, 1).flatten(2).transpose(1, 2).contiguous()
        if out is not None and out.shape == f.shape and out.dtype == f.dtype:      # rows of the caller's local bank (HipBackend.encode(out=))
            out.copy_(f)
            return out, Hf, Wf
        return f, Hf, Wf

    def affinity(self, bank, Hf, Wf, plan, cfg, phases=None):
        """`phases` as HipBackend.affinity: rows made only of pairs in `first` are computed BEFORE between() (= before the halo
        frames are awaited), the others after."""
        C = bank.shape[-1]
        n_rows = len(plan.slot_pair)
        first, between = phases if phases is not None else (range(len(plan.pairs)), lambda: None)
        first = set(first)
        early = [r for r in range(n_rows) if all(p in first for p in plan.slot_pair[r] if p >= 0)]
        late = [r for r in range(n_rows) if r not in early]
        self.phase_rows = (len(early), len(late))
        out = {}
        for r in early:
            out[r] = self._row(bank, Hf, Wf, plan, cfg, C, r)
        between()
        for r in late:
            out[r] = self._row(bank, Hf, Wf, plan, cfg, C, r)
        return torch.stack([out[r][0] for r in range(n_rows)], 0), torch.stack([out[r][1] for r in range(n_rows)], 0)

    def _row(self, bank, Hf, Wf, plan, cfg, C, row):
        if True:
            pids = [p for p in plan.slot_pair[row] if p >= 0]
            q = plan.pairs[pids[0]][0]
            ks = [plan.pairs[p][1] for p in pids]
            qm = bank[q].t().reshape(C, Hf, Wf)
            km = torch.stack([bank[k].t().reshape(C, Hf, Wf) for k in ks], 1)
            idx, logit = O.affinity_topk(qm, km, cfg.topk, cfg.temperature, neighbor_range=cfg.neighbor_range,
                                         normalize=False)
            return idx.to(torch.int32), O.topk_weights(logit)

    def sweep(self, idx, weight, slot_frame, plan, start, pts, Hf, Wf, h, w, cfg):
        T = plan.n_frames
        full0, lab0 = O.gaussian_labels(pts, h, w, h // Hf)
        P = pts.shape[0]
        labels = {start: lab0.reshape(P, -1)}
        preds = [full0]
        for f in range(start + 1, T):
            row = plan.out_rows[(start, f)]
            frames = slot_frame[row].tolist()
            i = idx[row].long()
            slot, pix = i // (Hf * Wf), i % (Hf * Wf)
            zero = torch.zeros_like(lab0.reshape(P, -1))                                   # padded slots are never indexed
            val = torch.stack([labels.get(frames[s], zero) for s in range(len(frames))], 0)  # (t_max,P,HW)
            g = val[slot, :, pix]                                                          # (HW,k,P)
            out = (g * weight[row].unsqueeze(-1)).sum(1).t()                               # (P,HW)
            labels[f] = out
            preds.append(O.upsample_bilinear(out.reshape(P, Hf, Wf), h, w))
        coords = O.img2coord(torch.stack(preds, 0).numpy())
        return torch.from_numpy(coords).permute(2, 1, 0)


def _free_port():
    """A rendezvous FILE, not a TCP port: a port probed free here can be taken before rank 0 binds it."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="fgvc_rdzv_")
    os.close(fd)
    os.unlink(path)
    return path


def _worker(rank, world, port, feats, qp, q, halo="exchange", precede=5, members=None, p2p_order=None):
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    if p2p_order is not None:
        os.environ["FGVC_P2P_ORDER"] = p2p_order
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from fgvc_amd import dist as D
    from fgvc_amd.engine import TrackerConfig
    cfg = TrackerConfig(neighbor_range=8, regroup=True, precede_frames=precede)
    h, w = feats.shape[-2] * 2, feats.shape[-1] * 2
    group = dist.new_group(members) if members is not None else None     # (every process of the job creates the group)
    if members is not None and rank not in members:
        q.put((rank, None, "not a member"))
        dist.barrier()
        dist.destroy_process_group()
        return
    try:
        timing = D.Timing()
        be = OracleBackend()
        cache = {}
        traj, order = _run(D, be, feats, qp, cfg, h, w, halo=halo, timing=timing, group=group, cache=cache)
        # second call, cached schedule: the bank exists before the encoder runs, the encoder writes into it, and with halo="exchange"
        # every rank posts its messages after its last `precede` frames and before the rest (one order of collectives on all ranks --
        # a rank that decided differently would hang here); same trajectories
        timing = D.Timing()
        traj2, order2 = _run(D, be, feats, qp, cfg, h, w, halo=halo, timing=timing, group=group, cache=cache)
        assert torch.equal(traj2, traj) and torch.equal(order2, order)
        assert cache["schedule"].get("bank_in_place", 0) == 1
        if halo == "auto":               # resolved when the schedule was built, from one measured constant that every rank holds
            why = cache["schedule"]["halo_why"]
            assert cache["schedule"]["halo"] == why["mode"] and why["link_gbps"] > 0 and D.measure_link_gbps(group) == why["link_gbps"]
            halo = cache["schedule"]["halo"]
        assert cache["schedule"].get("halo_early", 0) == (1 if halo == "exchange" else 0)
        rep = timing.report()
        rank = dist.get_rank(group) if group is not None else rank
        world = len(members) if members is not None else world
        short = feats.shape[0] <= world + 1                                  # a video with fewer query frames than ranks: some ranges are empty
        if halo == "exchange" and rank > 0 and not short:   # some rows were computed while the halo was in flight, some had to wait for it
            assert (be.phase_rows[0] > 0 or world > 2) and be.phase_rows[1] > 0 and "halo_wait" in rep, (be.phase_rows, rep)
        if halo == "recompute":
            assert be.phase_rows[1] == 0
        assert set(rep) >= {"encode", "broadcast_first_frames", "halo_exchange", "affinity", "all_gather_lists", "sweep_readout"}
        q.put((rank, traj, order))
    except Exception as e:  # surface the failure instead of letting the parent time out
        q.put((rank, repr(e), None))
        raise
    dist.barrier()
    dist.destroy_process_group()


def _run(D, backend, feats, qp, cfg, h, w, **kw):
    """track_points_sharded takes h,w from rgbs; wrap the feature clip so the last two dims read as the image."""
    class Wrapped:
        def __init__(self, f):
            self.f = f
            self.shape = (f.shape[0], 3, h, w)
            self.device = f.device

        def __getitem__(self, s):
            class _S:
                def __init__(s2, t):
                    s2.t = t

                def to(s2, dev):
                    return s2.t
            return _S(self.f[s])
    return D.track_points_sharded(backend, Wrapped(feats), qp, cfg, **kw)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,halo,precede,members,T,p2p", [(2, "exchange", 5, None, 11, None), (2, "recompute", 5, None, 11, None), (2, "auto", 5, None, 11, None),
                                                                (3, "exchange", 5, None, 11, None), (3, "exchange", 5, [1, 2], 11, None),
                                                                (3, "exchange", 5, None, 3, None), (4, "exchange", 5, [0, 2, 3], 3, "posted"),
                                                                (3, "exchange", 2, None, 11, "posted")])
def test_multi_rank_sharding_matches_unsharded_oracle(world, halo, precede, members, T, p2p):
    """2 ranks in both halo modes; 3 ranks with clips SHORTER than the halo (a rank then needs frames of two other ranks); a video
    sharded over a SUB-GROUP whose members are not ranks 0..n-1 of the job (the schedule counts ranks inside the group, the
    collectives take global ranks); a video with FEWER query frames than ranks (the last range is empty: that rank encodes nothing,
    posts no message, and still takes part in every collective -- also on the cached-schedule second call with its early halo, also in a
    sub-group); and both orders of the point-to-point batch (canonical = by peer, the default; FGVC_P2P_ORDER=posted)."""
    g = torch.Generator().manual_seed(21)
    C, Hf, Wf = 16, 10, 12
    feats = torch.randn(T, C, Hf, Wf, generator=g)
    qp = torch.tensor([[0., 5., 7.], [0., 17.3, 11.2], [4., 9.5, 3.25], [4., 20., 15.], [7., 2.2, 18.8]])
    if T < 8:
        qp[:, 0] = torch.tensor([0., 0., 1., 1., 0.])[: qp.shape[0]] if T > 2 else 0.
    h, w = 2 * Hf, 2 * Wf
    # un-sharded expectation
    exp = torch.zeros(T, qp.shape[0], 2, dtype=torch.float64)
    col = 0
    for s in sorted(set(qp[:, 0].int().tolist())):
        sel = (qp[:, 0] == s).nonzero().flatten()
        exp[s:, col:col + sel.numel()] = O.forward_test_main(feats[s:], qp[sel, 1:], h, w, neighbor_range=8, precede_frames=precede)[0]
        col += sel.numel()
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, feats, qp, q, halo, precede, members, p2p)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=400) for _ in range(world)], key=lambda r: r[0])
    assert all(r[2] is not None for r in res), res
    res = [r for r in res if r[1] is not None]                          # (processes outside the sub-group report None)
    assert len(res) == (len(members) if members is not None else world)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, traj, order in res:
        assert order.tolist() == [i for s_ in sorted(set(qp[:, 0].int().tolist())) for i in (qp[:, 0] == s_).nonzero().flatten().tolist()]
        assert torch.allclose(traj, exp, atol=1e-6), (rank, float((traj - exp).abs().max()))
    assert all(torch.equal(res[0][1], r[1]) for r in res[1:])          # replicated sweep is deterministic across ranks


@pytest.mark.timeout(900)
def test_eight_ranks_with_the_cfg4_schedule():
    """World size 8 -- the node the north_star names -- on the schedule of BASELINE configs[3] (a 64-frame video, precede_frames 5):
    first the schedule itself (pure host logic: 8 query frames per rank, 33 unique pairs on rank 0, 48 on the middle ranks, one 5-frame halo message per boundary, every needed frame arriving exactly once), then the eight gloo ranks with the oracle
    backend on a tiny grid (both calls: the second with the cached schedule, the bank in place and the early halo) against the
    un-sharded oracle driver.  (the largest world size rehearsed anywhere had been 4.)"""
    from fgvc_amd import dist as D
    from fgvc_amd import engine
    T, world = 64, 8
    cfg = engine.TrackerConfig(neighbor_range=8, regroup=True, precede_frames=5)
    ranges = D.shard_frames(T, world, first=1)
    assert ranges == [(8 * r + 1, min(8 * r + 9, T)) for r in range(8)]                      # 63 query frames: 8 per rank, 7 on the last
    own = D.own_ranges(ranges, 0)
    msgs = D.halo_messages(ranges, own, 0, 5)
    assert msgs == [(r, r + 1, 8 * r + 4, 8 * r + 9) for r in range(7)]                     # the last five frames of every clip, to the next rank
    # unique (query, key) pairs: frames 1..8 have 1,2,3,4,5,6,6,6 key frames (33), every later frame 6
    assert [len(engine.plan_clip(T, [0], cfg, frame_range=r).pairs) for r in ranges] == [33] + [48] * 6 + [42]
    g = torch.Generator().manual_seed(64)
    C, Hf, Wf = 8, 6, 8
    feats = torch.randn(T, C, Hf, Wf, generator=g)
    qp = torch.tensor([[0., 3., 4.], [0., 11.3, 7.2], [0., 6.5, 9.25]])
    h, w = 2 * Hf, 2 * Wf
    exp = O.forward_test_main(feats, qp[:, 1:], h, w, neighbor_range=8, precede_frames=5)[0]
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, feats, qp, q, "exchange", 5, None, None)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=700) for _ in range(world)], key=lambda r: r[0])
    assert all(r[2] is not None for r in res), res
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, traj, order in res:
        assert order.tolist() == [0, 1, 2]
        assert torch.allclose(traj, exp, atol=1e-6), (rank, float((traj - exp).abs().max()))
    assert all(torch.equal(res[0][1], r[1]) for r in res[1:])


def test_halo_auto_decision():
    """halo="auto" (round 6) as a pure function of the schedule: at the shapes of BASELINE configs[1] x 8 ranks (a 64-frame 480p video,
    8 query frames per rank) the 131 MB halo message takes 1.7 ms over one xGMI direction and hides behind the receiver's 33 halo-free
    pairs + its three front frames -> exchange; without the early halo 0.3 ms of it is exposed, still cheaper than five encoder frames;
    over a 10 GB/s link the exposed 11 ms lose to 1.8 ms of recompute; at world size 1 there is no boundary."""
    from fgvc_amd import dist as D
    from fgvc_amd.engine import TrackerConfig
    cfg = TrackerConfig()
    cost = D.halo_cost_model(480, 854)
    assert cost["frame_bytes"] == 120 * 214 * 256 * 4 and abs(cost["enc_frame_s"] - 0.36e-3) < 1e-9
    d = D.choose_halo(64, 8, [0], cfg, cost, link_gbps=77.0)
    assert d["mode"] == "exchange" and len(d["boundaries"]) == 7
    b = d["boundaries"][0]
    assert b["halo_frames"] == 5 and b["message_bytes"] == 5 * cost["frame_bytes"] and b["halo_free_pairs"] == 33
    assert 1.6e-3 < b["transfer_s"] < 1.8e-3 and b["exposed_s"] == 0.0
    d2 = D.choose_halo(64, 8, [0], cfg, cost, link_gbps=77.0, early_halo=False)
    assert d2["mode"] == "exchange" and 0.2e-3 < d2["boundaries"][0]["exposed_s"] < 0.4e-3
    d3 = D.choose_halo(64, 8, [0], cfg, cost, link_gbps=10.0)
    assert d3["mode"] == "recompute" and d3["boundaries"][0]["exposed_s"] > d3["boundaries"][0]["recompute_s"]
    assert D.choose_halo(8, 1, [0], cfg, cost)["boundaries"] == []
    # query groups that start later: the frames in front of a clip that another rank encodes never reach below the first start
    d4 = D.choose_halo(16, 4, [3], cfg, cost)
    assert all(bd["halo_frames"] <= 5 for bd in d4["boundaries"])


def test_halo_message_plan():
    from fgvc_amd import dist as D
    ranges = D.shard_frames(17, 4, first=1)                              # [(1,5),(5,9),(9,13),(13,17)]
    own = D.own_ranges(ranges, 0)
    assert own == [(0, 5), (5, 9), (9, 13), (13, 17)]
    msgs = D.halo_messages(ranges, own, 0, 5)
    # a 4-frame clip is shorter than the 5-frame halo: rank 2 needs frames 4..8 = one frame of rank 0 and all of rank 1's
    assert msgs == [(0, 1, 0, 5), (0, 2, 4, 5), (1, 2, 5, 9), (1, 3, 8, 9), (2, 3, 9, 13)]
    # every needed frame arrives exactly once
    for dst, (lo, hi) in enumerate(ranges):
        got = sorted(f for (s, d, a, b) in msgs if d == dst for f in range(a, b))
        assert got == list(range(max(0, lo - 5), own[dst][0]))
    assert D.halo_messages(D.shard_frames(9, 8, first=1), D.own_ranges(D.shard_frames(9, 8, first=1), 0), 0, 5)[:2] == [(0, 1, 0, 2), (0, 2, 0, 2)]


def _collect_worker(rank, world, port, q):
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from fgvc_amd import apis
    # rank r ran videos r, r + world, ... (mmpt/datasets/samplers/distributed_sampler.py:53); video i's result carries i
    mine = [tuple(torch.full((1, 2, 3), float(i)) for _ in range(5)) for i in range(rank, 7, world)]
    out = apis.collect_results(mine, size=7)
    # ... and the reference's default: part_{rank}.pkl files in a directory rank 0 makes (mmpt/apis/test.py:131-189)
    out_f = apis.collect_results_cpu(mine, size=7)
    assert (out is None) == (out_f is None)
    if out is not None:
        assert [float(r[2].flatten()[0]) for r in out_f] == [float(r[2].flatten()[0]) for r in out]
    q.put((rank, None if out is None else [float(r[2].flatten()[0]) for r in out]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_collect_results_two_ranks():
    """apis.collect_results (mmpt/apis/test.py:131-236): rank 0 gets every video's 5-tuple back in dataset order."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_collect_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=200) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0] and res[1] is None


def test_single_process_path_equals_engine_semantics():
    """world=1 (no process group): same function, no collectives."""
    from fgvc_amd import dist as D
    from fgvc_amd.engine import TrackerConfig
    g = torch.Generator().manual_seed(3)
    T, C, Hf, Wf = 6, 8, 8, 8
    feats = torch.randn(T, C, Hf, Wf, generator=g)
    qp = torch.tensor([[0., 3., 4.], [2., 10., 9.]])
    cfg = TrackerConfig(neighbor_range=6, regroup=True)
    traj, order = _run(D, OracleBackend(), feats, qp, cfg, 16, 16)
    e0 = O.forward_test_main(feats, qp[:1, 1:], 16, 16, neighbor_range=6)[0]
    e2 = O.forward_test_main(feats[2:], qp[1:, 1:], 16, 16, neighbor_range=6)[0]
    assert torch.allclose(traj[:, :1], e0, atol=1e-6) and torch.allclose(traj[2:, 1:], e2, atol=1e-6)
    assert float(traj[:2, 1:].abs().max()) == 0.0
