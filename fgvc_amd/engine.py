"""Label-propagation engine: the schedule the reference's driver runs frame by frame
(vanilla_tracker.py:305-412), restated as three device phases.

  phase 1  correlation + top-k for EVERY (query frame, key frame) pair of the clip in one launch
           (indices/scores depend on features only, so all frames go in parallel);
  phase 2  per output frame: merge its key slots' lists, temperature, softmax   (one launch);
  phase 3  the only sequential part: label propagation frame by frame (tiny gathers), then the
           fused upsample + soft-argmax read-out.

`plan_*` are pure host functions (tested on CPU); `run_*` enqueue HIP work and never synchronise.

De-duplication: the reference re-encodes and re-correlates the whole tail of
the clip once per distinct query time t0 (vanilla_tracker.py:249-295).  A (query frame, key frame)
pair's top-k does not depend on t0, and top-k of a union of slots == top-k of the union of the
slots' top-k lists, so pairs are computed ONCE and every group merges the pairs it needs;
frame t0 in slot 0 and again as a preceding frame (:353-362) costs one pair, not two.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .ops import MaskSpec


@dataclass
class TrackerConfig:
    """test_cfg_<task> keys the path reads (configs/eval/res18_d1_eval.py:12-58, vanilla_tracker.py:246,330-392).

    The reference reads `with_first` twice with DIFFERENT defaults: `.get('with_first', False)` decides whether points are
    regrouped by query time (:246) and `.get('with_first', True)` whether the group's first frame is key slot 0 (:353).
    They are two fields here (`regroup`, `with_first`) so that a config without the key behaves as the reference does:
    one group from frame 0, first frame in slot 0."""
    precede_frames: int = 5
    topk: int = 10
    temperature: float = 0.07
    neighbor_range: Optional[int] = 30
    mask_mode: str = "circle"
    with_first: bool = True            # key slot 0 = first frame of the group (:353)
    regroup: bool = False              # one pass per distinct query time (:246)
    with_first_neighbor: bool = True
    with_norm: bool = True
    mode: str = "softmax"
    sim_mode: str = "dot_product"      # or 'l2-distance' (local_attention.py:324-327), normalised features only
    test_mode: str = "v1"              # anything else = masked_attention_efficient_v2 (:379-392)
    sigma: float = 6.0
    pair_precision: str = "auto"   # ops.pair_topk_auto: "auto" | "f32" | "split" (not a reference key)
    pair_split_fmt: str = "f16"    # operand format of the split pair kernel: "f16" = split_f16x2 rows -> fgvc_pair_topk_f16x3 (three f16 products, 1e-7-grade),
                                   # "f16f6" = split_f16f6p rows -> fgvc_pair_topk_f16f6 (f16 + FP6 cross terms: half the matrix work, ~6e-5 logit);
                                   # VanillaTracker picks "f16f6" when its encoder computes in f16f8 / f16f6 and the mask allows it (engine_config)
    pair_refine: bool = True       # with "f16f6": the bank carries the exact f32 channels behind every row (split_f16f6x: 2 KiB rows) and the
                                   # merge re-scores near-ties from them (fgvc_merge_refine_topk_f32): the lists are the exact top-k in the exact
                                   # order wherever the approximate scores are within `pair_refine_eps` of the exact products (round 5: the
                                   # reference's own lists, profiles/r05_precision_ledger.json).  False: round 4's plain merge of approximate scores
    pair_refine_eps: float = ops.REFINE_EPS

    @staticmethod
    def from_test_cfg(cfg) -> "TrackerConfig":
        """Every key the reference's driver reads is either honoured or refused loudly -- never dropped."""
        g = cfg.get
        neighbor_range, mask_mode = g("neighbor_range", None), g("mask_mode", "circle")
        with_first_neighbor = g("with_first_neighbor", True)
        test_mode = g("test_mode", "v1")
        if test_mode != "v1":
            # masked_attention_efficient_v2: always the disc `dist < neighbor_range // 2` (local_attention.py:463-467), on every
            # key slot (non_mask_len is accepted and ignored there, :467-470); `mask_mode` is not read on this branch
            if neighbor_range is None:
                raise ValueError("test_mode != 'v1' needs neighbor_range (vanilla_tracker.py:384 computes neighbor_range // 2)")
            mask_mode, with_first_neighbor = "circle", True
        # (_v2 accepts `sim_mode` and never reads it, local_attention.py:453-455)
        sim_mode = g("sim_mode", "dot_product") if test_mode == "v1" else "dot_product"
        with_norm = g("with_norm", True)
        if sim_mode not in ("dot_product", "l2-distance"):
            raise NotImplementedError(f"fgvc_amd: sim_mode={sim_mode!r} (the reference knows 'dot_product' and 'l2-distance')")
        if sim_mode == "l2-distance" and not with_norm:
            raise NotImplementedError("fgvc_amd: sim_mode='l2-distance' is on the accelerated path for normalised features only")
        return TrackerConfig(
            precede_frames=g("precede_frames", 5), topk=g("topk", 10), temperature=g("temperature", 0.07),
            neighbor_range=neighbor_range, mask_mode=mask_mode,
            with_first=bool(g("with_first", True)), regroup=bool(g("with_first", False)),
            with_first_neighbor=with_first_neighbor, with_norm=with_norm, sim_mode=sim_mode, test_mode=test_mode,
            pair_precision=g("pair_precision", "auto"), pair_split_fmt=g("pair_split_fmt", "f16"),   # from here on: extension keys
            pair_refine=bool(g("pair_refine", True)), pair_refine_eps=float(g("pair_refine_eps", ops.REFINE_EPS)))

    @property
    def bank_fmt(self) -> str:
        """The format get_feats_hwc(split=True) / run_pairs() build the bank in: split_f16f6x() rows where the f16 + FP6 pair kernel runs
        with the refining merge behind it, else what `pair_split_fmt` names."""
        return "f16f6x" if (self.pair_split_fmt == "f16f6" and self.pair_refine) else self.pair_split_fmt

    @property
    def mask(self) -> MaskSpec:
        return MaskSpec.from_neighbor_range(self.neighbor_range, self.mask_mode)

    def softmax_temperature(self, channels: int) -> float:
        """Divisor of the raw dot products of L2-normalised rows before the softmax over the k survivors.
        'l2-distance' (local_attention.py:324-327): affinity = (2 k.q - |k|^2) / sqrt(C) with |k| = 1, no temperature: the
        same ranking as the dot product, and softmax((2 d - 1) / sqrt(C)) = softmax(d / (sqrt(C) / 2))."""
        if self.sim_mode == "l2-distance":
            if self.mode != "softmax":
                raise NotImplementedError("fgvc_amd: sim_mode='l2-distance' with mode='cosine'")
            return math.sqrt(channels) / 2.0
        return float(self.temperature)


def key_slots(frame: int, start: int, precede_frames: int, with_first: bool) -> List[int]:
    """Clip-absolute key frames of query frame `frame` in the group that starts at `start`
    (vanilla_tracker.py:346-362 with frame numbers shifted by `start`)."""
    ks = list(range(max(start, frame - precede_frames), frame))
    return ([start] + ks) if with_first else ks


@dataclass
class Plan:
    """Host-side schedule for one clip."""
    n_frames: int
    starts: List[int]                                   # distinct query times, ascending
    pairs: List[Tuple[int, int, bool]]                  # unique (query frame, key frame, masked)
    out_rows: Dict[Tuple[int, int], int]                # (start, frame) -> row of slot tables
    slot_pair: List[List[int]]                          # row -> pair id per key slot (-1 pad)
    slot_frame: List[List[int]]                         # row -> clip frame per key slot (0 pad)
    t_max: int = 0
    _dev: Dict[str, tuple] = field(default_factory=dict, repr=False)

    def tables(self, dev):
        """Device copies of the schedule (pairs int32 (n,4), slot_pair / slot_frame int32 (rows,t_max)).
        A plan depends only on (n_frames, starts, cfg), so the tables are uploaded once and reused."""
        key = str(dev)
        if key not in self._dev:
            rows = len(self.slot_pair)
            self._dev[key] = (
                ops.make_pairs(self.pairs, dev),
                torch.tensor(self.slot_pair, dtype=torch.int32, device=dev).reshape(rows, self.t_max),
                torch.tensor(self.slot_frame, dtype=torch.int32, device=dev).reshape(rows, self.t_max))
        return self._dev[key]


def plan_clip(n_frames: int, starts: Sequence[int], cfg: TrackerConfig,
              frame_range: Optional[Tuple[int, int]] = None) -> Plan:
    """Build the pair list and slot tables.  `frame_range` = [lo, hi) restricts the QUERY frames
    (clip sharding across GPUs: each rank plans only its own frames)."""
    starts = sorted(set(int(s) for s in starts))
    lo, hi = frame_range if frame_range is not None else (0, n_frames)
    non_mask_len = 0 if cfg.with_first_neighbor else 1
    pair_id: Dict[Tuple[int, int, bool], int] = {}
    rows: Dict[Tuple[int, int], int] = {}
    slot_pair, slot_frame = [], []
    t_max = cfg.precede_frames + (1 if cfg.with_first else 0)
    t_max = max(t_max, 1)
    for s in starts:
        for f in range(max(s + 1, lo), min(n_frames, hi)):
            ks = key_slots(f, s, cfg.precede_frames, cfg.with_first)
            sp, sf = [], []
            for t, kf in enumerate(ks):
                masked = not (t < non_mask_len) and not cfg.mask.is_none
                key = (f, kf, masked)
                if key not in pair_id:
                    pair_id[key] = len(pair_id)
                sp.append(pair_id[key])
                sf.append(kf)
            pad = t_max - len(ks)
            rows[(s, f)] = len(slot_pair)
            slot_pair.append(sp + [-1] * pad)
            slot_frame.append(sf + [0] * pad)
    pairs = [k for k, _ in sorted(pair_id.items(), key=lambda kv: kv[1])]
    return Plan(n_frames, starts, pairs, rows, slot_pair, slot_frame, t_max)


@dataclass
class DeviceTopk:
    """phase 1+2 results on the device."""
    plan: Plan
    idx: torch.Tensor        # (rows, HW, k) int32   slot*HW + pixel
    logit: torch.Tensor      # (rows, HW, k)
    weight: torch.Tensor     # (rows, HW, k)
    slot_frame: torch.Tensor  # (rows, t_max) int32
    row_map: Optional[Dict[int, int]] = None    # plan row -> row of idx / weight / slot_frame when only some rows were merged
    refine_stats: Optional[torch.Tensor] = None # the refining merge's int32 counters (3,) on the device: queries re-scored, of them from scratch, candidates

    def row(self, plan_row: int) -> int:
        return plan_row if self.row_map is None else self.row_map[plan_row]


@dataclass
class PairLists:
    """phase 1 results on the device: per (query frame, key frame) pair the top-k list of every query pixel."""
    plan: Plan
    idx: torch.Tensor        # (pairs, HW, k) int32   pixel index in the key frame
    score: torch.Tensor      # (pairs, HW, k)
    HW: int
    channels: int = 256      # un-padded feature channels (only 'l2-distance' reads it)
    exact: Optional[torch.Tensor] = None    # the scores are fgvc_pair_topk_f16f6's approximations; the bank whose exact rows re-score them
    geom: Optional[Tuple[int, int]] = None  # (Hf, Wf) (the refining merge needs the grid for the mask predicate)


def run_affinity(feats_hwc: torch.Tensor, Hf: int, Wf: int, plan: Plan, cfg: TrackerConfig,
                 pair_chunk: int = 16384, events=None, phases=None, channels: Optional[int] = None) -> DeviceTopk:
    """Phases 1 and 2 (= merge_pairs(run_pairs(...)))."""
    return merge_pairs(run_pairs(feats_hwc, Hf, Wf, plan, cfg, pair_chunk, events, channels=channels, phases=phases), cfg)


def merge_pairs(pl: PairLists, cfg: TrackerConfig, rows: Optional[Sequence[int]] = None) -> DeviceTopk:
    """Phase 2: per query frame, merge the lists of its key slots and softmax the k survivors.
    `rows` = plan rows to merge (default all).  track_points merges one query-time group at a time: with strided TAP-Vid queries
    a clip has one group per distinct query time and merging every (group, frame) row at once costs rows * HW * k * 12 bytes
    (T = 250 at 128 x 128: ~6k rows, 12 GB)."""
    plan = pl.plan
    _, slot_pair, slot_frame = plan.tables(pl.idx.device)
    row_map = None
    if rows is not None:
        rows = list(rows)
        sel = torch.tensor(rows, dtype=torch.int64, device=pl.idx.device)
        slot_pair, slot_frame = slot_pair[sel].contiguous(), slot_frame[sel].contiguous()
        row_map = {r: i for i, r in enumerate(rows)}
    if slot_pair.shape[0] == 0:
        e = torch.empty((0, pl.HW, cfg.topk), device=pl.idx.device)
        return DeviceTopk(plan, e.int(), e, e, slot_frame, row_map)
    if pl.exact is not None:
        pairs_dev = plan.tables(pl.idx.device)[0]
        Hf, Wf = pl.geom
        idx, logit, weight, stats = ops.merge_refine_topk(pl.idx, pl.score, slot_pair, pairs_dev, pl.exact, pl.exact, Hf, Wf, Hf, Wf,
                                                          cfg.mask, cfg.topk, cfg.softmax_temperature(pl.channels), cfg.mode,
                                                          eps=cfg.pair_refine_eps)
        tk = DeviceTopk(plan, idx, logit, weight, slot_frame, row_map)
        tk.refine_stats = stats
        return tk
    idx, logit, weight = ops.merge_topk(pl.idx, pl.score, slot_pair, pl.HW, cfg.topk, cfg.softmax_temperature(pl.channels),
                                        cfg.mode, validate=False)
    return DeviceTopk(plan, idx, logit, weight, slot_frame, row_map)


def run_pairs(feats_hwc: torch.Tensor, Hf: int, Wf: int, plan: Plan, cfg: TrackerConfig,
              pair_chunk: int = 16384, events=None, channels: Optional[int] = None, phases=None) -> PairLists:
    """Phase 1.  feats_hwc (T, HW, C) normalised channels-last features of the whole clip, f32 -- or their
    split form (T, HW, 2, C) int16 in the format cfg.pair_split_fmt names, where the split pair kernel applies
    (VanillaTracker.get_feats_hwc(split=True)).
    `events` = (start, end) torch.cuda.Events recorded around the pair top-k launch(es); `channels` = the encoder's channel
    count where it differs from the (zero-padded) row length.
    `phases` = (first, between): the pairs whose plan indices are in `first` are launched, then `between()` is called, then the
    rest -- clip sharding launches the pairs that touch no halo frame while the halo messages are in flight and lets
    `between` make the stream wait for them (the frames of the later pairs need not be valid in `feats_hwc` before that)."""
    dev = feats_hwc.device
    HW = Hf * Wf
    k = cfg.topk
    n = len(plan.pairs)
    pairs_dev, slot_pair, slot_frame = plan.tables(dev)
    rows = len(plan.slot_pair)
    exact = None
    pre_split = feats_hwc.dtype == torch.int16            # (T, HW, 2 | 4, C): the bank already in a pair kernel's operand format
    if n == 0:                                            # a one-frame clip, or every query point on the last frame: nothing to correlate
        e = torch.empty((0, HW, k), device=dev)
        return PairLists(plan, e.int(), e, HW, feats_hwc.shape[-1] if channels is None else channels)
    all_masked = all(m for (_, _, m) in plan.pairs)
    use_split = cfg.pair_precision == "split" or (
        cfg.pair_precision == "auto" and ops.split_path_ok(feats_hwc.shape[-1], Hf, Wf, k, cfg.with_norm, None, cfg.mask, all_masked))
    if pre_split and not use_split:
        raise ValueError("run_affinity: split features given, but the split pair kernel does not apply to this configuration")
    if cfg.sim_mode == "l2-distance" and channels is None:
        # the softmax divisor of this mode is sqrt(C) / 2 with the encoder's OWN channel count (local_attention.py:327 uses
        # att_channels); the row length may be that count rounded up with zero channels (normalize_to_hwc(pad=True)), so it must be given
        raise ValueError("run_pairs: sim_mode='l2-distance' needs channels= (the encoder's un-padded channel count)")
    if use_split:      # 16-bit matrix pipe on the two-part split of the (normalised) features, f32-grade scores
        if cfg.pair_split_fmt not in ("f16", "f16f6"):
            raise ValueError(f"pair_split_fmt={cfg.pair_split_fmt!r}: 'f16' (split_f16x2 rows) or 'f16f6' (split_f16f6p / split_f16f6x rows)")
        # a pre-split bank says itself whether its rows are split_f16f6x()'s; between split_f16x2() and split_f16f6p() rows (one shape) the
        # configuration's word is all there is -- VanillaTracker.get_feats_hwc builds the bank from the same configuration
        fmt = ops.bank_format(feats_hwc, cfg.pair_split_fmt) if pre_split else cfg.bank_fmt
        if pre_split and fmt == "f16f6x" and cfg.pair_split_fmt != "f16f6":
            raise ValueError("run_pairs: the bank holds split_f16f6x() rows, but the configuration asks for pair_split_fmt='f16' "
                             "(build the bank and run the pairs from ONE configuration: VanillaTracker.engine_config())")
        if fmt != "f16" and not ops.pair_f16f6_ok(feats_hwc.shape[-1], Hf, Wf, k, cfg.with_norm, None, cfg.mask, all_masked):
            if pre_split:
                raise ValueError("run_pairs: the bank is in the f16f6 format, but fgvc_pair_topk_f16f6 does not apply to this mask / pair list "
                                 "(every pair masked, at most 64 key blocks in reach): encode with pair_split_fmt='f16'")
            fmt = "f16"
        split = feats_hwc if pre_split else {"f16": ops.split_f16x2, "f16f6": ops.split_f16f6p, "f16f6x": ops.split_f16f6x}[fmt](feats_hwc)
        if fmt == "f16f6x":
            exact = split                                 # the refining merge reads the rows' second KiB
        pair_fn = lambda prs: ops.pair_topk_split(split, split, prs, Hf, Wf, Hf, Wf, cfg.mask, k, validate=False, all_masked=all_masked,
                                                  fmt=fmt)
    else:
        pair_fn = lambda prs: ops.pair_topk(feats_hwc, feats_hwc, prs, Hf, Wf, Hf, Wf, cfg.mask, k, validate=False)
    if events is not None:
        events[0].record()
    if phases is not None:
        first, between = phases
        fs = set(int(i) for i in first)
        key = ("phases", str(dev), tuple(sorted(fs)))
        if key not in plan._dev:
            sel = [[i for i in range(n) if i in fs], [i for i in range(n) if i not in fs]]
            plan._dev[key] = [(torch.tensor(ix, dtype=torch.int64, device=dev), ops.make_pairs([plan.pairs[i] for i in ix], dev))
                              for ix in sel]
        pidx = torch.empty((n, HW, k), device=dev, dtype=torch.int32)
        pscore = torch.empty((n, HW, k), device=dev, dtype=torch.float32)
        for ph, (ix, prs) in enumerate(plan._dev[key]):
            if ph == 1:
                between()
            for c0 in range(0, prs.shape[0], pair_chunk):
                c1 = min(prs.shape[0], c0 + pair_chunk)
                i, s_ = pair_fn(prs if (c0 == 0 and c1 == prs.shape[0]) else ops.make_pairs([plan.pairs[j] for j in ix[c0:c1].tolist()], dev))
                pidx.index_copy_(0, ix[c0:c1], i)
                pscore.index_copy_(0, ix[c0:c1], s_)
    elif n <= pair_chunk:
        pidx, pscore = pair_fn(pairs_dev)
    else:
        pidx = torch.empty((n, HW, k), device=dev, dtype=torch.int32)
        pscore = torch.empty((n, HW, k), device=dev, dtype=torch.float32)
        for c0 in range(0, n, pair_chunk):
            c1 = min(n, c0 + pair_chunk)
            i, s = pair_fn(pairs_dev[c0:c1])
            pidx[c0:c1], pscore[c0:c1] = i, s
    if events is not None:
        events[1].record()
    return PairLists(plan, pidx, pscore, HW, feats_hwc.shape[-1] if channels is None else channels, exact=exact, geom=(Hf, Wf))


def run_propagation(topk: DeviceTopk, start: int, points_xy: torch.Tensor, Hf: int, Wf: int, h: int, w: int,
                    cfg: TrackerConfig, frames: Optional[Tuple[int, int]] = None,
                    labels: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Phase 3 for the group that starts at `start`.  points_xy (P,2) on the device.
    Returns (labels bank (T, HW, P) with frames < start untouched/zero, coords (T-start, P, 2) f64)."""
    plan = topk.plan
    T = plan.n_frames
    dev = points_xy.device
    P = points_xy.shape[0]
    stride = h // Hf                                            # vanilla_tracker.py:197
    if labels is None:
        labels = torch.zeros((T, Hf * Wf, P), device=dev, dtype=torch.float32)
    lo, hi = frames if frames is not None else (start + 1, T)
    if lo <= start + 1:
        ops.gaussian_labels(points_xy, Hf, Wf, stride, cfg.sigma, out=labels[start])
    for f in range(max(lo, start + 1), hi):
        row = topk.row(plan.out_rows[(start, f)])
        ops.propagate_topk(labels, topk.slot_frame[row], topk.idx[row], topk.weight[row], Hf, Wf, Hf, Wf,
                           out=labels[f])
    coords = ops.softargmax_top5(labels[start:], Hf, Wf, h, w, gauss_points=points_xy, sigma=cfg.sigma)
    return labels, coords


def run_propagation_async(topk, start: int, points_xy: torch.Tensor, Hf: int, Wf: int, h: int, w: int,
                          cfg: TrackerConfig, stream: "torch.cuda.Stream"):
    """run_propagation on a side stream: the sweep and the read-out are a chain of small launches that leave most of the GPU
    idle (0.27 ms per 480p clip); on their own stream the NEXT clip's encoder runs under them.  `topk`: a DeviceTopk, or the
    PairLists of run_pairs() -- then the merge runs on the side stream as well.  The caller's stream does not wait: returns
    (labels, coords, event) -- wait for the event (or synchronise) before reading them.  The inputs are kept from the caching
    allocator until the side stream is done with them (record_stream)."""
    cur = torch.cuda.current_stream(points_xy.device)
    stream.wait_stream(cur)
    with torch.cuda.stream(stream):
        if isinstance(topk, PairLists):
            for t in (topk.idx, topk.score, topk.exact):
                if t is not None:
                    t.record_stream(stream)
            topk = merge_pairs(topk, cfg)
        for t in (topk.idx, topk.logit, topk.weight, topk.slot_frame, points_xy):
            t.record_stream(stream)
        labels, coords = run_propagation(topk, start, points_xy, Hf, Wf, h, w, cfg)
        done = torch.cuda.Event()
        done.record(stream)
    return labels, coords, done


def track_points(feats_hwc: torch.Tensor, Hf: int, Wf: int, h: int, w: int, query_points: torch.Tensor,
                 cfg: TrackerConfig, channels: Optional[int] = None, stats_out: Optional[list] = None):
    """The whole post-encoder path for one clip.  query_points (P,3) = (t, x, y) (any device).
    Returns traj_pred (T, P', 2) f64 on the device with points re-ordered by query time (the
    reference's regrouping, vanilla_tracker.py:257-299) and `order` (P',) original indices.
    `channels`: the encoder's channel count where the rows are zero-padded beyond it (read by sim_mode='l2-distance' only).
    `stats_out`: a list that receives the refining merge's counters (DeviceTopk.refine_stats) of every group."""
    T = feats_hwc.shape[0]
    dev = feats_hwc.device
    qp = query_points.detach().to("cpu")
    times = qp[:, 0].to(torch.int64)
    starts = sorted(set(times.tolist())) if cfg.regroup else [0]
    plan = plan_clip(T, starts, cfg)
    pl = run_pairs(feats_hwc, Hf, Wf, plan, cfg, channels=channels)
    traj = torch.zeros((T, qp.shape[0], 2), device=dev, dtype=torch.float64)
    order = []
    K = 0
    for s in starts:
        sel = (times == s).nonzero().flatten() if cfg.regroup else torch.arange(qp.shape[0])
        pts = qp[sel, 1:].to(dev, torch.float32)
        topk = merge_pairs(pl, cfg, [plan.out_rows[(s, f)] for f in range(s + 1, T)])     # this group's rows only
        if stats_out is not None and topk.refine_stats is not None:
            stats_out.append(topk.refine_stats)                                          # (the refining merge's device counters, per group)
        _, coords = run_propagation(topk, s, pts, Hf, Wf, h, w, cfg)
        traj[s:, K:K + sel.numel()] = coords
        order.extend(sel.tolist())
        K += sel.numel()
    return traj, torch.tensor(order, dtype=torch.int64)
