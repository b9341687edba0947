"""Sharding ONE long video by clip across the GPUs of a node (BASELINE.json north_star).

Why it is legal: top-k indices/weights of query frame f depend only on the features of
{start frame} U {f-p .. f-1} U {f}; only the label sweep is sequential in f, and it is tiny.

Per rank r (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the GPUs,
"gloo" in the CPU tests):
  1. owns a contiguous range of query frames [lo, hi); encodes frames [lo - p, hi) itself
     (p-frame halo recompute: cheaper and simpler than shipping 5 x 26 MB of features);
  2. EXCHANGE STEP 1 -- `broadcast` of each group's first-frame ("query") features from the rank
     that encoded it: C*HW*4 B = 26 MB at 480p/stride 4, fan-out over the 7 xGMI links;
  3. correlation + top-k + merge for its own frames (no communication);
  4. EXCHANGE STEP 2 -- `all_gather` of the merged per-frame lists (idx int32 + weight f32 =
     HW*k*8 B = 2 MB per frame);
  5. every rank runs the (cheap, deterministic) sequential label sweep + read-out, so no final
     broadcast of the coordinates is needed.
There is no all-reduce anywhere; the ring-bound per-link limit of xGMI is irrelevant at these sizes.

The compute is injected through a small backend object so that the choreography (halo ranges,
ownership, gather order) is exercised on CPU with gloo and an oracle-backed backend in
tests/test_dist_gloo.py, and by HipBackend on the GPUs.
"""
from __future__ import annotations

from dataclasses import replace
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import engine
from .engine import Plan, TrackerConfig


def shard_frames(n_frames: int, world: int, first: int = 1) -> List[Tuple[int, int]]:
    """Contiguous, balanced split of the query frames [first, n_frames) over `world` ranks
    (ranks beyond the number of frames get an empty range)."""
    n = max(0, n_frames - first)
    base, rem = divmod(n, world)
    out, lo = [], first
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def encode_range(lo: int, hi: int, starts: Sequence[int], cfg: TrackerConfig) -> Tuple[int, int]:
    """Frames rank must encode itself: its query frames plus the preceding-frame halo, clipped at the
    earliest start (frames before every start are never used)."""
    if hi <= lo:
        return (lo, lo)
    return (max(min(starts), lo - cfg.precede_frames), hi)


def owner_of(frame: int, enc_ranges: List[Tuple[int, int]]) -> int:
    """Lowest rank that encodes `frame` anyway (so a broadcast costs no extra encoder pass)."""
    for r, (a, b) in enumerate(enc_ranges):
        if a <= frame < b:
            return r
    return 0


class HipBackend:
    """The product backend: encoder through the tracker model, kernels through fgvc_amd.engine."""

    def __init__(self, model):
        self.model = model

    def encode(self, frames: torch.Tensor):
        return self.model.get_feats_hwc(frames)

    def affinity(self, bank: torch.Tensor, Hf: int, Wf: int, plan: Plan, cfg: TrackerConfig):
        tk = engine.run_affinity(bank, Hf, Wf, plan, cfg)
        return tk.idx, tk.weight

    def sweep(self, idx, weight, slot_frame, plan: Plan, start: int, pts, Hf, Wf, h, w, cfg):
        tk = engine.DeviceTopk(plan, idx, None, weight, slot_frame)
        return engine.run_propagation(tk, start, pts, Hf, Wf, h, w, cfg)[1]


def track_points_sharded(backend, rgbs: torch.Tensor, query_points: torch.Tensor, cfg: TrackerConfig,
                         group=None, device: Optional[torch.device] = None):
    """One video, all ranks.  rgbs (T,3,h,w) (every rank may hold the whole clip on the host; only
    its own frames are moved/encoded), query_points (P,3)=(t,x,y).
    Returns (traj (T,P',2) f64 regrouped by query time, order (P',)) on every rank."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    T, h, w = rgbs.shape[0], rgbs.shape[-2], rgbs.shape[-1]
    dev = device if device is not None else rgbs.device
    qp = query_points.detach().cpu()
    times = qp[:, 0].to(torch.int64)
    starts = sorted(set(times.tolist())) if cfg.regroup else [0]
    s_min = min(starts)

    ranges = shard_frames(T, world, first=s_min + 1)
    enc = [encode_range(lo, hi, starts, cfg) for lo, hi in ranges]
    lo, hi = ranges[rank]
    e_lo, e_hi = enc[rank]

    # ---- 1. local encode (own frames + halo)
    feats: Dict[int, torch.Tensor] = {}
    Hf = Wf = C = None
    if e_hi > e_lo:
        f, Hf, Wf = backend.encode(rgbs[e_lo:e_hi].to(dev))
        C = f.shape[-1]
        for i in range(e_hi - e_lo):
            feats[e_lo + i] = f[i]
    for s in starts:                       # a start frame nobody's range covers is encoded by its owner
        if owner_of(s, enc) == rank and s not in feats:
            f, Hf, Wf = backend.encode(rgbs[s:s + 1].to(dev))
            C = f.shape[-1]
            feats[s] = f[0]
    # ranks with an empty range still take part in the collectives: learn the shapes from rank 0
    shape = torch.tensor([Hf or 0, Wf or 0, C or 0], dtype=torch.int64, device=dev)
    if world > 1:
        shape0 = shape.clone()
        dist.broadcast(shape0, src=0, group=group)
        Hf, Wf, C = (int(v) for v in shape0.tolist())
    HW = Hf * Wf

    # ---- 2. exchange step 1: broadcast every group's first-frame features from its owner
    for s in starts:
        src = owner_of(s, enc)
        buf = feats[s].contiguous() if s in feats and rank == src else torch.empty((HW, C), device=dev,
                                                                                   dtype=torch.float32)
        if world > 1:
            dist.broadcast(buf, src=src, group=group)
        if s not in feats:
            feats[s] = buf

    # ---- 3. local affinity on a compact local bank
    plan = engine.plan_clip(T, starts, cfg, frame_range=(lo, hi))
    local_ids = sorted(feats)
    remap = {f: i for i, f in enumerate(local_ids)}
    needed = {f for (q, k, _) in plan.pairs for f in (q, k)}
    assert needed <= set(local_ids), f"rank {rank}: frames {sorted(needed - set(local_ids))} missing"
    k = cfg.topk
    if plan.pairs:
        bank = torch.stack([feats[f] for f in local_ids], 0)
        lplan = replace(plan, pairs=[(remap[q], remap[kf], m) for (q, kf, m) in plan.pairs], _dev={})
        idx, weight = backend.affinity(bank, Hf, Wf, lplan, cfg)
    else:
        idx = torch.empty((0, HW, k), device=dev, dtype=torch.int32)
        weight = torch.empty((0, HW, k), device=dev, dtype=torch.float32)

    # ---- 4. exchange step 2: all_gather of the merged lists (padded to the largest shard)
    plans = [engine.plan_clip(T, starts, cfg, frame_range=r) for r in ranges]   # deterministic on every rank
    rows = [len(p.slot_pair) for p in plans]
    if world > 1:
        mx = max(rows)
        pad_i = torch.zeros((mx, HW, k), device=dev, dtype=torch.int32)
        pad_w = torch.zeros((mx, HW, k), device=dev, dtype=torch.float32)
        pad_i[: idx.shape[0]] = idx
        pad_w[: weight.shape[0]] = weight
        all_i = [torch.empty_like(pad_i) for _ in range(world)]
        all_w = [torch.empty_like(pad_w) for _ in range(world)]
        dist.all_gather(all_i, pad_i, group=group)
        dist.all_gather(all_w, pad_w, group=group)
        idx = torch.cat([all_i[r][: rows[r]] for r in range(world)], 0)
        weight = torch.cat([all_w[r][: rows[r]] for r in range(world)], 0)
    # global plan = concatenation of the per-rank plans, rows renumbered in rank order
    out_rows, slot_frame, base = {}, [], 0
    for p in plans:
        for key, r in p.out_rows.items():
            out_rows[key] = base + r
        slot_frame.extend(p.slot_frame)
        base += len(p.slot_pair)
    gplan = Plan(T, starts, [], out_rows, [], slot_frame, plan.t_max)
    slot_frame_dev = torch.tensor(slot_frame, dtype=torch.int32, device=dev).reshape(len(slot_frame), plan.t_max)

    # ---- 5. sequential sweep + read-out, replicated on every rank
    traj = torch.zeros((T, qp.shape[0], 2), device=dev, dtype=torch.float64)
    order, col = [], 0
    for s in starts:
        sel = (times == s).nonzero().flatten() if cfg.regroup else torch.arange(qp.shape[0])
        pts = qp[sel, 1:].to(dev, torch.float32)
        coords = backend.sweep(idx, weight, slot_frame_dev, gplan, s, pts, Hf, Wf, h, w, cfg)
        traj[s:, col:col + sel.numel()] = coords
        order.extend(sel.tolist())
        col += sel.numel()
    return traj, torch.tensor(order, dtype=torch.int64)
