"""Sharding ONE long video by clip across the GPUs of a node (BASELINE.json north_star).

Why it is legal: top-k indices/weights of query frame f depend only on the features of
{start frame} U {f-p .. f-1} U {f}; only the label sweep is sequential in f, and it is tiny.

Per rank r (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the GPUs,
"gloo" in the CPU tests):
  1. owns a contiguous clip of query frames [lo, hi) and encodes exactly those (the first rank also the
     video's first frame);
  2. EXCHANGE STEP 1 -- `broadcast` of each group's first-frame ("query") features from the rank that
     encoded it: 26 MB at 480p / stride 4 in either bank format, fan-out over the 7 xGMI links;
  3. HALO -- the p = precede_frames frames in front of a clip belong to the previous rank(s):
       halo="exchange" (default): one point-to-point message per (source, destination) pair, all posted
         together (`batch_isend_irecv`): p x 26 MB = 131 MB over one xGMI link (~153 GB/s) = 0.9 ms, against
         ~2.5 ms for encoding five more 480p frames;
       halo="recompute": every rank encodes its own halo (no message, 5 more encoder frames per rank);
  4. correlation + top-k + merge for its own frames (no communication).  The halo messages are posted BEFORE it and the
     pairs that touch no halo frame (33 of a later rank's 48 at p = 5 and 8-frame clips) are launched while they are in
     flight; the stream waits for the messages only in front of the remaining pairs;
  5. EXCHANGE STEP 2 -- `all_gather` of the merged per-frame lists (idx int32 + weight f32 =
     HW*k*8 B = 2 MB per frame), on the backend's side stream when it has one: together with the sweep it then runs
     under whatever the caller enqueues next (the next video's encoder);
  6. every rank runs the (cheap, deterministic) sequential label sweep + read-out, so no final
     broadcast of the coordinates is needed.
There is no all-reduce anywhere; the ring-bound per-link limit of xGMI is irrelevant at these sizes.

The compute is injected through a small backend object so that the choreography (halo ranges,
ownership, message plan, gather order) is exercised on CPU with gloo and an oracle-backed backend in
tests/test_dist_gloo.py, and by HipBackend on the GPUs.  The feature bank travels in whatever form the
backend's encoder returns: f32 rows (HW, C) or the (hi, lo) bf16 split (HW, 2, C) int16 the pair kernel reads.
"""
from __future__ import annotations

import os
import time
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
    """halo="recompute": frames a rank encodes itself = its query frames plus the preceding-frame halo, clipped at the
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


def own_ranges(ranges: List[Tuple[int, int]], s_min: int) -> List[Tuple[int, int]]:
    """halo="exchange": every frame from s_min on is encoded by exactly one rank -- its query range, and frame s_min (a query
    frame of nobody) goes to the first rank that has any."""
    out, given = [], False
    for lo, hi in ranges:
        if hi > lo and not given:
            out.append((s_min, hi))
            given = True
        else:
            out.append((lo, hi))
    return out


def halo_messages(ranges: List[Tuple[int, int]], own: List[Tuple[int, int]], s_min: int, p: int) -> List[Tuple[int, int, int, int]]:
    """halo="exchange": the message plan, identical on every rank: (src, dst, frame_lo, frame_hi) = rank `src` sends its
    frames [frame_lo, frame_hi) to `dst`, whose clip starts within p frames behind them."""
    msgs = []
    for dst, (lo, hi) in enumerate(ranges):
        if hi <= lo:
            continue
        need_lo, need_hi = max(s_min, lo - p), own[dst][0]          # frames in front of what dst encodes itself
        for src, (a, b) in enumerate(own):
            x, y = max(need_lo, a), min(need_hi, b)
            if src != dst and y > x:
                msgs.append((src, dst, x, y))
    return msgs


# bytes this rank handed to each exchange step since the last reset (bench.py checks them against the schedule's arithmetic)
# ---- halo="auto": exchange or recompute, from the schedule's own byte and pair counts (round 6) ----------------------------------
LINK_GBPS_ASSUMED = 77.0        # one direction of one xGMI link between two GPUs of a node; replaced by `measure_link_gbps()` where it ran
_LINK_GBPS = {}                 # per process group: what the 16 MB ping measured


def halo_cost_model(h: int, w: int, stride: int = 4, channels: int = 256) -> dict:
    """What the decision needs to know about a frame, scaled from the 480p measurements of DESIGN section 6 (an MI355X: the encoder
    0.36 ms per 480 x 854 frame, the pair kernel 43 us per 120 x 214 pair) by pixel count; `frame_bytes` = what travels of a bank frame
    (the f32 half of a row: 4 bytes per channel and feature pixel).  A backend may carry its own `cost_model(h, w)`."""
    hf, wf = -(-h // stride), -(-w // stride)
    return {"enc_frame_s": 0.36e-3 * (h * w) / (480.0 * 854.0), "pair_s": 43e-6 * (hf * wf) / (120.0 * 214.0),
            "frame_bytes": hf * wf * channels * 4}


def choose_halo(T: int, world: int, starts, cfg: TrackerConfig, cost: dict, link_gbps: float = LINK_GBPS_ASSUMED, early_halo: bool = True) -> dict:
    """A pure function of the schedule (identical on every rank: it fixes the order of the communication calls).  Per boundary between
    two clips: the halo message is `p` frames of `frame_bytes` over one link direction; it is posted before the pair launch (after the
    sender's last p frames with the early halo: the receiver's remaining encoder frames also cover it) and awaited in front of the first
    pair that touches a halo frame, so what it can hide behind is the receiver's halo-free pairs (+ its front frames' encode); what is
    left over is EXPOSED.  Recomputing costs p encoder frames on the receiver and no message.  One mode for the whole video -- the clips
    are equal, so are the boundaries; the worst boundary decides.  Returns the mode and the numbers it was chosen from."""
    s_min = min(starts)
    p = cfg.precede_frames
    ranges = shard_frames(T, world, first=s_min + 1)
    rows = []
    for r, (lo, hi) in enumerate(ranges):
        if r == 0 or hi <= lo:
            continue
        n_halo = min(p, lo - s_min - 1) if lo - 1 > s_min else 0           # frames in front of the clip that another rank encodes (frame s_min comes by broadcast)
        if n_halo <= 0:
            continue
        plan = engine.plan_clip(T, list(starts), cfg, frame_range=(lo, hi))
        halo_frames = set(range(lo - n_halo, lo))
        free = sum(1 for (q, kf, _) in plan.pairs if q not in halo_frames and kf not in halo_frames)
        t_x = n_halo * cost["frame_bytes"] / (link_gbps * 1e9)
        cover = free * cost["pair_s"] + (max(0, (hi - lo) - p) * cost["enc_frame_s"] if early_halo else 0.0)
        rows.append(dict(rank=r, halo_frames=n_halo, message_bytes=n_halo * cost["frame_bytes"], transfer_s=t_x, halo_free_pairs=free,
                         cover_s=cover, exposed_s=max(0.0, t_x - cover), recompute_s=n_halo * cost["enc_frame_s"]))
    if not rows:
        return dict(mode="exchange", link_gbps=link_gbps, boundaries=[])
    worst = max(rows, key=lambda d: d["exposed_s"] - d["recompute_s"])
    return dict(mode="exchange" if worst["exposed_s"] <= worst["recompute_s"] else "recompute", link_gbps=link_gbps, boundaries=rows)


def measure_link_gbps(group=None, device: Optional[torch.device] = None, nbytes: int = 16 << 20, reps: int = 3) -> float:
    """The one constant `choose_halo` assumes, measured once per process group: rank 0 sends `nbytes` to rank 1 (`reps` times after a
    warm-up, the fastest counts) and the figure is broadcast, so that every rank decides from the same number.  World size 1: the
    assumed constant."""
    key = id(group)
    if key in _LINK_GBPS:
        return _LINK_GBPS[key]
    if not dist.is_initialized() or dist.get_world_size(group) < 2:
        return LINK_GBPS_ASSUMED
    import time
    rank = dist.get_rank(group)
    r0, r1 = _global_rank(group, 0), _global_rank(group, 1)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    best = float("inf")
    for i in range(reps + 1):
        dist.barrier(group=group)
        if buf.is_cuda:
            torch.cuda.synchronize(buf.device)
        t0 = time.perf_counter()
        if rank == 0:
            dist.send(buf, r1, group=group)
        elif rank == 1:
            dist.recv(buf, r0, group=group)
            if buf.is_cuda:
                torch.cuda.synchronize(buf.device)
        dt = time.perf_counter() - t0
        if i and rank == 1:
            best = min(best, dt)
    t = torch.tensor([nbytes / best / 1e9 if rank == 1 else 0.0], dtype=torch.float64, device=device)
    dist.broadcast(t, r1, group=group)
    _LINK_GBPS[key] = float(t.item())
    return _LINK_GBPS[key]


COMM_BYTES = {"broadcast": 0, "halo_send": 0, "halo_recv": 0, "all_gather_send": 0}


def reset_comm_bytes():
    for k in COMM_BYTES:
        COMM_BYTES[k] = 0


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo moves host memory only (its CUDA support covers broadcast / all_reduce): device tensors are staged through the host.
    That is how the HIP backend is exercised with two ranks on ONE GPU in tests (RCCL refuses two ranks on a device)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _wire(t: torch.Tensor) -> torch.Tensor:
    """The tensor as the process group moves it: int16 (the split bank) is a dtype neither RCCL nor gloo knows; its bytes
    travel as uint8 (a view: what is received lands in `t`)."""
    return t.view(torch.uint8) if t.dtype == torch.int16 else t


def _is_f16f6x(t: torch.Tensor) -> bool:
    """Rows of ops.split_f16f6x(): 2 KiB per pixel = the pair kernel's f16 + FP6 row AND the f32 channels it was made from.  Only the
    f32 half travels (1 KiB per pixel, what every other bank format moves); the receiver rebuilds the rows with the same kernel the
    sender's encoder epilogue is held to byte for byte (fgvc_split_f16f6x)."""
    return t.is_cuda and t.dtype == torch.int16 and t.dim() >= 3 and t.shape[-2] == 4 and t.shape[-1] == 256


def _pack_f32(t: torch.Tensor) -> torch.Tensor:
    from . import ops
    return ops.f32_of_f16f6x(t).contiguous()


def _unpack_f32(f32: torch.Tensor, into: torch.Tensor) -> None:
    from . import ops
    ops.split_f16f6x(f32.to(into.device), out=into)


def _global_rank(group, r: int) -> int:
    """The schedule counts ranks inside `group` (enumerate(ranges)); torch.distributed's src / dst / peer arguments are GLOBAL ranks."""
    return r if group is None or group is dist.group.WORLD else dist.get_global_rank(group, r)


def _broadcast(buf: torch.Tensor, src: int, group) -> None:
    """`src`: rank within `group`."""
    rows = buf if _is_f16f6x(buf) else None
    if rows is not None:
        buf = _pack_f32(rows)            # (on every rank: the owner's channels, a buffer to receive into elsewhere)
    buf = _wire(buf)
    src_g = _global_rank(group, src)
    COMM_BYTES["broadcast"] += buf.numel() * buf.element_size()
    if _host_staged(buf, group):
        tmp = buf.cpu()
        dist.broadcast(tmp, src=src_g, group=group)
        buf.copy_(tmp)
    else:
        dist.broadcast(buf, src=src_g, group=group)
    if rows is not None and (dist.get_rank(group) != src):
        _unpack_f32(buf, rows)


def _all_gather(outs: List[torch.Tensor], t: torch.Tensor, group) -> None:
    COMM_BYTES["all_gather_send"] += t.numel() * t.element_size()
    if _host_staged(t, group):
        tmp = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(tmp, t.cpu(), group=group)
        for o, c in zip(outs, tmp):
            o.copy_(c)
    else:
        dist.all_gather(outs, t, group=group)


class _Messages:
    """Point-to-point messages posted together (`batch_isend_irecv`), completed by .wait(): on RCCL that makes the CURRENT
    STREAM wait for the transfers (the host does not block), so kernels launched in between overlap them."""

    def __init__(self, group):
        self.group, self.ops, self.keep, self.land, self.reqs, self.split = group, [], [], [], None, []

    def send(self, t: torch.Tensor, dst: int):            # dst / src: ranks within the group
        t = _wire(_pack_f32(t) if _is_f16f6x(t) else t.contiguous())
        COMM_BYTES["halo_send"] += t.numel() * t.element_size()
        if _host_staged(t, self.group):
            t = t.cpu()
        self.keep.append(t)
        self.ops.append(dist.P2POp(dist.isend, t, _global_rank(self.group, dst), self.group))

    def recv(self, into: torch.Tensor, src: int):
        assert into.is_contiguous()
        if _is_f16f6x(into):             # the f32 channels arrive in a buffer of their own; wait() splits them into the bank's rows
            rows, into = into, torch.empty(tuple(into.shape[:-2]) + (256,), dtype=torch.float32, device=into.device)
            self.split.append((rows, into))
        into = _wire(into)
        COMM_BYTES["halo_recv"] += into.numel() * into.element_size()
        if _host_staged(into, self.group):
            tmp = torch.empty(into.shape, dtype=into.dtype)
            self.land.append((into, tmp))
            into = tmp
        self.keep.append(into)
        self.ops.append(dist.P2POp(dist.irecv, into, _global_rank(self.group, src), self.group))

    def post(self):
        """All messages of this rank in ONE batch (one RCCL group: no order inside it can deadlock).  The order of the operations
        inside the batch is canonical all the same -- by peer, and for a peer the lower rank's sends before the higher rank's: both
        ends of every pair of ranks list their common messages in the same sequence, which is what a transport that matches
        point-to-point operations strictly in call order needs (FGVC_P2P_ORDER=posted keeps the order of the send() / recv() calls)."""
        if self.ops:
            ops = self.ops
            if os.environ.get("FGVC_P2P_ORDER", "canonical") != "posted":
                me = dist.get_rank()

                def key(iop):
                    i, op = iop
                    low_to_high = (me < op.peer) if op.op is dist.isend else (op.peer < me)
                    return (op.peer, 0 if low_to_high else 1, i)
                ops = [op for _, op in sorted(enumerate(ops), key=key)]
            self.reqs = dist.batch_isend_irecv(ops)
        return self

    def wait(self):
        for req in self.reqs or []:
            req.wait()
        for into, tmp in self.land:
            into.copy_(tmp)
        for rows, f32 in self.split:
            _unpack_f32(f32, rows)
        self.reqs, self.land, self.keep, self.split = None, [], [], []


class Timing:
    """Optional per-phase timing of track_points_sharded: CUDA events on the current stream when the work is on a GPU (read
    with .report() after a synchronize), wall clock otherwise."""

    def __init__(self, device=None):
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.spans: Dict[str, list] = {}

    def start(self, name):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        else:
            e = time.perf_counter()
        self.spans.setdefault(name, []).append([e, None])

    def stop(self, name):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        else:
            e = time.perf_counter()
        self.spans[name][-1][1] = e

    def report(self) -> Dict[str, float]:
        """Total milliseconds per phase over everything recorded so far."""
        out = {}
        for name, sp in self.spans.items():
            out[name] = sum((a.elapsed_time(b) if self.cuda else (b - a) * 1e3) for a, b in sp if b is not None)
        return out


class HipBackend:
    """The product backend: encoder through the tracker model, kernels through fgvc_amd.engine.
    `tail_stream`: run the replicated sweep + read-out (a chain of small launches that leaves most CUs idle) on this side stream,
    so that whatever the caller enqueues next -- the next video's encoder -- starts under it; `tail_from="pairs"` (the default since
    round 6: +3.7 % frames/s at 480p on one MI355X, A/B/A/B in `profiles/r06_ab_tail_from.log`; it changed nothing in round 3, before the
    encoder's one-wave-per-SIMD kernels left tail rounds for the pair kernel's workgroups to fill) moves the pair top-k, the merge and the
    exchange steps there as well; `"sweep"` keeps them on the caller's stream.  The returned trajectories are then
    produced on that stream: wait for `backend.tail_event` (or synchronise) before reading them."""

    def __init__(self, model, tail_stream: Optional["torch.cuda.Stream"] = None, tail_from: str = "pairs"):
        if tail_from not in ("sweep", "pairs"):
            raise ValueError(f"tail_from={tail_from!r}")
        self.model = model
        self.tail_stream = tail_stream
        self.tail_from = tail_from          # "pairs": everything after the encoder runs on the side stream
        self.tail_event = None
        self.merge_on_tail = True           # the slot merge (pairs() / merge()) joins the sweep on the side stream (when there is one)

    def encode(self, frames: torch.Tensor, out: Optional[torch.Tensor] = None):
        # the bank in the form the pair kernel reads (the (h, l) f16 split where it applies): no second pass, same bytes to ship.
        # `out`: rows of the caller's local bank -- the encoder writes there when shape and dtype match (returned by identity)
        return self.model.get_feats_hwc(frames, split=True, out=out)

    def affinity(self, bank: torch.Tensor, Hf: int, Wf: int, plan: Plan, cfg: TrackerConfig, phases=None):
        """`phases` = (plan indices of the pairs to launch first, callable to run before the others): see engine.run_pairs."""
        tk = engine.run_affinity(bank, Hf, Wf, plan, cfg, phases=phases, channels=getattr(self.model, "feat_channels", None))
        self.refine_stats = tk.refine_stats          # (device counters of the refining merge, None behind an exact pair kernel: bench.py reports them)
        return tk.idx, tk.weight

    def pairs(self, bank: torch.Tensor, Hf: int, Wf: int, plan: Plan, cfg: TrackerConfig, phases=None):
        """affinity() in two steps: the pair top-k here (the caller's stream) ..."""
        return engine.run_pairs(bank, Hf, Wf, plan, cfg, phases=phases, channels=getattr(self.model, "feat_channels", None))

    def merge(self, pl, cfg: TrackerConfig):
        """... and the slot merge (+ the exact re-scoring of near-ties where the pair kernel's scores are approximate) wherever the caller
        puts it: track_points_sharded runs it on the side stream, under the next video's encoder (a chain of latency-bound launches:
        0.2 ms at 480p on the critical path otherwise)."""
        tk = engine.merge_pairs(pl, cfg)
        self.refine_stats = tk.refine_stats
        return tk.idx, tk.weight

    def sweep(self, idx, weight, slot_frame, plan: Plan, start: int, pts, Hf, Wf, h, w, cfg):
        tk = engine.DeviceTopk(plan, idx, None, weight, slot_frame)
        return engine.run_propagation(tk, start, pts, Hf, Wf, h, w, cfg)[1]

    def reset_calibration(self):
        """Called on EVERY rank when any rank's encoder overflowed: drop the per-tensor scales (and what was captured with them).  Under
        the canonical calibration (scales = a function of the weights) the retry of this video runs with 2^4 more headroom per overflow
        in a row -- the same number on every rank, whichever rank overflowed -- and `calibration_ok()` returns to the canonical scales
        after the first call that succeeds; under `calibration = "first_batch"` the next encode calibrates on its own frames."""
        bb = getattr(self.model, "backbone", None)
        cache = getattr(bb, "__dict__", {}).get("_split_cache") if bb is not None else None
        if cache:
            for k in [k for k in cache if isinstance(k, tuple) and k and k[0] == "scales"]:
                del cache[k]
            if hasattr(bb, "_drop_graphs"):
                bb._drop_graphs()
        if getattr(bb, "calibration", None) == "canonical":
            self._overflows_in_a_row = getattr(self, "_overflows_in_a_row", 0) + 1
            bb.__dict__["_headroom_extra"] = min(4 * self._overflows_in_a_row, 12)

    def calibration_ok(self):
        """A video went through without an overflow: the next one starts from the canonical scales again."""
        if getattr(self, "_overflows_in_a_row", 0):
            self._overflows_in_a_row = 0
            bb = getattr(self.model, "backbone", None)
            if hasattr(bb, "end_overflow_retry"):
                bb.end_overflow_retry()

    def failure_flags(self) -> Tuple[bool, bool]:
        """(a bounded wait of the pair kernel's LDS protocol gave up, an activation left the encoder's calibrated f16 range) since the
        last call -- either makes the results since then invalid (poison lists / saturated features).  Reads and clears both device
        flags (a synchronisation); an overflow also drops the encoder's scales (the next encode re-calibrates)."""
        from . import ops
        bb = getattr(self.model, "backbone", None)
        return bool(ops.pair_f16x3_timed_out()), bool(bb is not None and hasattr(bb, "check_overflow") and bb.check_overflow())


def _span(timing: Optional[Timing], name: str):
    class _Ctx:
        def __enter__(self_):
            if timing is not None:
                timing.start(name)

        def __exit__(self_, *exc):
            if timing is not None:
                timing.stop(name)
            return False
    return _Ctx()


def track_points_sharded(backend, rgbs: torch.Tensor, query_points: torch.Tensor, cfg: TrackerConfig,
                         group=None, device: Optional[torch.device] = None, halo: str = "exchange",
                         timing: Optional[Timing] = None, cache: Optional[dict] = None, check: bool = True):
    """One video, all ranks.  rgbs (T,3,h,w) (every rank may hold the whole clip on the host or the device; only
    its own frames are moved/encoded), query_points (P,3)=(t,x,y).
    Returns (traj (T,P',2) f64 regrouped by query time, order (P',)) on every rank.
    `cache`: a dict the caller keeps between calls with the SAME video shape, query points, cfg and process group; the schedule
    (frame ranges, message plan, slot tables on the device, query points on the device, bank geometry) is then built once --
    per call that is a dozen small blocking host-to-device copies and, at more than one rank, one tiny broadcast + host read.
    `check` (default on): before returning, read the backend's failure flags (`backend.failure_flags()`: pair-kernel timeout, encoder
    overflow), agree on them across the group (one MAX all_reduce of two words, so that every rank raises or none does -- a rank
    raising alone would leave the others in their next collective), and if ANY rank overflowed drop the encoder's scales on EVERY
    rank (`backend.reset_calibration()`: the ranks re-calibrate together instead of drifting apart), then raise RuntimeError.  Costs
    a device synchronisation per video; drivers that pipeline videos (bench.py) pass check=False and check once per loop.
    The device flags are process-wide words (read-and-clear): one consumer per process -- two trackers sharing a process also share them."""
    if halo not in ("exchange", "recompute", "auto"):
        raise ValueError(f"halo={halo!r}")
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    T, h, w = rgbs.shape[0], rgbs.shape[-2], rgbs.shape[-1]
    dev = device if device is not None else rgbs.device
    sc = cache.get("schedule") if cache is not None else None
    if sc is None:
        qp = query_points.detach().cpu()
        times = qp[:, 0].to(torch.int64)
        starts = sorted(set(times.tolist())) if cfg.regroup else [0]
        s_min = min(starts)
        ranges = shard_frames(T, world, first=s_min + 1)
        halo_why = None
        if halo == "auto":                      # from the schedule's own byte and pair counts and ONE measured constant: the same on every rank
            cm = getattr(backend, "cost_model", None)
            cost = cm(h, w) if cm is not None else halo_cost_model(h, w)
            halo_why = choose_halo(T, world, starts, cfg, cost, measure_link_gbps(group, dev) if world > 1 else LINK_GBPS_ASSUMED,
                                   early_halo=getattr(backend, "early_halo", True))
            halo = halo_why["mode"]
        if halo == "exchange" and world > 1:
            enc = own_ranges(ranges, s_min)
            msgs = halo_messages(ranges, enc, s_min, cfg.precede_frames)
        else:
            enc = [encode_range(lo, hi, starts, cfg) for lo, hi in ranges]
            msgs = []
        plan = engine.plan_clip(T, starts, cfg, frame_range=ranges[rank])
        plans = [engine.plan_clip(T, starts, cfg, frame_range=r) for r in ranges]   # deterministic on every rank
        # global plan = concatenation of the per-rank plans, rows renumbered in rank order
        out_rows, slot_frame, base = {}, [], 0
        for pp in plans:
            for key, r in pp.out_rows.items():
                out_rows[key] = base + r
            slot_frame.extend(pp.slot_frame)
            base += len(pp.slot_pair)
        groups, order = [], []
        for s in starts:
            sel = (times == s).nonzero().flatten() if cfg.regroup else torch.arange(qp.shape[0])
            groups.append((s, sel.numel(), qp[sel, 1:].to(dev, torch.float32)))
            order.extend(sel.tolist())
        sc = dict(starts=starts, s_min=s_min, ranges=ranges, enc=enc, msgs=msgs, plan=plan, rows=[len(pp.slot_pair) for pp in plans],
                  gplan=Plan(T, starts, [], out_rows, [], slot_frame, plan.t_max),
                  slot_frame_dev=torch.tensor(slot_frame, dtype=torch.int32, device=dev).reshape(len(slot_frame), plan.t_max),
                  groups=groups, order=torch.tensor(order, dtype=torch.int64), n_points=qp.shape[0], lplan=None, geom=None,
                  halo=halo, halo_why=halo_why)
        if cache is not None:
            cache["schedule"] = sc
    starts, ranges, enc, msgs, plan = sc["starts"], sc["ranges"], sc["enc"], sc["msgs"], sc["plan"]
    halo = sc.get("halo", halo)                 # ("auto" was resolved when the schedule was built)
    lo, hi = ranges[rank]
    e_lo, e_hi = enc[rank]

    # ---- 1. local encode
    feats: Dict[int, torch.Tensor] = {}
    Hf = Wf = None
    frame_shape = None
    enc_bank = None
    # With the bank's geometry known from an earlier call (cached schedule), the local bank -- every frame this rank will hold,
    # ascending -- exists BEFORE the encoder runs and the encoder writes its frames into their rows: no copy of the clip's features.
    pre_bank, pre_ids = None, None
    if world > 1 and sc["geom"] is not None and _takes_out(backend):      # (the same decision on every rank: it fixes the order of the collectives)
        g_Hf, g_Wf, g_dtype, g_shape = sc["geom"]
        mine_ = [(src, a, b) for (src, dst, a, b) in msgs if dst == rank]
        own_ = set(range(e_lo, e_hi)) | {s for s in starts if owner_of(s, enc) == rank}
        pre_ids = sorted(own_ | set(starts) | {f for (_, a, b) in mine_ for f in range(a, b)})
        pre_bank = torch.empty((len(pre_ids),) + tuple(g_shape), device=dev, dtype=g_dtype)
    # Early halo (every call after the first of a cached schedule, `halo="exchange"`): a clip's halo is the LAST p frames of the clip
    # before it, so every rank encodes its last p frames FIRST (into the bank), posts its messages -- the sends of those frames and
    # the receives of its own halo, one batch -- and only then encodes the frames in front of them: the transfer (5 x 26 MB over one
    # xGMI direction, ~1.7 ms) runs under the rest of the encoder instead of behind it.  Every rank posts at the same point of its
    # schedule (also the last rank, which only receives) and before the first-frame broadcast: one order of communication calls
    # for the whole group.
    early_halo = pre_bank is not None and halo == "exchange" and bool(msgs) and getattr(backend, "early_halo", True)
    pending = None

    def post_halo():
        nonlocal pending
        ppos = {f: i for i, f in enumerate(pre_ids)}
        pending = _Messages(group)
        for (src, dst, a, b) in msgs:
            if src == rank:
                pending.send(pre_bank[ppos[a]:ppos[a] + (b - a)], dst)
            elif dst == rank:
                pending.recv(pre_bank[ppos[a]:ppos[a] + (b - a)], src)
        with _span(timing, "halo_exchange"):
            pending.post()

    with _span(timing, "encode"):
        if e_hi > e_lo:
            if pre_bank is not None:
                p0 = pre_ids.index(e_lo)
                rows = pre_bank[p0:p0 + (e_hi - e_lo)]
                split_at = max(e_lo, e_hi - cfg.precede_frames) if early_halo else e_lo
                if split_at > e_lo:
                    ft, Hf, Wf = backend.encode(rgbs[split_at:e_hi].to(dev), out=rows[split_at - e_lo:])
                    if ft.data_ptr() != rows[split_at - e_lo].data_ptr():
                        raise RuntimeError("track_points_sharded: the backend did not encode into the local bank (geometry changed "
                                           "under a cached schedule?)")     # (a silent fallback would change the order of collectives on this rank only)
                    post_halo()
                    fh, Hf, Wf = backend.encode(rgbs[e_lo:split_at].to(dev), out=rows[:split_at - e_lo])
                    # (cannot differ from the call above -- same row shape and dtype -- and that one was checked BEFORE anything was
                    # posted: a rank that raised here alone would leave its peers in their messages)
                    if fh.data_ptr() != rows.data_ptr():      # (an explicit raise: `python -O` strips asserts, and the pair kernel would read rows nobody wrote)
                        raise RuntimeError("track_points_sharded: the backend did not encode into the local bank")
                    f = rows
                else:
                    f, Hf, Wf = backend.encode(rgbs[e_lo:e_hi].to(dev), out=rows)
                    if f.data_ptr() != rows.data_ptr():                  # the backend produced another shape / dtype
                        if early_halo:
                            raise RuntimeError("track_points_sharded: the backend did not encode into the local bank")
                        pre_bank = None                                  # the copy path below
                    elif early_halo:
                        post_halo()
            else:
                f, Hf, Wf = backend.encode(rgbs[e_lo:e_hi].to(dev))
            enc_bank = f
            frame_shape, frame_dtype = tuple(f.shape[1:]), f.dtype
            for i in range(e_hi - e_lo):
                feats[e_lo + i] = f[i]
        elif early_halo:
            post_halo()                                                  # a rank with an empty range still takes part
        for s in starts:                       # a start frame nobody's range covers is encoded by its owner
            if owner_of(s, enc) == rank and s not in feats:
                f, Hf, Wf = backend.encode(rgbs[s:s + 1].to(dev))
                frame_shape, frame_dtype = tuple(f.shape[1:]), f.dtype
                feats[s] = f[0]
    # Everything after the encoder can run on the backend's side stream (`HipBackend(tail_from="pairs")`): the caller's next video
    # then encodes UNDER this video's pair top-k (a kernel paced by key-block traffic and vector issue, beside convolutions paced by
    # the matrix pipe and the LDS).  The bank is a fresh allocation per call; record_stream keeps it alive for the side stream.
    tail = getattr(backend, "tail_stream", None)
    early = tail is not None and getattr(backend, "tail_from", "sweep") == "pairs"
    if early:
        tail.wait_stream(torch.cuda.current_stream(dev))
        for t in ([enc_bank] if enc_bank is not None else []) + [t for f_, t in feats.items() if not (e_lo <= f_ < e_hi)]:
            t.record_stream(tail)
    with (torch.cuda.stream(tail) if early else _Null()):
        # ranks with an empty range still take part in the collectives: learn the bank's geometry from rank 0
        if world > 1:
            if sc["geom"] is None:
                meta = torch.zeros(8, dtype=torch.int64, device=dev)
                if rank == 0:
                    vals = [Hf, Wf, 1 if frame_dtype == torch.int16 else 0, len(frame_shape)] + list(frame_shape)
                    meta[:len(vals)] = torch.tensor(vals, dtype=torch.int64)
                _broadcast(meta, 0, group)
                m = meta.tolist()
                sc["geom"] = (int(m[0]), int(m[1]), torch.int16 if m[2] else torch.float32, tuple(int(v) for v in m[4:4 + int(m[3])]))
            Hf, Wf, frame_dtype, frame_shape = sc["geom"]
        HW = Hf * Wf
        k = cfg.topk

        halo_frames = set()
        if world > 1:
            # ---- the local bank: every frame this rank will hold, ascending, in ONE tensor the pair kernel reads; what arrives from
            #      other ranks lands in its slices directly
            mine = [(src, a, b) for (src, dst, a, b) in msgs if dst == rank]
            halo_frames = {f for (_, a, b) in mine for f in range(a, b)} - set(feats)
            local_ids = sorted(set(feats) | set(starts) | halo_frames)
            pos = {f: i for i, f in enumerate(local_ids)}
            if pre_bank is not None and pre_ids == local_ids and tuple(pre_bank.shape[1:]) == tuple(frame_shape) and pre_bank.dtype == frame_dtype:
                bank = pre_bank                                                    # the encoder already wrote this rank's frames into it
                sc["bank_in_place"] = sc.get("bank_in_place", 0) + 1               # (observability: tests / the rehearsal tool read it)
            else:
                bank = torch.empty((len(local_ids),) + tuple(frame_shape), device=dev, dtype=frame_dtype)
                if enc_bank is not None:
                    bank[pos[e_lo]:pos[e_lo] + (e_hi - e_lo)].copy_(enc_bank)      # consecutive frames = consecutive bank rows
            for f, t in feats.items():
                if not (e_lo <= f < e_hi):
                    bank[pos[f]].copy_(t)
            have = set(feats)
            feats = {f: bank[pos[f]] for f in local_ids}

            # ---- 2. exchange step 1: broadcast every group's first-frame features from its owner
            with _span(timing, "broadcast_first_frames"):
                for s_ in starts:
                    _broadcast(bank[pos[s_]], owner_of(s_, enc), group)
                    have.add(s_)

            # ---- 3. halo: the p frames in front of this rank's clip, from the rank(s) that encoded them -- posted here, awaited in
            #      front of the first pair that reads one of them
            if pending is not None:                       # early halo: posted after this rank's last p frames were encoded
                assert bank is pre_bank
                sc["halo_early"] = sc.get("halo_early", 0) + 1
            else:
                with _span(timing, "halo_exchange"):
                    pending = _Messages(group)
                    for (src, dst, a, b) in msgs:
                        if src == rank:
                            pending.send(bank[pos[a]:pos[a] + (b - a)], dst)
                        elif dst == rank:
                            pending.recv(bank[pos[a]:pos[a] + (b - a)], src)
                    pending.post()
        else:
            with _span(timing, "broadcast_first_frames"):
                pass
            with _span(timing, "halo_exchange"):
                pass
            local_ids = sorted(feats)
            bank = None

        # ---- 4. local affinity on the compact local bank
        def halo_landed():
            if pending is not None:
                with _span(timing, "halo_wait"):
                    pending.wait()

        # the merge of the pair lists goes to the side stream with the sweep where the backend offers the two steps
        pl = None
        # (`backend.merge_on_tail`, default on: measured at 480p, N = 1, A/B/A/B on one box -- 4.80 ms per step with the merge on the side
        # stream against 4.86 behind the pair kernel.  Its launches are latency-bound, but their workgroups keep the next video's
        # convolutions off the CUs they sit on: the encode phase grows by 0.23 ms where the affinity phase loses 0.30)
        split_merge = (getattr(backend, "tail_stream", None) is not None and not early and getattr(backend, "merge_on_tail", False)
                       and hasattr(backend, "pairs") and hasattr(backend, "merge"))
        with _span(timing, "affinity"):
            if plan.pairs:
                if sc["lplan"] is None or sc["lplan"][0] != local_ids:
                    remap = {f: i for i, f in enumerate(local_ids)}
                    needed = {f for (q, kk, _) in plan.pairs for f in (q, kk)}
                    assert needed <= set(local_ids), f"rank {rank}: frames {sorted(needed - set(local_ids))} missing"
                    first = [i for i, (q, kf, _) in enumerate(plan.pairs) if q not in halo_frames and kf not in halo_frames]
                    sc["lplan"] = (local_ids, replace(plan, pairs=[(remap[q], remap[kf], m) for (q, kf, m) in plan.pairs], _dev={}), first)
                if bank is None:
                    if enc_bank is not None and local_ids == list(range(e_lo, e_hi)):
                        bank = enc_bank                                    # nothing came from elsewhere: the encoder's own tensor, no copy
                    else:
                        bank = torch.stack([feats[f] for f in local_ids], 0)
                run = backend.pairs if split_merge else backend.affinity
                if pending is not None and _takes_phases(backend):
                    res = run(bank, Hf, Wf, sc["lplan"][1], cfg, phases=(sc["lplan"][2], halo_landed))
                else:
                    halo_landed()
                    res = run(bank, Hf, Wf, sc["lplan"][1], cfg)
                if split_merge:
                    pl, idx, weight = res, None, None
                else:
                    idx, weight = res
            else:
                halo_landed()
                idx = torch.empty((0, HW, k), device=dev, dtype=torch.int32)
                weight = torch.empty((0, HW, k), device=dev, dtype=torch.float32)

        # ---- 5. exchange step 2: all_gather of the merged lists (padded to the largest shard), on the side stream when there is one
        rows = sc["rows"]
        gplan, slot_frame_dev = sc["gplan"], sc["slot_frame_dev"]
        tail = getattr(backend, "tail_stream", None)
        if tail is not None:
            tail.wait_stream(torch.cuda.current_stream(dev))
            # everything the side stream reads that was allocated on another stream: keep it from the caching allocator until the side
            # stream is done (without a `cache` the schedule -- slot table, the groups' query points -- dies when this function returns)
            held = [slot_frame_dev] + [pts for (_, _, pts) in sc["groups"]]
            held += [idx, weight] if pl is None else [t for t in (pl.idx, pl.score, pl.exact) if t is not None]
            for t in held:
                if t.is_cuda:
                    t.record_stream(tail)
        with (torch.cuda.stream(tail) if tail is not None else _Null()):
            if pl is not None:
                with _span(timing, "merge"):
                    idx, weight = backend.merge(pl, cfg)
            with _span(timing, "all_gather_lists"):
                if world > 1:
                    mx = max(rows)
                    pad_i = torch.zeros((mx, HW, k), device=dev, dtype=torch.int32)
                    pad_w = torch.zeros((mx, HW, k), device=dev, dtype=torch.float32)
                    pad_i[: idx.shape[0]] = idx
                    pad_w[: weight.shape[0]] = weight
                    all_i = [torch.empty_like(pad_i) for _ in range(world)]
                    all_w = [torch.empty_like(pad_w) for _ in range(world)]
                    _all_gather(all_i, pad_i, group)
                    _all_gather(all_w, pad_w, group)
                    idx = torch.cat([all_i[r][: rows[r]] for r in range(world)], 0)
                    weight = torch.cat([all_w[r][: rows[r]] for r in range(world)], 0)

        # ---- 6. sequential sweep + read-out, replicated on every rank
        col = 0
        with _span(timing, "sweep_readout"):
            with (torch.cuda.stream(tail) if tail is not None else _Null()):
                traj = torch.zeros((T, sc["n_points"], 2), device=dev, dtype=torch.float64)
                for s, n_sel, pts in sc["groups"]:
                    coords = backend.sweep(idx, weight, slot_frame_dev, gplan, s, pts, Hf, Wf, h, w, cfg)
                    traj[s:, col:col + n_sel] = coords
                    col += n_sel
                if tail is not None:
                    backend.tail_event = torch.cuda.Event()
                    backend.tail_event.record(tail)
        if check and hasattr(backend, "failure_flags"):
            tail_s = getattr(backend, "tail_stream", None)
            if tail_s is not None:
                tail_s.synchronize()
            flags = torch.tensor([float(v) for v in backend.failure_flags()], device=dev if world > 1 and not _host_staged(traj, group) else "cpu")
            if world > 1:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
            t_out, ovf = (bool(v) for v in flags.tolist())
            if ovf and hasattr(backend, "reset_calibration"):
                backend.reset_calibration()                 # on every rank, also those whose own flag was clear
            if not ovf and hasattr(backend, "calibration_ok"):
                backend.calibration_ok()
            if t_out or ovf:
                raise RuntimeError("track_points_sharded: " + " and ".join(
                    m for m, f in (("a bounded wait of the pair kernel timed out on some rank", t_out),
                                   ("an activation left the encoder's calibrated f16 range on some rank (scales dropped: the next "
                                    "call re-calibrates)", ovf)) if f) + "; this video's results are invalid")
        return traj, sc["order"]


def _takes_out(backend) -> bool:
    """Does backend.encode accept `out=` (rows of the caller's bank to write into)?"""
    import inspect
    try:
        return "out" in inspect.signature(backend.encode).parameters
    except (TypeError, ValueError):
        return False


def _takes_phases(backend) -> bool:
    import inspect
    try:
        return "phases" in inspect.signature(backend.affinity).parameters
    except (TypeError, ValueError):
        return False


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
