"""Tensor-level wrappers over the C ABI (include/fgvc_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every kernel is
libfgvc_hip.so.  Inputs must live on a ROCm device; nothing falls back to torch ops.

Layouts: features channels-last (frames, HW, C); labels pixel-major (frames, HW, P);
top-k lists (…, HW, k) in canonical order (score desc, index asc).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import NO_LIMIT, PAIR_MASKED, WEIGHT_COSINE, WEIGHT_RAW, WEIGHT_SOFTMAX  # noqa: F401


@dataclass(frozen=True)
class MaskSpec:
    """keep <=> dy^2+dx^2 <= r2max and |dy| <= ry and |dx| <= rx   (offset = key - query)."""
    r2max: int = NO_LIMIT
    ry: int = NO_LIMIT
    rx: int = NO_LIMIT

    @property
    def is_none(self) -> bool:
        return self.r2max >= NO_LIMIT and self.ry >= NO_LIMIT and self.rx >= NO_LIMIT

    @staticmethod
    def none() -> "MaskSpec":
        return MaskSpec()

    @staticmethod
    def circle(radius: float) -> "MaskSpec":
        """sqrt(dy^2+dx^2) < radius in float32 (affinity_utils.py:98-109, local_attention.py:463-467)."""
        return MaskSpec(r2max=max(_lib.load().fgvc_r2max_for_radius(float(radius)), 0) if radius > 0 else 0)

    @staticmethod
    def square(neighbor_range) -> "MaskSpec":
        """|dy| <= nr_h//2, |dx| <= nr_w//2 (affinity_utils.py:86-96)."""
        nr = (neighbor_range, neighbor_range) if isinstance(neighbor_range, int) else tuple(neighbor_range)
        return MaskSpec(ry=nr[0] // 2, rx=nr[1] // 2)

    @staticmethod
    def from_neighbor_range(neighbor_range, mode: str = "circle") -> "MaskSpec":
        if neighbor_range is None:
            return MaskSpec.none()
        if mode == "circle":
            return MaskSpec.circle(neighbor_range // 2)
        if mode == "square":
            return MaskSpec.square(neighbor_range)
        raise ValueError(mode)


def set_option(name: str, value: int) -> None:
    """fgvc_set_option: process-wide tuning / ablation knobs (include/fgvc_hip.h lists them), e.g. set_option("readout_prune", 0)."""
    _lib.call("fgvc_set_option", name.encode(), int(value))


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the current stream's handle without building a Stream object


def _stream(t: torch.Tensor):
    """hipStream_t of torch's current stream on t's device.  Called once per launch: the small clips are host-bound (32 launches per
    encoder call at 26 us each), and torch.cuda.current_stream() was a fifth of that."""
    if _raw_stream is not None:
        idx = t.device.index
        return C.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.FgvcHipError(f"{name} must be on the GPU (fgvc_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


MFMA_CHANNELS = (32, 64, 128, 256)   # channel counts fgvc_pair_topk_f32 / fgvc_corr_volume_f32 are built for


def padded_channels(c: int) -> int:
    for m in MFMA_CHANNELS:
        if c <= m:
            return m
    raise _lib.FgvcHipError(f"C={c} > {MFMA_CHANNELS[-1]} channels is not supported by the correlation kernels")


def normalize_to_hwc(x: torch.Tensor, normalize: bool = True, pad: bool = False) -> torch.Tensor:
    """(n, C, H, W) or (n, C, HW) f32 NCHW -> (n, HW, C') L2-normalised over C (eps 1e-12).
    pad=True rounds C' up to a kernel-supported channel count with zero channels."""
    x = _chk(x, torch.float32, "x")
    n, Cc = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    Co = padded_channels(Cc) if pad else Cc
    out = torch.empty((n, HW, Co), device=x.device, dtype=torch.float32)
    _lib.call("fgvc_normalize_chw_to_hwc_f32", _ptr(x), _ptr(out), n, Cc, HW, int(normalize), Co, _stream(x))
    return out


def make_pairs(pairs, device, masked=True) -> torch.Tensor:
    """[(query_frame, key_frame[, masked])...] -> int32 (n,4) device tensor."""
    rows = []
    for p in pairs:
        m = masked if len(p) < 3 else p[2]
        rows.append((int(p[0]), int(p[1]), PAIR_MASKED if m else 0, 0))
    t = torch.tensor(rows, dtype=torch.int32, device=device).reshape(-1, 4)
    t._fgvc_rows = rows                      # host copy: pair_runs() groups the pairs without a device read-back
    return t


def pair_runs(pairs: torch.Tensor) -> Optional[torch.Tensor]:
    """Runs of consecutive pairs with one query frame and one mask flag, as fgvc_pair_topk_f16x3_runs takes them: int32 (n_runs, 2)
    = (first pair, count) on the pairs' device, longest runs first.  None when the pairs were not built by make_pairs() (no host
    copy to group by).  Cached on the tensor."""
    rows = getattr(pairs, "_fgvc_rows", None)
    if rows is None or len(rows) != pairs.shape[0]:
        return None
    cached = getattr(pairs, "_fgvc_runs", None)
    if cached is None:
        runs, i = [], 0
        while i < len(rows):
            j = i
            while j + 1 < len(rows) and rows[j + 1][0] == rows[i][0] and rows[j + 1][2] == rows[i][2]:
                j += 1
            runs.append((i, j - i + 1))
            i = j + 1
        runs.sort(key=lambda r: -r[1])
        cached = torch.tensor(runs, dtype=torch.int32, device=pairs.device).reshape(-1, 2)
        pairs._fgvc_runs = cached
    return cached


def pair_topk(qfeat: torch.Tensor, kfeat: torch.Tensor, pairs: torch.Tensor, Hq: int, Wq: int, Hk: int, Wk: int,
              mask: MaskSpec, topk: int, validate: bool = True,
              dense_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Windowed correlation + top-k per (query frame, key frame) pair.
    qfeat (nq, HqWq, C), kfeat (nk, HkWk, C), pairs int32 (n,4).
    Returns idx (n, HqWq, topk) int32 (key pixel, -1 = none), score (n, HqWq, topk) f32 raw dot."""
    qfeat, kfeat = _chk(qfeat, torch.float32, "qfeat"), _chk(kfeat, torch.float32, "kfeat")
    pairs = _chk(pairs, torch.int32, "pairs")
    assert qfeat.shape[1] == Hq * Wq and kfeat.shape[1] == Hk * Wk and qfeat.shape[2] == kfeat.shape[2]
    n = pairs.shape[0]
    if n and validate:
        # host-side operand check before a hand-written kernel indexes with these (costs a sync;
        # callers that built `pairs` on the host from known-good frame numbers pass validate=False)
        lim = pairs[:, :2].amax(0).tolist()
        assert lim[0] < qfeat.shape[0] and lim[1] < kfeat.shape[0] and int(pairs[:, :2].min()) >= 0, "pair out of range"
    if dense_mask is not None:
        dense_mask = _chk(dense_mask.to(torch.bool), torch.bool, "dense_mask")
        assert dense_mask.shape == (Hk * Wk, Hq * Wq) and mask.is_none
    idx = torch.empty((n, Hq * Wq, topk), device=qfeat.device, dtype=torch.int32)
    score = torch.empty((n, Hq * Wq, topk), device=qfeat.device, dtype=torch.float32)
    _lib.call("fgvc_pair_topk_f32", _ptr(qfeat), _ptr(kfeat), _ptr(pairs), n, qfeat.shape[2], Hq, Wq, Hk, Wk,
              mask.r2max, mask.ry, mask.rx, topk, _ptr(dense_mask), _ptr(idx), _ptr(score), _stream(qfeat))
    return idx, score


def pair_topk_split(qsplit: torch.Tensor, ksplit: torch.Tensor, pairs: torch.Tensor, Hq: int, Wq: int, Hk: int,
                    Wk: int, mask: MaskSpec, topk: int, validate: bool = True,
                    all_masked: bool = False, fmt: str = "f16", use_runs: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """pair_topk() on the 16-bit matrix pipe (fgvc_pair_topk_f16x3): qsplit (nq, HqWq, 2, 256), ksplit (nk, HkWk, 2, 256) int16 =
    split_f16x2() of L2-NORMALISED features.  Same outputs as pair_topk().  all_masked=True: the caller built `pairs` with
    PAIR_MASKED on every row (then only the mask's reach, not the whole key grid, must fit the kernel's block list).
    `fmt` names the operand format: "f16" = split_f16x2() rows -> fgvc_pair_topk_f16x3; "f16f6" = split_f16f6p() rows ->
    fgvc_pair_topk_f16f6; "f16f6x" = split_f16f6x() rows (frames, HW, 4, 256) -> fgvc_pair_topk_f16f6x (the same kernel at a row stride
    of 2 KiB).  split_f16x2() and split_f16f6p() rows have ONE shape and dtype: the caller's `fmt` is all that tells them apart."""
    if fmt not in ("f16", "f16f6", "f16f6x"):
        raise ValueError(f"pair_topk_split: fmt={fmt!r} ('f16': split_f16x2 operands, 'f16f6': split_f16f6p operands, 'f16f6x': split_f16f6x operands)")
    if fmt != "f16" and not (all_masked and pair_blocks_reached(mask) <= PAIR_F16F6_MAX_BLOCKS):
        raise ValueError(f"pair_topk_split(fmt={fmt!r}): every pair must be masked (all_masked=True) by an analytic mask that reaches at most "
                         f"{PAIR_F16F6_MAX_BLOCKS} key blocks per query tile; use fmt='f16'")
    sym = {"f16": "fgvc_pair_topk_f16x3", "f16f6": "fgvc_pair_topk_f16f6", "f16f6x": "fgvc_pair_topk_f16f6x"}[fmt]
    qsplit, ksplit = _chk(qsplit, torch.int16, "qsplit"), _chk(ksplit, torch.int16, "ksplit")
    pairs = _chk(pairs, torch.int32, "pairs")
    parts = 4 if fmt == "f16f6x" else 2          # a bank says whether its rows are 2 KiB (f16f6x): a mismatch would be read as garbage, not refused
    assert qsplit.dim() == 4 and ksplit.dim() == 4 and qsplit.shape[2] == parts and ksplit.shape[2] == parts, \
        f"pair_topk_split(fmt={fmt!r}): operands must be (frames, HW, {parts}, 256) int16, got {tuple(qsplit.shape)} / {tuple(ksplit.shape)}"
    assert qsplit.shape[1] == Hq * Wq and ksplit.shape[1] == Hk * Wk and qsplit.shape[3] == ksplit.shape[3]
    n = pairs.shape[0]
    if n and validate:
        lim = pairs[:, :2].amax(0).tolist()
        assert lim[0] < qsplit.shape[0] and lim[1] < ksplit.shape[0] and int(pairs[:, :2].min()) >= 0, "pair out of range"
        assert not all_masked or bool((pairs[:, 2] & PAIR_MASKED).all()), "all_masked=True but a pair is not masked"
    idx = torch.empty((n, Hq * Wq, topk), device=qsplit.device, dtype=torch.int32)
    score = torch.empty((n, Hq * Wq, topk), device=qsplit.device, dtype=torch.float32)
    runs = pair_runs(pairs) if (use_runs and n) else None
    if runs is not None:            # a query frame's pairs in one workgroup (query prologue once, the key-block ring never drains)
        _lib.call(sym + "_runs", _ptr(qsplit), _ptr(ksplit), _ptr(pairs), n, qsplit.shape[3], Hq, Wq, Hk, Wk,
                  mask.r2max, mask.ry, mask.rx, topk, int(all_masked), _ptr(runs), runs.shape[0], _ptr(idx), _ptr(score),
                  _stream(qsplit))
        return idx, score
    _lib.call(sym, _ptr(qsplit), _ptr(ksplit), _ptr(pairs), n,
              qsplit.shape[3], Hq, Wq, Hk, Wk, mask.r2max, mask.ry, mask.rx, topk, int(all_masked), _ptr(idx), _ptr(score),
              _stream(qsplit))
    return idx, score


def pair_f16x3_timed_out() -> bool:
    """True if a wave of an earlier fgvc_pair_topk_f16x3 launch gave up waiting on its key-block ring (a kernel bug: its spins are
    bounded so that it cannot hang the GPU; the workgroup then writes poison lists -- index 0, score +inf, NaN weights after the
    merge -- so the results cannot pass for valid ones).  Read-and-clear; synchronises the device."""
    return _lib.load().fgvc_pair_topk_f16x3_timed_out() != 0


PAIR_LIST_CAP = 4096   # key blocks (4x8 pixels) one query tile may visit in fgvc_pair_topk_f16x3 (csrc/pair_topk_v5.hip)


def pair_blocks_needed(Hk: int, Wk: int, mask: Optional[MaskSpec] = None, all_masked: bool = False) -> int:
    """Key blocks a 8x16-pixel query tile of fgvc_pair_topk_f16x3 may have to list (pair_topk_v5_launch's own check):
    the blocks within the mask's reach when every pair is masked, the whole key grid otherwise."""
    whole = -(-Hk // 4) * -(-Wk // 8)
    if not all_masked or mask is None or mask.is_none:
        return whole
    rr = math.isqrt(mask.r2max) if mask.r2max < NO_LIMIT else NO_LIMIT
    reach_y, reach_x = min(mask.ry, rr, Hk), min(mask.rx, rr, Wk)
    return min(-(-Hk // 4), (7 + 2 * reach_y) // 4 + 2) * min(-(-Wk // 8), (15 + 2 * reach_x) // 8 + 2)


PAIR_F16F6_MAX_BLOCKS = 64   # key blocks one query tile may visit in fgvc_pair_topk_f16f6 (6 bits of a selection key name the block)


def pair_blocks_reached(mask: Optional[MaskSpec]) -> int:
    """Key blocks (4 x 8 pixels) an interior 8 x 16 query tile reaches under `mask` (csrc/pair_topk_v7.hpp: pair_v7_blocks_reached, the
    same arithmetic); a huge number when there is no analytic mask.  Cached per mask; a reach whose bounding box alone holds more than
    4 x 64 blocks is answered without counting (the only question asked of the result is `<= 64`)."""
    if mask is None or mask.is_none:
        return 1 << 30
    return _pair_blocks_reached(int(mask.r2max), int(mask.ry), int(mask.rx))


@functools.lru_cache(maxsize=256)
def _pair_blocks_reached(r2max: int, ry: int, rx: int) -> int:
    mask = MaskSpec(r2max, ry, rx)
    rr = math.isqrt(mask.r2max) if mask.r2max < NO_LIMIT else NO_LIMIT
    reach_y, reach_x = min(mask.ry, rr), min(mask.rx, rr)
    if reach_y > 4096 or reach_x > 4096:
        return 1 << 30
    if (2 * reach_y // 4) * (2 * reach_x // 8) > 4 * PAIR_F16F6_MAX_BLOCKS and mask.r2max >= reach_y * reach_y + reach_x * reach_x:
        return 1 << 30                                  # a rectangle (no disc cutting its corners) far beyond the bound: no loop
    QBH, QBW = 4, 8
    ny, nx = (2 * QBH - 1 + 2 * reach_y) // QBH + 2, (2 * QBW - 1 + 2 * reach_x) // QBW + 2
    TY0, TX0 = (reach_y // QBH + 1) * QBH, (reach_x // QBW + 1) * QBW
    count = 0
    for by in range(ny + reach_y // QBH + 2):
        for bx in range(nx + reach_x // QBW + 2):
            hit = False
            for b in range(4):
                wy0, wx0, ky0, kx0 = TY0 + (b & 1) * QBH, TX0 + (b >> 1) * QBW, by * QBH, bx * QBW
                dy = max(0, ky0 - (wy0 + QBH - 1), wy0 - (ky0 + QBH - 1))
                dx = max(0, kx0 - (wx0 + QBW - 1), wx0 - (kx0 + QBW - 1))
                hit = hit or (dy * dy + dx * dx <= mask.r2max and dy <= mask.ry and dx <= mask.rx)
            count += hit
    return count


def pair_f16f6_ok(C: int, Hk: int, Wk: int, topk: int, normalized: bool, dense_mask=None, mask: Optional[MaskSpec] = None,
                  all_masked: bool = False) -> bool:
    """Whether fgvc_pair_topk_f16f6 applies: what split_path_ok() asks, every pair masked, and a mask reach of at most 64 key blocks."""
    return (split_path_ok(C, Hk, Wk, topk, normalized, dense_mask, mask, all_masked) and all_masked
            and pair_blocks_reached(mask) <= PAIR_F16F6_MAX_BLOCKS)


def split_path_ok(C: int, Hk: int, Wk: int, topk: int, normalized: bool, dense_mask=None, mask: Optional[MaskSpec] = None,
                  all_masked: bool = False) -> bool:
    """Whether fgvc_pair_topk_f16x3 applies: 256 channels, top-k <= 10, analytic mask, L2-normalised rows (its
    fixed-point keys assume |q.k| <= 1) and at most 4096 key blocks per query tile (pair_blocks_needed)."""
    return (normalized and C == 256 and 1 <= topk <= 10 and dense_mask is None
            and pair_blocks_needed(Hk, Wk, mask, all_masked) <= PAIR_LIST_CAP and Hk < 16384 and Wk < 32768)


def pair_topk_auto(qfeat: torch.Tensor, kfeat: torch.Tensor, pairs: torch.Tensor, Hq: int, Wq: int, Hk: int, Wk: int,
                   mask: MaskSpec, topk: int, normalized: bool, precision: str = "auto", validate: bool = True,
                   dense_mask: Optional[torch.Tensor] = None, all_masked: bool = False,
                   split_fmt: str = "f16") -> Tuple[torch.Tensor, torch.Tensor]:
    """pair_topk() with the kernel chosen by `precision`:
      "f32"   fgvc_pair_topk_f32 (f32 MFMA);
      "split" fgvc_pair_topk_f16x3 on split_f16x2() of the features; raises when it does not apply;
      "auto"  "split" where split_path_ok(), else "f32".
    qfeat/kfeat are the f32 channels-last features either way (the split costs one extra pass over them)."""
    if precision not in ("auto", "f32", "split"):
        raise ValueError(f"precision={precision!r}")
    ok = split_path_ok(qfeat.shape[2], Hk, Wk, topk, normalized, dense_mask, mask, all_masked)
    use_split = precision == "split" or (precision == "auto" and ok)
    if not use_split:
        return pair_topk(qfeat, kfeat, pairs, Hq, Wq, Hk, Wk, mask, topk, validate, dense_mask)
    if not ok:
        raise ValueError("the split pair kernels need C == 256, topk <= 10, an analytic mask, normalised features and "
                         f"<= {PAIR_LIST_CAP} key blocks per query tile")
    if split_fmt not in ("f16", "f16f6", "f16f6x"):
        raise ValueError(f"split_fmt={split_fmt!r}")
    if split_fmt != "f16" and not pair_f16f6_ok(qfeat.shape[2], Hk, Wk, topk, normalized, dense_mask, mask, all_masked):
        if precision == "split":
            raise ValueError("fgvc_pair_topk_f16f6 needs every pair masked by an analytic mask within 64 key blocks of a query tile")
        split_fmt = "f16"                         # "auto": the three-product kernel takes what the f16 + FP6 one cannot
    splitter = {"f16": split_f16x2, "f16f6": split_f16f6p, "f16f6x": split_f16f6x}[split_fmt]      # the rows the kernel of `split_fmt` reads
    ks = splitter(kfeat)
    qs = ks if qfeat is kfeat else splitter(qfeat)
    return pair_topk_split(qs, ks, pairs, Hq, Wq, Hk, Wk, mask, topk, validate, all_masked, fmt=split_fmt)


def merge_topk(pair_idx: torch.Tensor, pair_score: torch.Tensor, slot_pair: torch.Tensor, HWk: int, topk: int,
               temperature: float, mode: str = "softmax", validate: bool = True):
    """slot_pair int32 (n_out, T): pair feeding key slot t of output frame f (-1 unused).
    Returns idx (n_out, HWq, k) int32 = slot*HWk + pixel, logit, weight."""
    pair_idx, pair_score = _chk(pair_idx, torch.int32, "pair_idx"), _chk(pair_score, torch.float32, "pair_score")
    slot_pair = _chk(slot_pair, torch.int32, "slot_pair")
    assert pair_idx.shape == pair_score.shape and pair_idx.shape[2] == topk
    n_out, T = slot_pair.shape
    if validate:
        assert int(slot_pair.max()) < pair_idx.shape[0]
    HWq = pair_idx.shape[1]
    wm = {"softmax": WEIGHT_SOFTMAX, "cosine": WEIGHT_COSINE}[mode]
    idx = torch.empty((n_out, HWq, topk), device=pair_idx.device, dtype=torch.int32)
    logit = torch.empty((n_out, HWq, topk), device=pair_idx.device, dtype=torch.float32)
    weight = torch.empty_like(logit)
    _lib.call("fgvc_merge_topk_f32", _ptr(pair_idx), _ptr(pair_score), _ptr(slot_pair), n_out, T, HWq, HWk, topk,
              float(temperature), wm, _ptr(idx), _ptr(logit), _ptr(weight), _stream(pair_idx))
    return idx, logit, weight


def propagate_topk(labels: torch.Tensor, slot_frame: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor,
                   Hq: int, Wq: int, Hk: int, Wk: int, window_L: int = 0,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """labels (n_frames, HkWk, P), slot_frame int32 (T,), idx/weight (HqWq, k) -> (HqWq, P)."""
    labels = _chk(labels, torch.float32, "labels")
    slot_frame = _chk(slot_frame, torch.int32, "slot_frame")
    idx, weight = _chk(idx, torch.int32, "idx"), _chk(weight, torch.float32, "weight")
    P, k = labels.shape[2], idx.shape[-1]
    assert labels.shape[1] == Hk * Wk and idx.shape[-2] == Hq * Wq and idx.shape == weight.shape
    if out is None:
        out = torch.empty((Hq * Wq, P), device=labels.device, dtype=torch.float32)
    else:
        assert out.is_contiguous() and out.shape == (Hq * Wq, P) and out.dtype == torch.float32
    _lib.call("fgvc_propagate_topk_f32", _ptr(labels), _ptr(slot_frame), slot_frame.numel(), _ptr(idx), _ptr(weight),
              Hq, Wq, Hk, Wk, P, k, window_L, _ptr(out), _stream(labels))
    return out


def split_bf16(feat: torch.Tensor) -> torch.Tensor:
    """(…, C) f32 -> (…, 2, C) int16 holding bf16 bit patterns: hi = bf16(x), lo = bf16(x - hi)."""
    feat = _chk(feat, torch.float32, "feat")
    Cc = feat.shape[-1]
    n = feat.numel() // Cc
    out = torch.empty((*feat.shape[:-1], 2, Cc), device=feat.device, dtype=torch.int16)
    _lib.call("fgvc_split_bf16", _ptr(feat), _ptr(out), n, Cc, _stream(feat))
    return out


def split_f16x2(feat: torch.Tensor) -> torch.Tensor:
    """(…, C) f32 L2-normalised rows -> (…, 2, C) int16 holding f16 bit patterns: h = f16(2^14 x), l = f16(2^14 x - h), the
    operand format of fgvc_pair_topk_f16x3."""
    feat = _chk(feat, torch.float32, "feat")
    Cc = feat.shape[-1]
    n = feat.numel() // Cc
    out = torch.empty((*feat.shape[:-1], 2, Cc), device=feat.device, dtype=torch.int16)
    _lib.call("fgvc_split_f16x2", _ptr(feat), _ptr(out), n, Cc, _stream(feat))
    return out


def split_f16f6p(feat: torch.Tensor) -> torch.Tensor:
    """(…, 256) f32 L2-normalised rows -> (…, 2, 256) int16 = the 1 KiB rows of fgvc_split_f16f6p (h = f16(256 x) + FP6 forms of h
    and of the residual with their block scales, laid out for the lanes of fgvc_pair_topk_f16f6; opaque bytes in the shape and dtype
    of split_f16x2()'s output, so that a feature bank travels and is sliced the same way in either format)."""
    feat = _chk(feat, torch.float32, "feat")
    assert feat.shape[-1] == 256, "fgvc_split_f16f6p: 256 channels"
    n = feat.numel() // 256
    out = torch.empty((*feat.shape[:-1], 2, 256), device=feat.device, dtype=torch.int16)
    _lib.call("fgvc_split_f16f6p", _ptr(feat), _ptr(out), n, 256, _stream(feat))
    return out


def split_f16f6x(feat: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(…, 256) f32 L2-normalised rows -> (…, 4, 256) int16 = 2 KiB per pixel: [the row of split_f16f6p() | the 256 f32 channels
    themselves].  The bank of the default configuration (round 5): fgvc_pair_topk_f16f6x multiplies the first KiB, the refining merge
    (merge_refine_topk) re-scores near-ties exactly from the second; one tensor, sliced and sent like any other bank."""
    feat = _chk(feat, torch.float32, "feat")
    assert feat.shape[-1] == 256, "fgvc_split_f16f6x: 256 channels"
    n = feat.numel() // 256
    if out is None:
        out = torch.empty((*feat.shape[:-1], 4, 256), device=feat.device, dtype=torch.int16)
    else:       # rows of the caller's bank (clip sharding: a halo frame arrives as its f32 channels and is split into its bank rows here)
        assert out.dtype == torch.int16 and tuple(out.shape) == (*feat.shape[:-1], 4, 256) and out.is_contiguous() and out.device == feat.device
    _lib.call("fgvc_split_f16f6x", _ptr(feat), _ptr(out), n, 256, _stream(feat))
    return out


def bank_format(bank: torch.Tensor, declared: str = "f16") -> str:
    """What a feature bank holds, from the tensor itself where it says so: f32 rows (…, C) -> "f32"; int16 (…, 4, 256) -> "f16f6x" (no other
    format has four 512-byte parts); int16 (…, 2, C): `declared` -- split_f16x2() and split_f16f6p() rows are both two 512-byte parts and
    cannot be told apart ("f16" | "f16f6" | "bf16": the caller's word, checked for shape only)."""
    if bank.dtype == torch.float32:
        return "f32"
    if bank.dtype == torch.int16 and bank.dim() >= 3 and bank.shape[-2] == 4 and bank.shape[-1] == 256:
        return "f16f6x"
    if bank.dtype == torch.int16 and bank.dim() >= 3 and bank.shape[-2] == 2:
        if declared == "f16f6x":
            raise ValueError("a (…, 2, C) int16 bank cannot be in the f16f6x format (2 KiB rows: (…, 4, 256))")
        return declared
    raise TypeError(f"not a feature bank: {bank.dtype} {tuple(bank.shape)}")


def exact_rows(bank: torch.Tensor):
    """(tensor, byte offset, frame bytes, row bytes) of the EXACT f32 rows inside a bank (frames, HW, …): an f32 bank (frames, HW, 256) or
    the second KiB of split_f16f6x() rows."""
    assert bank.is_contiguous() and bank.dim() in (3, 4)
    if bank.dtype == torch.float32:
        assert bank.shape[-1] == 256
        return bank, 0, bank.shape[1] * 1024, 1024
    assert bank_format(bank) == "f16f6x"
    return bank, 1024, bank.shape[1] * 2048, 2048


def f32_of_f16f6x(bank: torch.Tensor) -> torch.Tensor:
    """The f32 channels of split_f16f6x() rows as a (…, 256) f32 VIEW of the bank's second KiB."""
    assert bank_format(bank) == "f16f6x"
    return bank[..., 2:, :].view(torch.float32).flatten(-2)      # (…, 2, 128) f32 -> (…, 256): the two parts are adjacent in memory


REFINE_EPS = 2e-5   # |fgvc_pair_topk_f16f6 score - exact product| assumed by the refining merge, raw dot-product units (2.9e-4 logit at tau = 0.07);
                    # measured on the fixtures and at 480p: 7e-6 (tests/test_gpu_refine.py holds every launch's scores to half of this)


def merge_refine_workspace(n_out: int, HWq: int, device) -> torch.Tensor:
    nbytes = _lib.load().fgvc_merge_refine_workspace_bytes(int(n_out), int(HWq))
    return torch.empty(((nbytes + 15) // 16) * 4, dtype=torch.int32, device=device)


def merge_refine_topk(pair_idx: torch.Tensor, pair_score: torch.Tensor, slot_pair: torch.Tensor, pairs: torch.Tensor,
                      q_bank: torch.Tensor, k_bank: torch.Tensor, Hq: int, Wq: int, Hk: int, Wk: int, mask: MaskSpec, topk: int,
                      temperature: float, mode: str = "softmax", eps: float = REFINE_EPS, workspace: Optional[torch.Tensor] = None):
    """merge_topk() behind the approximate scores of fgvc_pair_topk_f16f6, index-exact: candidates whose approximate scores are within
    2 eps of a neighbour are re-scored from the exact rows of `q_bank` / `k_bank` (f32 (frames, HW, 256), or split_f16f6x() banks),
    queries whose window the pair lists do not close are recomputed from every candidate under `mask`.
    Returns idx, logit, weight as merge_topk(), and the int32 statistics {work items (queries re-scored + sampled), of them from scratch,
    candidates re-scored, from-scratch queries beyond the scan queue (slow path), f32 bits of the largest |approximate - exact| score
    among the re-scored candidates, the same over the UNBIASED sample (round 6: every listed entry of a pseudo-random 1/64 of the queries
    whose order needed no re-scoring), entries in that sample, queries in it (counted in the first word too)} as a device tensor (8,) (a view of the workspace: read it before the next
    call that uses the same workspace; refine_max_error() / refine_sample_error() decode words 4 and 5)."""
    pair_idx, pair_score = _chk(pair_idx, torch.int32, "pair_idx"), _chk(pair_score, torch.float32, "pair_score")
    slot_pair, pairs = _chk(slot_pair, torch.int32, "slot_pair"), _chk(pairs, torch.int32, "pairs")
    assert pair_idx.shape == pair_score.shape and pair_idx.shape[2] == topk and pair_idx.shape[1] == Hq * Wq and pairs.shape[0] == pair_idx.shape[0]
    n_out, T = slot_pair.shape
    qb, qo, qfb, qrb = exact_rows(q_bank)
    kb, ko, kfb, krb = exact_rows(k_bank)
    assert qb.is_cuda and kb.is_cuda and qb.shape[1] == Hq * Wq and kb.shape[1] == Hk * Wk
    wm = {"softmax": WEIGHT_SOFTMAX, "cosine": WEIGHT_COSINE}[mode]
    dev = pair_idx.device
    idx = torch.empty((n_out, Hq * Wq, topk), device=dev, dtype=torch.int32)
    logit = torch.empty((n_out, Hq * Wq, topk), device=dev, dtype=torch.float32)
    weight = torch.empty_like(logit)
    need = _lib.load().fgvc_merge_refine_workspace_bytes(int(n_out), int(Hq * Wq))
    if workspace is None or workspace.numel() * workspace.element_size() < need or workspace.device != dev:
        workspace = merge_refine_workspace(n_out, Hq * Wq, dev)
    _lib.call("fgvc_merge_refine_topk_f32", _ptr(pair_idx), _ptr(pair_score), _ptr(slot_pair), _ptr(pairs),
              C.c_void_p(qb.data_ptr() + qo), C.c_int64(qfb), qrb, C.c_void_p(kb.data_ptr() + ko), C.c_int64(kfb), krb,
              n_out, T, Hq, Wq, Hk, Wk, 256, topk, float(temperature), wm, float(eps), mask.r2max, mask.ry, mask.rx,
              _ptr(idx), _ptr(logit), _ptr(weight), _ptr(workspace), _stream(pair_idx))
    return idx, logit, weight, workspace.view(torch.int32)[:8]


def refine_max_error(stats: torch.Tensor) -> float:
    """The largest |approximate - exact| score (dot-product units) the refining merge saw among the candidates it re-scored: `eps` as
    measured on that call's data.  Synchronises (reads the device counters)."""
    e = stats[4:6].cpu().view(torch.float32) if stats.numel() >= 6 else stats[4:5].cpu().view(torch.float32)
    return float(e.max())                 # (both samples: the clustered candidates' and the unbiased one's -- the bound must hold for either)


def refine_sample_error(stats: torch.Tensor):
    """(largest |approximate - exact| over the unbiased sample, entries in it): refine_max_error() over candidates the refining merge did
    NOT need to re-score -- every listed entry, inside the window or not, of 1/64 of the queries whose order was proven as it stood."""
    if stats.numel() < 8:
        return None, 0
    c = stats[:8].cpu()
    return float(c[5:6].view(torch.float32)[0]), int(c[6])


def refine_counts(stats: torch.Tensor) -> dict:
    """The refining merge's counters by name (synchronises)."""
    c = stats.cpu().tolist()
    smp_q = c[7] if len(c) >= 8 else 0
    return dict(queries_rescored=c[0] - smp_q, of_them_from_scratch=c[1], candidates_rescored=c[2], beyond_scan_queue=c[3],
                sampled_queries=smp_q, sampled_entries=c[6] if len(c) >= 7 else 0)


def unsplit_f16f6p(split: torch.Tensor) -> torch.Tensor:
    """The h part of split_f16f6p() rows as f32: (…, 2, 256) int16 -> (…, 256) = h / 256 (an 11-bit approximation of the rows: tests)."""
    b = split.contiguous().view(torch.uint8).reshape(*split.shape[:-2], 1024)
    return b[..., :512].contiguous().view(torch.float16).float() * (1.0 / 256.0)


def split_f16f8(feat: torch.Tensor) -> torch.Tensor:
    """(…, C) f32 L2-normalised rows -> (…, 4 C) uint8: per pixel [h = f16(256 x) | h8 = e4m3(h) | l8 = e4m3(256 (256 x - h))],
    the operand format of fgvc_corr_volume_f16f8."""
    feat = _chk(feat, torch.float32, "feat")
    Cc = feat.shape[-1]
    n = feat.numel() // Cc
    out = torch.empty((*feat.shape[:-1], 4 * Cc), device=feat.device, dtype=torch.uint8)
    _lib.call("fgvc_split_f16f8", _ptr(feat), _ptr(out), n, Cc, _stream(feat))
    return out


def split_f16f6(feat: torch.Tensor) -> torch.Tensor:
    """(…, 256) f32 L2-normalised rows -> (…, 1024) uint8: per pixel [h = f16(256 x) | h6 | l6 | scales | pad] with h6 / l6 the
    block-scaled e2m3 forms of h and of its residual (one E8M0 scale per 32 channels; layout: csrc/corr_volume_f6.hip),
    the operand format of fgvc_corr_volume_f16f6."""
    feat = _chk(feat, torch.float32, "feat")
    Cc = feat.shape[-1]
    n = feat.numel() // Cc
    out = torch.empty((*feat.shape[:-1], 4 * Cc), device=feat.device, dtype=torch.uint8)
    _lib.call("fgvc_split_f16f6", _ptr(feat), _ptr(out), n, Cc, _stream(feat))
    return out


def corr_volume(qfeat: torch.Tensor, kfeat: torch.Tensor, temperature: float = 1.0, precision: str = "f32",
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense volume vol[key j][query i] = <k_j, q_i>/temperature, (HWk, HWq) f32.
    precision 'f32': qfeat (HWq,C), kfeat (HWk,C) f32.  'bf16x3' / 'bf16': the split_bf16() forms (HW,2,C).
    'f16f6' / 'f16f8': the split_f16f6() / split_f16f8() forms (HW, 4C) uint8 of L2-normalised rows, C == 256 (parity-grade;
    f16f6 is the fastest)."""
    HWq, HWk, Cc = qfeat.shape[0], kfeat.shape[0], qfeat.shape[-1]
    if out is None:
        out = torch.empty((HWk, HWq), device=qfeat.device, dtype=torch.float32)
    else:
        assert out.is_contiguous() and out.shape == (HWk, HWq) and out.dtype == torch.float32
    if precision == "f32":
        qfeat, kfeat = _chk(qfeat, torch.float32, "qfeat"), _chk(kfeat, torch.float32, "kfeat")
        _lib.call("fgvc_corr_volume_f32", _ptr(qfeat), _ptr(kfeat), Cc, HWq, HWk, float(temperature), _ptr(out),
                  _stream(qfeat))
    elif precision in ("f16f8", "f16f6"):
        qfeat, kfeat = _chk(qfeat, torch.uint8, "qfeat"), _chk(kfeat, torch.uint8, "kfeat")
        assert qfeat.dim() == 2 and qfeat.shape[1] % 4 == 0 and kfeat.shape[1] == qfeat.shape[1]
        _lib.call("fgvc_corr_volume_" + precision, _ptr(qfeat), _ptr(kfeat), qfeat.shape[1] // 4, HWq, HWk, float(temperature), _ptr(out),
                  _stream(qfeat))
    elif precision in ("bf16x3", "bf16"):
        qfeat, kfeat = _chk(qfeat, torch.int16, "qfeat"), _chk(kfeat, torch.int16, "kfeat")
        assert qfeat.dim() == 3 and qfeat.shape[1] == 2
        _lib.call("fgvc_corr_volume_" + precision, _ptr(qfeat), _ptr(kfeat), Cc, HWq, HWk, float(temperature),
                  _ptr(out), _stream(qfeat))
    else:
        raise ValueError(precision)
    return out


def dense_attend(qfeat: torch.Tensor, kfeat: torch.Tensor, labels: torch.Tensor, Hq: int, Wq: int, Hk: int, Wk: int,
                 mask: MaskSpec, temperature: float, mode: str = "softmax", non_mask_len: int = 0,
                 dense_mask: Optional[torch.Tensor] = None, precision: str = "f32") -> torch.Tensor:
    """topk=None branch (local_attention.py:376-383): weights over every unmasked key of every key slot.
    qfeat (HWq, C) f32 rows, kfeat (T, HWk, C), labels (T, HWk, P) -> (HWq, P).  One HWk x HWq volume slab lives at a time
    (fgvc_corr_volume_f32, or _bf16x3 with precision='bf16x3' for C % 64 == 0), streamed once by fgvc_dense_attend_f32.
    mode: 'softmax' | 'cosine' (clamp(min=0)^2) | 'raw' (the affinity itself is the weight: local_square_attention, :38-103)."""
    qfeat, kfeat = _chk(qfeat, torch.float32, "qfeat"), _chk(kfeat, torch.float32, "kfeat")
    labels = _chk(labels, torch.float32, "labels")
    T, HWk, P = labels.shape
    HWq = qfeat.shape[0]
    assert HWq == Hq * Wq and HWk == Hk * Wk and kfeat.shape[:2] == (T, HWk) and 0 <= non_mask_len <= T
    dev = qfeat.device
    wm = {"softmax": WEIGHT_SOFTMAX, "cosine": WEIGHT_COSINE, "raw": WEIGHT_RAW}[mode]
    ns = _lib.load().fgvc_dense_attend_splits(HWq, HWk)
    state = torch.empty((ns, HWq, P + 2), device=dev, dtype=torch.float32)
    vol = torch.empty((HWk, HWq), device=dev, dtype=torch.float32)
    if precision == "bf16x3":
        qs, ks = split_bf16(qfeat), split_bf16(kfeat)
    if dense_mask is not None:
        dense_mask = _chk(dense_mask.to(torch.bool), torch.bool, "dense_mask")
        assert dense_mask.shape == (HWk, HWq) and mask.is_none
    for t in range(T):
        if precision == "bf16x3":
            corr_volume(qs, ks[t], temperature, "bf16x3", out=vol)
        else:
            corr_volume(qfeat, kfeat[t], temperature, "f32", out=vol)
        masked = t >= non_mask_len
        if dense_mask is not None and masked:
            vol.masked_fill_(~dense_mask, float("-inf"))             # a user's arbitrary mask tensor: applied to the slab as given
        _lib.call("fgvc_dense_attend_f32", _ptr(vol), _ptr(labels[t]), Hq, Wq, Hk, Wk, P, int(masked and not mask.is_none),
                  min(mask.r2max, NO_LIMIT), min(mask.ry, NO_LIMIT), min(mask.rx, NO_LIMIT), wm, int(t == 0), _ptr(state), ns,
                  _stream(qfeat))
    out = torch.empty((HWq, P), device=dev, dtype=torch.float32)
    _lib.call("fgvc_dense_attend_finish_f32", _ptr(state), ns, HWq, P, wm, _ptr(out), _stream(qfeat))
    return out


def dense_propagate(aff: torch.Tensor, labels: torch.Tensor, topk: Optional[int] = None) -> torch.Tensor:
    """`propagate` for a GIVEN dense affinity (affinity_utils.py:33-50): aff (HWk, HWq) f32, labels (HWk, P) f32 -> (HWq, P)
    = labels^T-weighted column sums; topk: weights max(aff - k-th largest of the column, 0), normalised by their sum (:36-44).
    One streaming pass over `aff` per 32 label channels (fgvc_dense_propagate_f32) + one for the thresholds (fgvc_dense_kth_f32)."""
    aff, labels = _chk(aff, torch.float32, "aff"), _chk(labels, torch.float32, "labels")
    HWk, HWq = aff.shape
    assert labels.shape[0] == HWk and labels.dim() == 2
    P = labels.shape[1]
    dev = aff.device
    ns = _lib.load().fgvc_dense_attend_splits(HWq, HWk)
    thr = None
    if topk is not None:
        if not 1 <= topk <= min(64, HWk):
            raise _lib.FgvcHipError(f"propagate: topk={topk} outside 1..min(64, HWk)")
        part = torch.empty((ns, HWq, 16 if topk <= 16 else 64), device=dev, dtype=torch.float32)
        thr = torch.empty((HWq,), device=dev, dtype=torch.float32)
        _lib.call("fgvc_dense_kth_f32", _ptr(aff), HWk, HWq, int(topk), _ptr(part), ns, _ptr(thr), _stream(aff))
    out = torch.empty((HWq, P), device=dev, dtype=torch.float32)
    for p0 in range(0, P, 32):
        lab = labels[:, p0:p0 + 32].contiguous()
        pp = lab.shape[1]
        state = torch.empty((ns, HWq, pp + 2), device=dev, dtype=torch.float32)
        o = torch.empty((HWq, pp), device=dev, dtype=torch.float32)
        _lib.call("fgvc_dense_propagate_f32", _ptr(aff), _ptr(lab), HWk, HWq, pp, _ptr(thr), _ptr(state), ns, _ptr(o), _stream(aff))
        out[:, p0:p0 + pp] = o
    return out


def local_corr_topk(qfeat: torch.Tensor, kfeat: torch.Tensor, H: int, W: int, R: int, topk: int,
                    temperature: float, normalized: bool = False, split_fmt: str = "f16", presplit: bool = False):
    """A7: qfeat (1, HW, C), kfeat (K, HW, C) -> idx (HW,k) int32 = slot*(2R+1)^2 + tap, logit, weight.
    normalized=True (rows are L2-normalised) lets C == 256 / k <= 10 run on the 16-bit matrix pipe (fgvc_local_corr_topk_f16x3).
    presplit=True (round 6): qfeat / kfeat ARE split_f16x2() banks (1, HW, 2, 256) / (K, HW, 2, 256) int16 of normalised rows -- a tracker
    splits a frame once when it enters its bank, not once per query frame that correlates with it (at 480 x 854 the split of seven frames
    was 0.96 of the 4.2 ms this call took: tools/bench_cfg3.py)."""
    if presplit:
        qs, ks = _chk(qfeat, torch.int16, "qfeat"), _chk(kfeat, torch.int16, "kfeat")
        assert qs.dim() == 4 and ks.dim() == 4 and qs.shape[2:] == (2, 256) and ks.shape[2:] == (2, 256) and qs.shape[1] == ks.shape[1] == H * W
        if not split_path_ok(256, H, W, topk, True, None, MaskSpec(ry=R, rx=R), True):
            raise ValueError("local_corr_topk(presplit=True): the 16-bit local-window path needs topk <= 10 and a window within its block list")
        K, dev = ks.shape[0], qs.device
        pairs = make_pairs([(0, t) for t in range(K)], dev)
        ws_i = torch.empty((K, H * W, topk), device=dev, dtype=torch.int32)
        ws_s = torch.empty((K, H * W, topk), device=dev, dtype=torch.float32)
        idx = torch.empty((H * W, topk), device=dev, dtype=torch.int32)
        logit = torch.empty((H * W, topk), device=dev, dtype=torch.float32)
        weight = torch.empty_like(logit)
        _lib.call("fgvc_local_corr_topk_f16x3", _ptr(qs), _ptr(ks), _ptr(pairs), K, 256, H, W, R, topk,
                  float(temperature), _ptr(ws_i), _ptr(ws_s), _ptr(idx), _ptr(logit), _ptr(weight), _stream(qs))
        return idx, logit, weight
    qfeat, kfeat = _chk(qfeat, torch.float32, "qfeat"), _chk(kfeat, torch.float32, "kfeat")
    K = kfeat.shape[0]
    dev = qfeat.device
    if split_fmt != "f16":
        raise ValueError(f"split_fmt={split_fmt!r} (only 'f16')")
    if split_path_ok(qfeat.shape[-1], H, W, topk, normalized, None, MaskSpec(ry=R, rx=R), True):
        qs, ks = split_f16x2(qfeat), split_f16x2(kfeat)
        pairs = make_pairs([(0, t) for t in range(K)], dev)
        ws_i = torch.empty((K, H * W, topk), device=dev, dtype=torch.int32)
        ws_s = torch.empty((K, H * W, topk), device=dev, dtype=torch.float32)
        idx = torch.empty((H * W, topk), device=dev, dtype=torch.int32)
        logit = torch.empty((H * W, topk), device=dev, dtype=torch.float32)
        weight = torch.empty_like(logit)
        _lib.call("fgvc_local_corr_topk_f16x3", _ptr(qs), _ptr(ks), _ptr(pairs), K, qfeat.shape[-1], H, W, R, topk,
                  float(temperature), _ptr(ws_i), _ptr(ws_s), _ptr(idx), _ptr(logit), _ptr(weight), _stream(qfeat))
        return idx, logit, weight
    pairs = make_pairs([(0, t) for t in range(K)], dev)
    ws_i = torch.empty((K, H * W, topk), device=dev, dtype=torch.int32)
    ws_s = torch.empty((K, H * W, topk), device=dev, dtype=torch.float32)
    idx = torch.empty((H * W, topk), device=dev, dtype=torch.int32)
    logit = torch.empty((H * W, topk), device=dev, dtype=torch.float32)
    weight = torch.empty_like(logit)
    _lib.call("fgvc_local_corr_topk_f32", _ptr(qfeat), _ptr(kfeat), _ptr(pairs), K, qfeat.shape[-1], H, W, R, topk,
              float(temperature), _ptr(ws_i), _ptr(ws_s), _ptr(idx), _ptr(logit), _ptr(weight), _stream(qfeat))
    return idx, logit, weight


def topk_coord(idx: torch.Tensor, weight: torch.Tensor, H: int, W: int, R: int, scale: int) -> torch.Tensor:
    """A7 get_coord: (HW,k) window lists of ONE key slot -> (HW,2) expected (x,y) image coordinates."""
    idx, weight = _chk(idx, torch.int32, "idx"), _chk(weight, torch.float32, "weight")
    assert idx.shape == weight.shape == (H * W, idx.shape[1])
    out = torch.empty((H * W, 2), device=idx.device, dtype=torch.float32)
    _lib.call("fgvc_topk_coord_f32", _ptr(idx), _ptr(weight), H, W, R, idx.shape[1], scale, _ptr(out), _stream(idx))
    return out


def c2f_refine(coarse_arg: torch.Tensor, qfine: torch.Tensor, kfine: torch.Tensor, vfine: torch.Tensor, H: int,
               W: int, scale: int, Rf: int, topk: int, temperature: float, mode: str = "softmax"):
    """A6 fine stage. coarse_arg int32 (T, HW); qfine (sHsW, Cf); kfine (T, sHsW, Cf); vfine (T, sHsW, P); mode "softmax" or
    "cosine" (clamp(affinity, 0)^2, local_attention.py:858-861)."""
    coarse_arg = _chk(coarse_arg, torch.int32, "coarse_arg")
    qfine, kfine = _chk(qfine, torch.float32, "qfine"), _chk(kfine, torch.float32, "kfine")
    vfine = _chk(vfine, torch.float32, "vfine")
    T, Cf, P = kfine.shape[0], kfine.shape[-1], vfine.shape[-1]
    assert coarse_arg.shape == (T, H * W) and kfine.shape[1] == H * W * scale * scale == vfine.shape[1]
    assert int(coarse_arg.min()) >= 0 and int(coarse_arg.max()) < H * W
    dev = qfine.device
    out = torch.empty((H * W, P), device=dev, dtype=torch.float32)
    idx = torch.empty((H * W, topk), device=dev, dtype=torch.int32)
    logit = torch.empty((H * W, topk), device=dev, dtype=torch.float32)
    _lib.call("fgvc_c2f_refine_mode_f32", _ptr(coarse_arg), _ptr(qfine), _ptr(kfine), _ptr(vfine), T, H, W, scale, Cf, P,
              Rf, topk, float(temperature), {"softmax": WEIGHT_SOFTMAX, "cosine": WEIGHT_COSINE}[mode], _ptr(out), _ptr(idx), _ptr(logit),
              _stream(qfine))
    return out, idx, logit


def bn_act(x: torch.Tensor, bn: "torch.nn.BatchNorm2d", residual: Optional[torch.Tensor] = None, relu: bool = True,
           inplace: bool = True) -> torch.Tensor:
    """Inference BatchNorm2d (+ residual) (+ ReLU) in one pass over a contiguous NCHW f32 tensor."""
    x = _chk(x, torch.float32, "x")
    N, Cc = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if residual is not None:
        residual = _chk(residual, torch.float32, "residual")
        assert residual.shape == x.shape
    out = x if inplace else torch.empty_like(x)
    _lib.call("fgvc_bn_act_f32", _ptr(x), _ptr(residual), _ptr(bn.running_mean), _ptr(bn.running_var),
              _ptr(bn.weight), _ptr(bn.bias), float(bn.eps), int(relu), _ptr(out), N, Cc, HW, _stream(x))
    return out


# ---- encoder convolutions on the bf16 pipe (fgvc_conv_split_f32) ---------------------------------------------------
def conv_pad_dims(H: int, W: int) -> Tuple[int, int]:
    """(Hp, Wp) of the zero-bordered "padded split NHWC" activation buffers for an H x W image."""
    return 8 * (-(-H // 8)) + 2, 32 * (-(-W // 32)) + 8


def _split_pair(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def prepare_conv_split(weight: torch.Tensor, bn: "torch.nn.BatchNorm2d") -> Tuple[torch.Tensor, torch.Tensor]:
    """Fold eval-mode BatchNorm into a conv weight (Cout, Cin, KS, KS) and lay it out for fgvc_conv_split_f32:
    returns (w int16 [KS*KS][Cin/32][Cout][64] = (hi 32 ci | lo 32 ci) bf16 bit patterns, bias f32 [Cout])."""
    Cout, Cin, KS, _ = weight.shape
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
    w = weight.float() * scale.view(-1, 1, 1, 1)
    bias = (bn.bias - bn.running_mean * scale).float().contiguous()
    w = w.permute(2, 3, 1, 0).reshape(KS * KS, Cin // 32, 32, Cout).permute(0, 1, 3, 2).contiguous()   # [tap][chunk][co][32 ci]
    hi, lo = _split_pair(w)
    packed = torch.cat([hi, lo], dim=-1).contiguous().view(torch.int16)
    return packed, bias


# formats of the encoder's split activation / weight tensors (include/fgvc_hip.h: FGVC_ACT_*) and the fixed power-of-two scales of the
# e4m3 parts of the F16F8 form (csrc/common.hpp: F8_AX ...): h8 = e4m3(h 2^-A), l8 = e4m3(l 2^B)
ACT_BF16X2, ACT_F16F8, ACT_F16X2, ACT_F16F6 = 0, 1, 2, 3
ACT_FMT = {"bf16x3": ACT_BF16X2, "f16f8": ACT_F16F8, "f16x3": ACT_F16X2, "f16f6": ACT_F16F6}
F8_AX, F8_BX, F8_AW, F8_BW = 7, 3, 2, 9
F16_TARGET_LOG2 = 8            # a calibrated tensor's largest value sits at ~2^8 of the f16 range (top 2^16): 2^7-2^8 of headroom


def act_scale_log2(amax: float, target_log2: int = F16_TARGET_LOG2) -> int:
    """log2 of the power-of-two scale that puts a tensor whose largest magnitude is `amax` at (2^(target-1), 2^target] in f16
    (default (2^7, 2^8]; ResNet's canonical calibration asks for 2^6: 2^10 of headroom)."""
    return _pow2_exponent(amax, target_log2)


def _pow2_exponent(amax: float, target_log2: int) -> int:
    """e with amax * 2^e in (2^(target-1), 2^target]; 0 for an all-zero tensor (a zero-initialised residual branch: any scale
    represents it exactly); clamped to +-45 so that the combined exponents stay inside the C ABI's range; a tensor holding inf /
    NaN is an error, not a scale."""
    amax = float(amax)
    if not math.isfinite(amax):
        raise ValueError("a tensor with inf / NaN values has no f16 scale")
    if amax <= 0.0:
        return 0
    return max(-45, min(45, target_log2 - int(math.ceil(math.log2(amax)))))


def _e2m3_blocks(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(..., 32) f32 holding f16 values -> (24 bytes per block (..., 24) uint8, field (...,) int32): the block's FP6 (e2m3) string as
    v_cvt_scalef32_pk32_fp6_f16 writes it (element e in bits [6 e, 6 e + 6), round to nearest even) under the scale 2^(field - 127) =
    the smallest power of two that keeps the block's largest magnitude at or below 7.5 (csrc/common.hpp: split_f16f6_chunk)."""
    m = x.abs().amax(-1).float().contiguous()
    mb = m.view(torch.int32)
    field = ((mb >> 23) - 2 + ((mb & 0x7FFFFF) > 0x700000).to(torch.int32)).clamp(min=32)
    y = x.float() * torch.exp2((127 - field).float()).unsqueeze(-1)              # exact: a power of two
    a = y.abs().clamp(max=7.5)
    step = torch.where(a < 2, 0.125, torch.where(a < 4, 0.25, 0.5))
    r = (torch.round(a / step) * step).clamp(max=7.5)                            # torch.round: half to even
    code = torch.where(r < 2, 8 * r, torch.where(r < 4, 8 + 4 * r, 16 + 2 * r)).to(torch.int32) | ((y < 0).to(torch.int32) << 5)
    c = code.reshape(*code.shape[:-1], 8, 4)
    w24 = c[..., 0] | (c[..., 1] << 6) | (c[..., 2] << 12) | (c[..., 3] << 18)
    by = torch.stack([w24 & 255, (w24 >> 8) & 255, (w24 >> 16) & 255], dim=-1).reshape(*code.shape[:-1], 24).to(torch.uint8)
    return by, field


def _e2m3_decode(by: torch.Tensor, field: torch.Tensor) -> torch.Tensor:
    """The inverse of _e2m3_blocks on its grid: (..., 24) uint8, (...,) scale fields -> (..., 32) f32."""
    b = by.to(torch.int32).reshape(*by.shape[:-1], 8, 3)
    w24 = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    code = torch.stack([(w24 >> (6 * i)) & 63 for i in range(4)], dim=-1).reshape(*by.shape[:-1], 32)
    e, mnt = (code >> 3) & 3, (code & 7).float()
    mag = torch.where(e == 0, mnt * 0.125, (1.0 + mnt * 0.125) * torch.exp2((e - 1).float()))
    val = torch.where((code & 32) != 0, -mag, mag)
    return val * torch.exp2((field.to(torch.int32) - 127).float()).unsqueeze(-1)


def _f16f6_slots(h: torch.Tensor, l: torch.Tensor, first: str) -> torch.Tensor:
    """h (..., 32) f16, l (..., 32) f32 = the exact residuals -> the 128-byte rows (..., 128) uint8 of format ACT_F16F6:
    [h 64 B | slot 4, 5: the main 16 bytes of the two FP6 blocks | slot 6, 7: their 8-byte tails + scale byte + zeros]; `first` names the
    block in slots 4 / 6 ("l": activations, "h": weights)."""
    h6, fh = _e2m3_blocks(h.float())
    l6, fl = _e2m3_blocks((l * 2048.0).to(torch.float16).float())
    fl = fl - 11

    def tail(b6, f):
        t = torch.zeros((*b6.shape[:-1], 16), dtype=torch.uint8, device=b6.device)
        t[..., :8] = b6[..., 16:]
        t[..., 8] = f.to(torch.uint8)
        return t
    A, B = ((l6, fl), (h6, fh)) if first == "l" else ((h6, fh), (l6, fl))
    return torch.cat([h.contiguous().view(torch.uint8).reshape(*h.shape[:-1], 64), A[0][..., :16], B[0][..., :16], tail(*A), tail(*B)], dim=-1)


def prepare_conv_split_f16(weight: torch.Tensor, bn: "torch.nn.BatchNorm2d", fmt: int, force_exp: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """prepare_conv_split for the f16 operand forms: returns (w int16 [KS*KS][Cin/32][Cout][64] -- 128-byte rows in format `fmt` --,
    bias f32 [Cout], log2 of the weights' scale s_w).  ACT_F16F8 rows: [h = f16(s_w w) 32 | h8 = e4m3(h / 4) 32 B | l8 = e4m3(512 (s_w w - h))
    32 B]; ACT_F16X2 rows: [h 32 | l = f16(s_w w - h) 32]; ACT_F16F6 rows: [h 32 | the block-scaled FP6 forms of h (slots 4, 6) and of
    the residual (slots 5, 7): csrc/common.hpp]; s_w = the power of two that puts max|w| at (2^9, 2^10], or 2^force_exp."""
    assert fmt in (ACT_F16F8, ACT_F16X2, ACT_F16F6)
    Cout, Cin, KS, _ = weight.shape
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
    w = weight.detach().float() * scale.view(-1, 1, 1, 1)
    bias = (bn.bias - bn.running_mean * scale).detach().float().contiguous()
    w = w.permute(2, 3, 1, 0).reshape(KS * KS, Cin // 32, 32, Cout).permute(0, 1, 3, 2).contiguous()   # [tap][chunk][co][32 ci]
    e = _pow2_exponent(float(w.abs().max()), 10)
    if force_exp is not None:         # (a projection folded into another convolution's sums shares that convolution's combined scale)
        e = int(force_exp)
        assert float(w.abs().max()) * 2.0 ** e < 32768.0, "prepare_conv_split_f16: the forced scale leaves the f16 range"
    ws = w * (2.0 ** e)
    h = ws.to(torch.float16)
    l = ws - h.float()                                                  # exact in f32
    if fmt == ACT_F16X2:
        packed = torch.cat([h, l.to(torch.float16)], dim=-1).contiguous().view(torch.int16)
        return packed, bias, e
    if fmt == ACT_F16F6:
        return _f16f6_slots(h, l, "h").contiguous().view(torch.int16), bias, e
    h8 = (h.float() * 2.0 ** -F8_AW).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    l8 = (l * 2.0 ** F8_BW).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    packed = torch.cat([h.contiguous().view(torch.uint8), h8.view(torch.uint8), l8.view(torch.uint8)], dim=-1).contiguous()   # 64 + 32 + 32 B
    return packed.view(torch.int16), bias, e


def prepare_conv64(weight: torch.Tensor, bn: "torch.nn.BatchNorm2d") -> Tuple[torch.Tensor, torch.Tensor]:
    """Folded weights (64, 64, 3, 3) for fgvc_conv64_split_f32: prepare_conv_split's values in MFMA-operand order
    [2 output tiles][9 taps][2 chunks][2 k-steps][hi | lo][lane][8]; returns (w int16, bias f32 [64])."""
    assert tuple(weight.shape) == (64, 64, 3, 3)
    packed, bias = prepare_conv_split(weight, bn)                      # [tap 9][chunk 2][Cout 64][part 2 x 32 ci]
    v = packed.view(9, 2, 2, 32, 2, 2, 2, 8)                           # [t][c][ct][n][part][s][h][j]
    v = v.permute(2, 0, 1, 5, 4, 6, 3, 7).contiguous()                 # [ct][t][c][s][part][h][n][j]
    return v.view(2, 9, 2, 2, 2, 64, 8), bias


def prepare_conv64_f16(weight: torch.Tensor, bn: "torch.nn.BatchNorm2d") -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Folded weights (64, 64, 3, 3) for fgvc_conv64_split_fmt_f32 with in_fmt = ACT_F16F8: prepare_conv_split_f16's rows
    ([h 64 B | h8 32 B | l8 32 B] per output channel and 32-channel chunk) in MFMA-operand order, per (output tile, tap, chunk) 4 KiB:
    [f16 k-step 0][64 lanes][16 B], [f16 k-step 1][64 lanes][16 B], then [64 lanes][h8 16 B | l8 16 B] (lane = 32 * half + cout % 32;
    a half's 16 bytes of an fp8 vector are its bytes 16 half .. 16 half + 15).  Returns (w int16 (2, 9, 2, 2048), bias, log2 s_w)."""
    assert tuple(weight.shape) == (64, 64, 3, 3)
    packed, bias, e = prepare_conv_split_f16(weight, bn, ACT_F16F8)      # int16 [tap 9][chunk 2][Cout 64][64]
    rows = packed.view(torch.uint8).view(9, 2, 2, 32, 128)               # [t][c][ct][n][128 B]
    hpart = rows[..., :64].reshape(9, 2, 2, 32, 2, 2, 16)                # [t][c][ct][n][s][half][16 B]
    f16 = hpart.permute(2, 0, 1, 4, 5, 3, 6)                             # [ct][t][c][s][half][n][16]
    h8 = rows[..., 64:96].reshape(9, 2, 2, 32, 2, 16)                    # [t][c][ct][n][half][16]
    l8 = rows[..., 96:128].reshape(9, 2, 2, 32, 2, 16)
    x = torch.stack([h8, l8], dim=-2).permute(2, 0, 1, 4, 3, 5, 6)       # [ct][t][c][half][n][h8 | l8][16]
    out = torch.cat([f16.reshape(2, 9, 2, 2048), x.reshape(2, 9, 2, 2048)], dim=-1).contiguous()   # 2 KiB of f16 fragments + 2 KiB of fp8
    return out.view(torch.int16), bias, e


def prepare_conv_s2(weight: torch.Tensor, bn: "torch.nn.BatchNorm2d") -> Tuple[torch.Tensor, torch.Tensor]:
    """Folded weights for fgvc_conv_s2_split_f32: prepare_conv_split's values in MFMA-operand order
    [KS*KS][Cin/32][Cout/32][hi k0-15 | hi k16-31 | lo k0-15 | lo k16-31][lane = 32 * (k >> 3 & 1) + cout % 32][k & 7]
    (one contiguous KiB per operand); returns (w int16, bias f32 [Cout])."""
    packed, bias = prepare_conv_split(weight, bn)                      # [tap][chunk][Cout][part 2 x 32 ci]
    T, nch, Cout, _ = packed.shape
    assert Cout % 32 == 0
    v = packed.view(T, nch, Cout // 32, 32, 2, 2, 2, 8)                # [t][c][ct][n][part][s][h][j]
    v = v.permute(0, 1, 2, 4, 5, 6, 3, 7).contiguous()                 # [t][c][ct][part][s][h][n][j]
    return v.view(T, nch, Cout // 32, 4, 64, 8), bias


def prepare_stem7(weight: torch.Tensor, bn: "torch.nn.BatchNorm2d") -> Tuple[torch.Tensor, torch.Tensor]:
    """Folded weights (64, 3, 7, 7) for fgvc_stem7_split_f32: per kernel row ky a K block of 32 with k = 4 kx + c (c = 3
    and kx = 7 are zero), as (hi, lo) bf16 in MFMA-operand order [7][2 k-steps][2 output tiles][hi | lo][lane][8];
    returns (w int16, bias f32 [64])."""
    Cout, Cin, KS, _ = weight.shape
    assert (Cout, Cin, KS) == (64, 3, 7)
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
    w = weight.float() * scale.view(-1, 1, 1, 1)
    bias = (bn.bias - bn.running_mean * scale).float().contiguous()
    k = torch.zeros((Cout, 7, 8, 4), device=w.device, dtype=torch.float32)        # [co][ky][kx (8)][c (4)]
    k[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    k = k.reshape(2, 32, 7, 2, 2, 8)                                                # [ct][n][ky][s][h][j]
    k = k.permute(2, 3, 0, 4, 1, 5).contiguous()                                    # [ky][s][ct][h][n][j]
    hi, lo = _split_pair(k)
    packed = torch.stack([hi, lo], dim=3).contiguous()                              # [ky][s][ct][part][h][n][j]
    return packed.view(torch.int16).view(7, 2, 2, 2, 64, 8), bias


def alloc_split_nhwc(N: int, C: int, H: int, W: int, device) -> torch.Tensor:
    Hp, Wp = conv_pad_dims(H, W)
    return torch.zeros((N, Hp, Wp, C // 32, 64), device=device, dtype=torch.int16)


def alloc_nhwc(N: int, C: int, H: int, W: int, device) -> torch.Tensor:
    """Dense NHWC f32 (N,H,W,C): residual / f32 output of conv_split; `.permute(0,3,1,2)` is a channels_last NCHW tensor."""
    return torch.empty((N, H, W, C), device=device, dtype=torch.float32)


def nhwc_to_split(x: torch.Tensor, out: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """dense NHWC f32 (N,H,W,C) -> padded split NHWC `out` (zero border); relu=True applies ReLU to x IN PLACE first."""
    x = _chk(x, torch.float32, "x")
    N, H, W, C = x.shape
    Hp, Wp = conv_pad_dims(H, W)
    assert out.shape == (N, Hp, Wp, C // 32, 64) and out.dtype == torch.int16 and out.is_contiguous()
    _lib.call("fgvc_nhwc_to_split_f32", _ptr(x), _ptr(out), N, C, H, W, Hp, Wp, int(relu), _stream(x))
    return out


def nchw_to_split_nhwc(x: torch.Tensor, out: Optional[torch.Tensor] = None, out_f32: Optional[torch.Tensor] = None,
                       want_split: bool = True) -> Optional[torch.Tensor]:
    """f32 (N,C,H,W) -> padded split NHWC (N,Hp,Wp,C/32,64) int16 [and/or dense NHWC f32 `out_f32` (N,H,W,C)];
    the split destination must have a zero border (alloc_split_nhwc)."""
    x = _chk(x, torch.float32, "x")
    N, C, H, W = x.shape
    if out is None and want_split:
        out = alloc_split_nhwc(N, C, H, W, x.device)
    Hp, Wp = conv_pad_dims(H, W)
    if out is not None:
        assert out.shape == (N, Hp, Wp, C // 32, 64) and out.dtype == torch.int16 and out.is_contiguous()
    if out_f32 is not None:
        assert out_f32.shape == (N, H, W, C) and out_f32.dtype == torch.float32 and out_f32.is_contiguous()
    _lib.call("fgvc_nchw_to_split_nhwc_f32", _ptr(x), _ptr(out), _ptr(out_f32), N, C, H, W, Hp, Wp, _stream(x))
    return out


def conv_split(x_split: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, H: int, W: int, relu: bool,
               residual: Optional[torch.Tensor] = None, out_split: Optional[torch.Tensor] = None,
               out_f32: Optional[torch.Tensor] = None, in_fmt: int = ACT_BF16X2, in_scale_log2: int = 0,
               out_fmt: int = ACT_BF16X2, out_scale_log2: int = 0, overflow: Optional[torch.Tensor] = None,
               x2_split: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None) -> None:
    """fgvc_conv_split_fmt_f32: y = conv(x, w) + bias [+ residual] [ReLU] into out_split and/or out_f32 (interiors only).
    With `x2_split` / `w2` (fgvc_conv_split_proj_fmt_f32; 3 x 3, 256 output channels): + conv1x1(x2, w2) in the same sums -- a block's
    projection shortcut; x2 in x's format and geometry, w2 (1, Cin2/32, 256, 64) scaled so that s_x2 s_w2 = 2^in_scale_log2 too, `bias`
    the sum of both folded biases.
    in_fmt: ACT_* format of x_split AND w (prepare_conv_split / prepare_conv_split_f16), in_scale_log2 = log2(s_x s_w);
    out_fmt / out_scale_log2: format and scale of out_split; overflow: int32 device word OR-ed with 1 when an f16-format output
    leaves the f16 range."""
    x_split, w = _chk(x_split, torch.int16, "x_split"), _chk(w, torch.int16, "w")
    bias = _chk(bias, torch.float32, "bias")
    N, Hp, Wp, nch, _ = x_split.shape
    taps, nch_w, Cout, _ = w.shape
    assert nch_w == nch and taps in (1, 9) and bias.shape == (Cout,)
    for t, dt, shape in ((residual, torch.float32, (N, H, W, Cout)), (out_f32, torch.float32, (N, H, W, Cout)),
                         (out_split, torch.int16, (N, Hp, Wp, Cout // 32, 64))):
        if t is not None:
            assert t.dtype == dt and tuple(t.shape) == shape and t.is_contiguous() and t.device == x_split.device, "conv_split buffer"
    if out_fmt != ACT_BF16X2 and out_split is not None:
        assert overflow is not None and overflow.dtype == torch.int32 and overflow.device == x_split.device
    if x2_split is not None:
        x2_split, w2 = _chk(x2_split, torch.int16, "x2_split"), _chk(w2, torch.int16, "w2")
        assert taps == 9 and Cout == 256, "conv_split: a second input goes with the 3 x 3 form of 256 output channels"
        assert tuple(x2_split.shape[:3]) == (N, Hp, Wp) and x2_split.shape[4] == 64 and tuple(w2.shape) == (1, x2_split.shape[3], Cout, 64)
        _lib.call("fgvc_conv_split_proj_fmt_f32", _ptr(x_split), _ptr(w), _ptr(x2_split), _ptr(w2), _ptr(bias), _ptr(residual), _ptr(out_split),
                  _ptr(out_f32), N, H, W, Hp, Wp, nch * 32, x2_split.shape[3] * 32, int(relu), int(in_fmt), int(in_scale_log2), int(out_fmt),
                  int(out_scale_log2), _ptr(overflow), _stream(x_split))
        return
    _lib.call("fgvc_conv_split_fmt_f32", _ptr(x_split), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out_split), _ptr(out_f32),
              N, H, W, Hp, Wp, nch * 32, Cout, 3 if taps == 9 else 1, int(relu), int(in_fmt), int(in_scale_log2), int(out_fmt),
              int(out_scale_log2), _ptr(overflow), _stream(x_split))


def conv_split_to_bank(x_split: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, H: int, W: int, relu: bool, bank: torch.Tensor,
                       residual: Optional[torch.Tensor] = None, in_fmt: int = ACT_BF16X2, in_scale_log2: int = 0,
                       normalize: bool = True) -> torch.Tensor:
    """fgvc_conv_split_bank_f16f6p_f32: conv_split() for Cout = 256 whose epilogue L2-normalises every pixel and writes it as a row of
    split_f16f6p() -- `bank` (N, H*W, 2, 256) int16 (a slice of the tracker's feature bank) -- instead of any dense or split tensor:
    byte for byte normalize_nhwc(out_f32 of conv_split(...), split="f16f6")."""
    x_split, w = _chk(x_split, torch.int16, "x_split"), _chk(w, torch.int16, "w")
    bias = _chk(bias, torch.float32, "bias")
    N, Hp, Wp, nch, _ = x_split.shape
    taps, nch_w, Cout, _ = w.shape
    assert nch_w == nch and taps == 9 and bias.shape == (Cout,) and Cout == 256, "conv_split_to_bank: a 3 x 3 convolution with 256 output channels"
    assert bank.dtype == torch.int16 and tuple(bank.shape) in ((N, H * W, 2, 256), (N, H * W, 4, 256)) and bank.is_contiguous() and bank.device == x_split.device
    if residual is not None:
        assert residual.dtype == torch.float32 and tuple(residual.shape) == (N, H, W, Cout) and residual.is_contiguous()
    # (N, H*W, 4, 256): the 2 KiB rows of split_f16f6x() -- the normalised f32 channels behind every row
    _lib.call("fgvc_conv_split_bank_f16f6x_f32" if bank.shape[2] == 4 else "fgvc_conv_split_bank_f16f6p_f32", _ptr(x_split), _ptr(w), _ptr(bias), _ptr(residual), _ptr(bank), N, H, W, Hp, Wp,
              nch * 32, 3, int(relu), int(in_fmt), int(in_scale_log2), int(normalize), _stream(x_split))
    return bank


def conv64_split(x_split: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, H: int, W: int, relu: bool,
                 residual: Optional[torch.Tensor] = None, out_split: Optional[torch.Tensor] = None,
                 out_f32: Optional[torch.Tensor] = None, residual_split: Optional[torch.Tensor] = None,
                 in_fmt: int = ACT_BF16X2, in_scale_log2: int = 0, out_fmt: int = ACT_BF16X2, out_scale_log2: int = 0,
                 overflow: Optional[torch.Tensor] = None) -> None:
    """conv_split for Cin = Cout = 64, 3x3, with register-resident weights (fgvc_conv64_split_fmt_f32).  in_fmt ACT_BF16X2: w, bias
    from prepare_conv64; ACT_F16F8: from prepare_conv64_f16, in_scale_log2 = log2(s_x) + its log2 s_w.  out_fmt / out_scale_log2 /
    overflow as in conv_split.  The residual is dense NHWC f32 (`residual`) or, for bf16 tensors, a padded split NHWC tensor of
    x_split's shape (`residual_split`: hi + lo is added)."""
    x_split = _chk(x_split, torch.int16, "x_split")
    N, Hp, Wp, nch, _ = x_split.shape
    assert nch == 2 and bias.shape == (64,)
    assert tuple(w.shape) == ((2, 9, 2, 2, 2, 64, 8) if in_fmt == ACT_BF16X2 else (2, 9, 2, 2048)), "conv64_split weights / in_fmt"
    assert residual is None or residual_split is None, "one residual, f32 or split"
    for t, dt, shape in ((residual, torch.float32, (N, H, W, 64)), (out_f32, torch.float32, (N, H, W, 64)),
                         (out_split, torch.int16, (N, Hp, Wp, 2, 64)), (residual_split, torch.int16, (N, Hp, Wp, 2, 64))):
        if t is not None:
            assert t.dtype == dt and tuple(t.shape) == shape and t.is_contiguous() and t.device == x_split.device, "conv64_split buffer"
    _lib.call("fgvc_conv64_split_fmt_f32", _ptr(x_split), _ptr(w), _ptr(bias), _ptr(residual), _ptr(residual_split), _ptr(out_split),
              _ptr(out_f32), N, H, W, Hp, Wp, int(relu), int(in_fmt), int(in_scale_log2), int(out_fmt), int(out_scale_log2),
              _ptr(overflow), _stream(x_split))


def conv_s2_split(x_split: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, H: int, W: int, relu: bool,
                  out_split: Optional[torch.Tensor] = None, out_f32: Optional[torch.Tensor] = None, out_fmt: int = ACT_BF16X2,
                  out_scale_log2: int = 0, overflow: Optional[torch.Tensor] = None) -> None:
    """3x3 / stride 2 / pad 1 or 1x1 / stride 2 convolution + bias (+ ReLU) on the bf16 pipe (fgvc_conv_s2_split_f32).
    x_split: padded split NHWC of the (N, H, W) input; w, bias from prepare_conv_s2; outputs for the
    ((H-1)//2+1, (W-1)//2+1) result: out_split (padded split NHWC) and / or out_f32 (dense NHWC f32)."""
    x_split = _chk(x_split, torch.int16, "x_split")
    N, Hp, Wp, nch, _ = x_split.shape
    taps, nch_w, n_ct = w.shape[:3]
    Cout = n_ct * 32
    assert tuple(w.shape[3:]) == (4, 64, 8) and nch_w == nch and taps in (1, 9) and bias.shape == (Cout,)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hop, Wop = conv_pad_dims(Ho, Wo)
    if out_split is not None:
        assert out_split.dtype == torch.int16 and tuple(out_split.shape) == (N, Hop, Wop, Cout // 32, 64) and out_split.is_contiguous()
    if out_f32 is not None:
        assert out_f32.dtype == torch.float32 and tuple(out_f32.shape) == (N, Ho, Wo, Cout) and out_f32.is_contiguous()
    if out_fmt != ACT_BF16X2 and out_split is not None:
        assert overflow is not None and overflow.dtype == torch.int32 and overflow.device == x_split.device
    _lib.call("fgvc_conv_s2_split_fmt_f32", _ptr(x_split), _ptr(w), _ptr(bias), _ptr(out_split), _ptr(out_f32), N, H, W, Hp, Wp,
              nch * 32, Cout, 3 if taps == 9 else 1, Hop, Wop, int(relu), int(out_fmt), int(out_scale_log2), _ptr(overflow),
              _stream(x_split))


def stem7_split(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, relu: bool = True,
                out_split: Optional[torch.Tensor] = None, out_f32: Optional[torch.Tensor] = None, out_fmt: int = ACT_BF16X2,
                out_scale_log2: int = 0, overflow: Optional[torch.Tensor] = None) -> None:
    """7x7 / stride 2 / pad 3 stem (3 -> 64 channels) + bias (+ ReLU) on the bf16 pipe (fgvc_stem7_split_fmt_f32): f32 NCHW
    frames (N, 3, H, W) -> out_f32 (N, Ho, Wo, 64) dense NHWC f32 and / or out_split (padded split NHWC in `out_fmt`: ACT_BF16X2, or
    ACT_F16F8 at scale 2^out_scale_log2 with the overflow word)."""
    x = _chk(x, torch.float32, "x")
    N, C, H, W = x.shape
    assert C == 3 and x.is_contiguous() and tuple(w.shape) == (7, 2, 2, 2, 64, 8) and bias.shape == (64,)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hop, Wop = conv_pad_dims(Ho, Wo)
    if out_split is not None:
        assert out_split.dtype == torch.int16 and tuple(out_split.shape) == (N, Hop, Wop, 2, 64) and out_split.is_contiguous()
    if out_f32 is not None:
        assert out_f32.dtype == torch.float32 and tuple(out_f32.shape) == (N, Ho, Wo, 64) and out_f32.is_contiguous()
    _lib.call("fgvc_stem7_split_fmt_f32", _ptr(x), _ptr(w), _ptr(bias), _ptr(out_split), _ptr(out_f32), N, H, W, Hop, Wop,
              int(relu), int(out_fmt), int(out_scale_log2), _ptr(overflow), _stream(x))


def normalize_nhwc(x: torch.Tensor, normalize: bool = True, split=False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dense NHWC f32 (N,H,W,C) -> (N, H*W, C) f32 rows, L2-normalised (the layout of normalize_to_hwc); split=True / "bf16"
    returns split_bf16() of those rows instead, split="f16" their split_f16x2(), (N, H*W, 2, C) int16, produced in the same
    single pass over x.  `out`: write there."""
    fmt = "bf16" if split is True else split
    assert fmt in (False, "bf16", "f16", "f16f6", "f16f6x")
    x = _chk(x, torch.float32, "x")
    N, H, W, C = x.shape
    shape, dt = ((N, H * W, 4 if fmt == "f16f6x" else 2, C), torch.int16) if split else ((N, H * W, C), torch.float32)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=dt)
    else:
        assert tuple(out.shape) == shape and out.dtype == dt and out.is_contiguous() and out.device == x.device
    if fmt in ("f16f6", "f16f6x"):                    # the rows of split_f16f6p() / split_f16f6x() (fgvc_pair_topk_f16f6's operands), same pass
        assert C == 256
        _lib.call(f"fgvc_normalize_split_{fmt}p_nhwc_f32" if fmt == "f16f6" else "fgvc_normalize_split_f16f6x_nhwc_f32", _ptr(x), _ptr(out), N, C, H, W,
                  int(normalize), _stream(x))
        return out
    _lib.call("fgvc_normalize_split_f16x2_nhwc_f32" if fmt == "f16" else "fgvc_normalize_split_nhwc_f32", _ptr(x),
              _ptr(None if split else out), _ptr(out if split else None), N, C, H, W, int(normalize), _stream(x))
    return out


def unsplit_f16x2(split: torch.Tensor) -> torch.Tensor:
    """(…, 2, C) int16 split_f16x2() features -> (…, C) f32 = (h + l) / 2^14 (2^-22-relative approximation of the rows)."""
    v = split.view(torch.float16).float()
    return (v[..., 0, :] + v[..., 1, :]) * (1.0 / 16384.0)


def unsplit_act(split: torch.Tensor, fmt: int, scale_log2: int = 0) -> torch.Tensor:
    """A padded split NHWC activation tensor (N, Hp, Wp, C/32, 64) int16 in format `fmt` -> (N, Hp, Wp, C) f32 (tests, calibration)."""
    n5 = split.shape
    if fmt == ACT_BF16X2:
        v = split.view(torch.bfloat16).float()
        return (v[..., :32] + v[..., 32:]).reshape(*n5[:3], -1)
    b = split.contiguous().view(torch.uint8).reshape(*n5[:4], 128)
    h = b[..., :64].contiguous().view(torch.float16).float()
    if fmt == ACT_F16X2:
        l = b[..., 64:].contiguous().view(torch.float16).float()
    elif fmt == ACT_F16F6:       # slots 4 / 6: the residual's FP6 block (main, tail + scale byte)
        l = _e2m3_decode(torch.cat([b[..., 64:80], b[..., 96:104]], dim=-1), b[..., 104])
    else:
        l = b[..., 64:96].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -F8_BX
    return ((h + l) * 2.0 ** -scale_log2).reshape(*n5[:3], -1)


def unsplit_bf16(split: torch.Tensor) -> torch.Tensor:
    """(…, 2, C) int16 split features -> (…, C) f32 = hi + lo (2^-17-relative approximation of the rows that were split)."""
    v = split.view(torch.bfloat16).float()
    return v[..., 0, :] + v[..., 1, :]


def gaussian_labels(points: torch.Tensor, Hf: int, Wf: int, stride: int, sigma: float = 6.0,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """points (P,2)=(x,y) -> (HfWf, P) initial labels (vanilla_tracker.py:204-221)."""
    points = _chk(points.to(torch.float32), torch.float32, "points")
    P = points.shape[0]
    if out is None:
        out = torch.empty((Hf * Wf, P), device=points.device, dtype=torch.float32)
    _lib.call("fgvc_gaussian_labels_f32", _ptr(points), P, Hf, Wf, stride, float(sigma), _ptr(out), _stream(points))
    return out


def softargmax_top5(labels: torch.Tensor, Hf: int, Wf: int, h: int, w: int,
                    gauss_points: Optional[torch.Tensor] = None, sigma: float = 6.0) -> torch.Tensor:
    """labels (n_frames, HfWf, P) -> coords (n_frames, P, 2) float64 = (x, y)."""
    labels = _chk(labels, torch.float32, "labels")
    n, P = labels.shape[0], labels.shape[2]
    assert labels.shape[1] == Hf * Wf
    if gauss_points is not None:
        gauss_points = _chk(gauss_points.to(torch.float32), torch.float32, "gauss_points")
        assert gauss_points.shape == (P, 2)
    coords = torch.empty((n, P, 2), device=labels.device, dtype=torch.float64)
    ws_bytes = _lib.load().fgvc_softargmax_workspace_bytes(n, P)
    ws = torch.empty((max(ws_bytes, 4) + 3) // 4, device=labels.device, dtype=torch.float32)
    _lib.call("fgvc_softargmax_top5_f32", _ptr(labels), n, Hf, Wf, P, h, w, _ptr(gauss_points), float(sigma),
              _ptr(coords), _ptr(ws), _stream(labels))
    return coords
