"""CPU restatement of FGVC's label-propagation inference hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module, and only as the
checker / the reported CPU baseline.  The product path (`fgvc_amd`) never imports
it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against OUTPUTS OF THE REFERENCE
ITSELF: `tests/golden/gen_golden.py` imports the genuine reference operators from
/root/reference in the build container (via oracle/ref_import.py) and commits
their inputs/outputs as fixtures under tests/golden/; `tests/test_oracle.py`
checks every function below against those fixtures (re-running gen_golden.py in
the build container reproduces them bit for bit).
Third-party arithmetic that is NOT under /root/reference and therefore stays
"parity unpinned": mmcv-full==1.5.2 `ConvModule` (restated as Conv2d+BatchNorm2d+
ReLU, exact given identical weights) and `mmcv.ops.Correlation` (restated from the
reference's own torch-only twin `masked_attention_efficient_correlation_v2`).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  All tensors are torch CPU tensors; dtype follows the inputs
(float32 = the reference's arithmetic, float64 = the "infinitely precise" arm
used to find near-ties).

Tie policy (torch.topk / np.argsort leave tie order unspecified):
    top-k lists are in CANONICAL order = (score descending, index ascending);
    the soft-argmax read-out prefers the HIGHER flat index among equal values
    (what a stable ascending argsort followed by [-5:] yields).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

NEG_INF = float("-inf")


# ----------------------------------------------------------------------------
# A4  spatial neighbourhood mask
# ----------------------------------------------------------------------------
def neighbor_predicate(dy: torch.Tensor, dx: torch.Tensor, neighbor_range, mode: str = "circle") -> torch.Tensor:
    """The mask predicate on integer offsets (key - query).

    mmpt/models/common/affinity_utils.py:75-112: 'circle' keeps
    sqrt(dy^2+dx^2) < neighbor_range//2 evaluated in float32 (:101-109);
    'square' keeps |dy| <= nr_h//2 and |dx| <= nr_w//2 (:88-96).
    """
    if mode == "circle":
        radius = neighbor_range // 2
        d = (dy.to(torch.float32) ** 2 + dx.to(torch.float32) ** 2) ** 0.5
        return d < radius
    if mode == "square":
        nr = (neighbor_range, neighbor_range) if isinstance(neighbor_range, int) else tuple(neighbor_range)
        return (dy.abs() <= nr[0] // 2) & (dx.abs() <= nr[1] // 2)
    raise ValueError(mode)


def spatial_neighbor(height: int, width: int, neighbor_range, mode: str = "circle") -> torch.Tensor:
    """(HW, HW) bool, mask[key, query]  (affinity_utils.py:75-112, dim=1 layout)."""
    ys = torch.arange(height).view(height, 1).expand(height, width).reshape(-1)
    xs = torch.arange(width).view(1, width).expand(height, width).reshape(-1)
    dy = ys.view(-1, 1) - ys.view(1, -1)
    dx = xs.view(-1, 1) - xs.view(1, -1)
    return neighbor_predicate(dy, dx, neighbor_range, mode)


def radius_predicate_r2max(radius: float) -> int:
    """Largest integer d2 with float32 sqrt(d2) < radius (v2's `dist < radius`,
    local_attention.py:463-467, and the circle mask above).  -1 if none."""
    r = np.float32(radius)
    d2 = int(math.ceil(float(radius) ** 2)) + 2
    while d2 >= 0 and not (np.sqrt(np.float32(d2)) < r):
        d2 -= 1
    return d2


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def l2_normalize(x: torch.Tensor, dim: int = 1) -> torch.Tensor:
    """F.normalize(p=2, eps=1e-12) (local_attention.py:308-310)."""
    return x / x.norm(p=2, dim=dim, keepdim=True).clamp_min(1e-12)


def topk_canonical(aff: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k along dim 0 of an (M, S) slab in canonical order.

    Restates `cur_affinity.topk(k, dim=1)` (local_attention.py:356) with the tie
    order fixed to (score desc, index asc).  Returns (values (k,S), indices (k,S)).
    """
    M, S = aff.shape
    k = min(k, M)
    vals, _ = aff.topk(k, dim=0)
    kth = vals[k - 1]                                   # (S,)
    gt = aff > kth                                      # strictly better than the k-th value
    need = k - gt.sum(0)                                # how many k-th-valued entries to take
    eq = aff == kth
    take = eq & (eq.cumsum(0) <= need)                  # lowest indices among the equals
    sel = gt | take                                     # exactly k per column
    idx_by_index = sel.t().nonzero()[:, 1].view(S, k)   # per column, ascending index
    v = aff.t().gather(1, idx_by_index)
    order = torch.sort(v, dim=1, descending=True, stable=True)[1]
    idx = idx_by_index.gather(1, order).t().contiguous()
    val = v.gather(1, order).t().contiguous()
    return val, idx


# ----------------------------------------------------------------------------
# A5''  dense correlation volume
# ----------------------------------------------------------------------------
def corr_volume(query: torch.Tensor, key: torch.Tensor, temperature: float = 1.0,
                normalize: bool = True, q_slice: Optional[slice] = None) -> torch.Tensor:
    """A[t*HWk + j, i] = <k_hat[:,t,j], q_hat[:,i]> / temperature.

    query (C,Hq,Wq), key (C,T,Hk,Wk) -> (T*HWk, HWq[q_slice]).
    local_attention.py:321-323 (chunked), :231 (`masked_attention`),
    affinity_utils.py:6-21 (`compute_affinity`, transposed operand roles).
    """
    if key.dim() == 3:
        key = key.unsqueeze(1)
    if normalize:
        query = l2_normalize(query, 0)
        key = l2_normalize(key, 0)
    C = query.shape[0]
    qv = query.reshape(C, -1)
    kv = key.reshape(C, -1)
    if q_slice is not None:
        qv = qv[:, q_slice]
    return torch.einsum("ci,cj->ij", kv, qv) / temperature


def mask_slab(Hq: int, Wq: int, Hk: int, Wk: int, T: int, q_index: torch.Tensor,
              neighbor_range, mode: str, non_mask_len: int = 0) -> torch.Tensor:
    """(T*HkWk, len(q_index)) bool validity slab (local_attention.py:329-353)."""
    assert (Hq, Wq) == (Hk, Wk), "the analytic mask is defined on equal grids"
    qy, qx = q_index // Wq, q_index % Wq
    ky = torch.arange(Hk).view(Hk, 1).expand(Hk, Wk).reshape(-1)
    kx = torch.arange(Wk).view(1, Wk).expand(Hk, Wk).reshape(-1)
    m = neighbor_predicate(ky.view(-1, 1) - qy.view(1, -1), kx.view(-1, 1) - qx.view(1, -1),
                           neighbor_range, mode)              # (HkWk, S)
    m = m.unsqueeze(0).expand(T, -1, -1).clone()
    m[:non_mask_len] = True
    return m.reshape(T * Hk * Wk, -1)


# ----------------------------------------------------------------------------
# A5  affinity top-k  +  propagation
# ----------------------------------------------------------------------------
def affinity_topk(query: torch.Tensor, key: torch.Tensor, topk: int, temperature: float = 1.0,
                  neighbor_range=None, mask_mode: str = "circle", mask: Optional[torch.Tensor] = None,
                  normalize: bool = True, non_mask_len: int = 0, step: int = 512,
                  q_index: Optional[torch.Tensor] = None, canonical: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per query pixel: canonical top-k (logit, flat key index t*HWk+j).

    query (C,Hq,Wq), key (C,T,Hk,Wk).  Returns idx (S,k) int64, logit (S,k)
    where S = HqWq or len(q_index).  Follows local_attention.py:308-356.
    Either `neighbor_range`(+`mask_mode`) (analytic) or a dense `mask`
    (HkWk, HqWq) bool, or neither (no mask).
    canonical=False uses plain `torch.topk` exactly as the reference does (tie order unspecified): that is
    the form timed as the CPU baseline; canonical=True (default) is the form used as the checker.
    """
    if key.dim() == 3:
        key = key.unsqueeze(1)
    C, Hq, Wq = query.shape
    _, T, Hk, Wk = key.shape
    if normalize:
        query = l2_normalize(query, 0)
        key = l2_normalize(key, 0)
    qv = query.reshape(C, -1)
    kv = key.reshape(C, -1)
    if q_index is None:
        q_index = torch.arange(Hq * Wq)
    idx_out, val_out = [], []
    for p in range(0, q_index.numel(), step):
        qi = q_index[p:p + step]
        if mask is not None:
            m = mask[:, qi].unsqueeze(0).expand(T, -1, -1).clone()
            m[:non_mask_len] = True
            m = m.reshape(T * Hk * Wk, -1)
        elif neighbor_range is not None:
            m = mask_slab(Hq, Wq, Hk, Wk, T, qi, neighbor_range, mask_mode, non_mask_len)
        else:
            m = None
        i, v = affinity_chunk(kv, qv[:, qi], m, topk, temperature, canonical)
        idx_out.append(i)
        val_out.append(v)
    return torch.cat(idx_out, 0), torch.cat(val_out, 0)


def affinity_chunk(kv: torch.Tensor, qv_chunk: torch.Tensor, mask_chunk: Optional[torch.Tensor], topk: int,
                   temperature: float, canonical: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """One query chunk of the reference's loop body (local_attention.py:321-356): einsum / temperature,
    masked_fill_(-inf), topk.  kv (C, T*HWk), qv_chunk (C, s), mask_chunk (T*HWk, s) bool or None.
    Returns idx (s,k), logit (s,k).  This is also the unit the CPU baseline times (with a pre-built mask,
    as the reference pre-builds its mask once per video, vanilla_tracker.py:332-340)."""
    aff = torch.einsum("ci,cj->ij", kv, qv_chunk) / temperature          # :321-323
    if mask_chunk is not None:
        aff = aff.masked_fill_(~mask_chunk, NEG_INF)                      # :353
    v, i = topk_canonical(aff, topk) if canonical else aff.topk(topk, dim=0)   # :356
    return i.t(), v.t()


def topk_weights(logit: torch.Tensor, mode: str = "softmax") -> torch.Tensor:
    """local_attention.py:368-373: softmax over the k, or clamp(min=0)**2."""
    if mode == "softmax":
        return logit.softmax(dim=-1)
    if mode == "cosine":
        return logit.clamp(min=0) ** 2
    raise ValueError(mode)


def propagate_topk(value: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """out[p, i] = sum_r weight[i, r] * value[p, idx[i, r]]  (local_attention.py:360-375).

    value (P, T*HWk) (flattened), idx/weight (S,k) -> (P, S).
    """
    P = value.shape[0]
    g = value.reshape(P, -1)[:, idx.reshape(-1)].reshape(P, *idx.shape)   # (P,S,k)
    return (g * weight.unsqueeze(0)).sum(-1)


def dense_affinity(query, key, temperature=1.0, normalize=True, sim_mode="dot_product", neighbor_range=None,
                   mask_mode="circle", mask=None, non_mask_len=0) -> torch.Tensor:
    """The whole masked affinity slab (T*HWk, HWq) of local_attention.py:308-353, for either `sim_mode`:
    'dot_product' = <k,q>/temperature (:321-323); 'l2-distance' = (2<k,q> - |k|^2)/sqrt(C), no temperature (:324-327)."""
    C, Hq, Wq = query.shape
    _, T, Hk, Wk = key.shape
    if normalize:
        query, key = l2_normalize(query, 0), l2_normalize(key, 0)
    qv, kv = query.reshape(C, -1), key.reshape(C, -1)
    if sim_mode == "dot_product":
        aff = torch.einsum("ci,cj->ij", kv, qv) / temperature
    elif sim_mode == "l2-distance":
        aff = (2 * (kv.t() @ qv) - kv.pow(2).sum(0).unsqueeze(1)) / math.sqrt(C)
    else:
        raise ValueError(sim_mode)
    if mask is not None:
        m = mask.unsqueeze(0).expand(T, -1, -1).clone()
        m[:non_mask_len] = True
        aff = aff.masked_fill(~m.reshape(T * Hk * Wk, -1), NEG_INF)
    elif neighbor_range is not None:
        aff = aff.masked_fill(~mask_slab(Hq, Wq, Hk, Wk, T, torch.arange(Hq * Wq), neighbor_range, mask_mode, non_mask_len),
                              NEG_INF)
    return aff


def masked_attention_efficient(query, key, value, mask=None, temperature=1, topk=None, normalize=True,
                               step=32, non_mask_len=0, mode="softmax", neighbor_range=None,
                               mask_mode="circle", sim_mode="dot_product"):
    """Same contract as local_attention.py:267-389 for N=1.

    query (1,C,Hq,Wq), key (1,C,T,Hk,Wk), value (1,P,T,Hk,Wk) -> (1,P,Hq,Wq).
    topk=None: weights over EVERY key (softmax, or clamp(min=0)**2) instead of the k best (:376-383).
    """
    assert query.shape[0] == 1
    if key.dim() == 4:
        key, value = key.unsqueeze(2), value.unsqueeze(2)
    P = value.shape[1]
    if topk is None or sim_mode != "dot_product":
        aff = dense_affinity(query[0], key[0], temperature, normalize, sim_mode, neighbor_range, mask_mode, mask, non_mask_len)
        if topk is None:
            w = aff.softmax(0) if mode == "softmax" else aff.clamp(min=0) ** 2                   # :377-380
            out = value[0].reshape(P, -1) @ w                                                    # :383
        else:
            v, i = topk_canonical(aff, topk)
            out = propagate_topk(value[0].reshape(P, -1), i.t(), topk_weights(v.t(), mode))
        return out.reshape(1, P, query.shape[2], query.shape[3])
    idx, logit = affinity_topk(query[0], key[0], topk, temperature, neighbor_range, mask_mode, mask,
                               normalize, non_mask_len, step)
    w = topk_weights(logit, mode)
    out = propagate_topk(value[0].reshape(P, -1), idx, w)
    return out.reshape(1, P, query.shape[2], query.shape[3])


# ----------------------------------------------------------------------------
# A5''  dense formulations without a shipped caller: propagate, non_local_attention, local_square_attention
# ----------------------------------------------------------------------------
def propagate(img: torch.Tensor, affinity: torch.Tensor, topk: Optional[int] = None) -> torch.Tensor:
    """affinity_utils.py:33-50: img (N,C,H,W), affinity (N, HW src, HW dst) -> img @ affinity; topk: subtract each column's k-th
    largest entry (:39-41), clamp at 0 (:43), normalise by the column sum clamped at 1e-12 (:45).  (Out of place: the
    reference rewrites the caller's affinity.)"""
    N, C, H, W = img.shape
    if topk is not None:
        kth = affinity.topk(dim=1, k=topk)[0][:, topk - 1].view(N, 1, H * W)
        affinity = (affinity - kth).clamp(min=0)
        affinity = affinity / affinity.sum(keepdim=True, dim=1).clamp(min=1e-12)
    return torch.bmm(img.reshape(N, C, -1), affinity).reshape(N, C, H, W)


def non_local_attention(tar: torch.Tensor, refs: torch.Tensor, per_ref: bool = True, temperature: float = 1.0, mask=None,
                        scaling: bool = False, norm: bool = False, att_only: bool = False):
    """correlation.py:32-83, mode='dot': tar (B,C,H,W), refs (B,t,C,H,W) -> att (B,t,HW tar,HW ref) = <tar_i, ref_tj> / temperature
    (:52-56) [/ sqrt(C) (:64-66)] [masked_fill(-inf) (:69-71)]; att_only returns it (:73); else softmax over the reference
    pixels of each frame (per_ref, :77-80) or over all frames' pixels pooled (:81-85).  Returns (B, att) like the reference,
    whose first value is the batch size bound by `_, t, feat_dim, w_, h_ = refs.shape` (:46)."""
    B, t, C = refs.shape[:3]
    tv = tar.flatten(2).permute(0, 2, 1)
    rv = refs.flatten(3).permute(0, 1, 3, 2)
    if norm:
        tv, rv = F.normalize(tv, dim=-1), F.normalize(rv, dim=-1)
    att = torch.einsum("bic,btjc->btij", tv, rv) / temperature
    if scaling:
        att = att / math.sqrt(C)
    if mask is not None:
        att = att.masked_fill(~mask.bool(), NEG_INF)
    if att_only:
        return att
    if per_ref:
        return B, att.softmax(-1)
    return B, att.permute(0, 2, 1, 3).flatten(2).softmax(-1)


def local_square_attention(query, key, value, kernel_size, temperature: float = 1.0, topk: Optional[int] = None,
                           batch_as_context: bool = False) -> torch.Tensor:
    """local_attention.py:38-103: RAW dot products of each query pixel with the zero-padded (kh x kw) window of the key around it
    (F.unfold, padding k // 2, :67-68) / temperature (:96), optionally the topk largest (:99-104); the attention values themselves
    weight the unfolded values (:106, no softmax).  batch_as_context (:83-92): the windows of all key batch entries are one
    context for a single query."""
    ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
    pad = (ks[0] // 2, ks[1] // 2)
    N, C, H, W = key.shape
    P = value.shape[1]
    uk = F.unfold(key, ks, padding=pad).view(N, C, ks[0] * ks[1], H * W)
    uv = F.unfold(value, ks, padding=pad).view(N, P, ks[0] * ks[1], H * W)
    uq = query.reshape(query.shape[0], C, 1, H * W)
    if batch_as_context:
        uk = uk.transpose(0, 1).reshape(1, C, N * ks[0] * ks[1], H * W)
        uv = uv.transpose(0, 1).reshape(1, P, N * ks[0] * ks[1], H * W)
    att = (uq * uk).sum(1, keepdim=True) / temperature
    if topk is not None:
        att, ti = att.topk(k=topk, dim=2)
        uv = uv.gather(2, ti.expand(-1, P, -1, -1))
    return (att * uv).sum(2).reshape(-1, P, H, W)


# ----------------------------------------------------------------------------
# A7 / A7'  single-scale local-window correlation + top-k
# ----------------------------------------------------------------------------
def local_corr(query: torch.Tensor, keys: torch.Tensor, radius: int, normalize: bool = True) -> torch.Tensor:
    """mmcv.ops.Correlation(max_displacement=R, kernel_size=1) semantics as relied on at
    vanilla_tracker.py:435-443: out[k, dy*(2R+1)+dx, y, x] = sum_c q[c,y,x]*key[k,c,y+dy-R,x+dx-R],
    zero outside.  query (C,H,W), keys (K,C,H,W) -> (K,(2R+1)^2,H,W).  (Twin: local_attention.py:1190-1198.)
    """
    if normalize:
        query = l2_normalize(query, 0)
        keys = l2_normalize(keys, 1)
    K, C, H, W = keys.shape
    L = 2 * radius + 1
    unf = F.unfold(keys, kernel_size=L, padding=radius).reshape(K, C, L * L, H, W)
    return (unf * query.view(1, C, 1, H, W)).sum(1)


def local_corr_topk(query, keys, values, radius: int, topk: int, temperature: float = 1.0,
                    normalize: bool = True):
    """HRVanillaTracker.forward_test_main inner step (vanilla_tracker.py:547-566) /
    masked_attention_efficient_correlation_v2 (local_attention.py:1190-1240).

    query (C,H,W), keys (K,C,H,W), values (K,P,H,W) -> out (P,H,W), idx (HW,k), logit (HW,k)
    where idx = k_slot*(2R+1)^2 + dy*(2R+1)+dx and logit = corr/temperature (divided AFTER top-k).
    """
    K, P, H, W = values.shape
    L = 2 * radius + 1
    corr = local_corr(query, keys, radius, normalize).reshape(K * L * L, H * W)
    val, idx = topk_canonical(corr, topk)                           # :558 / :1229
    unf_v = F.unfold(values, kernel_size=L, padding=radius).reshape(K, P, L * L, H * W)
    unf_v = unf_v.permute(1, 0, 2, 3).reshape(P, K * L * L, H * W)  # :550-555
    g = unf_v.gather(1, idx.unsqueeze(0).expand(P, -1, -1))         # :561
    logit = val / temperature                                       # :563
    w = logit.softmax(0)                                            # :564
    out = (g * w.unsqueeze(0)).sum(1).reshape(P, H, W)              # :566
    return out, idx.t().contiguous(), logit.t().contiguous()


def get_coord(query, key, radius: int, topk: int, temperature: float, scale: int, normalize: bool = True):
    """HRVanillaTracker.get_coord (vanilla_tracker.py:445-488) for one key frame.
    query, key (C,H,W) -> (2,H,W) expected (x,y) image coordinates (feature coordinate * scale)."""
    C, H, W = query.shape
    L = 2 * radius + 1
    corr = local_corr(query, key.unsqueeze(0), radius, normalize).reshape(L * L, H * W)       # :450, :470
    xs = (torch.arange(W, dtype=torch.float32) * scale).view(1, W).expand(H, W)
    ys = (torch.arange(H, dtype=torch.float32) * scale).view(H, 1).expand(H, W)
    grid = torch.stack([xs, ys], 0).unsqueeze(0)                                               # :461-464 (sampled grid)
    grid_unf = F.unfold(grid, kernel_size=L, padding=radius).reshape(2, L * L, H * W)          # :467
    val, idx = topk_canonical(corr, topk)                                                      # :476
    g = grid_unf.gather(1, idx.unsqueeze(0).expand(2, -1, -1))                                 # :479
    w = (val / temperature).softmax(0)                                                         # :481-482
    return (g * w.unsqueeze(0)).sum(1).reshape(2, H, W)                                        # :485


# ----------------------------------------------------------------------------
# A6  coarse-to-fine refine
# ----------------------------------------------------------------------------
def c2f_attention(query, key, query_fine, key_fine, value, topk: int, temperature: float = 1.0,
                  neighbor_range=None, mask_mode="circle", normalize: bool = True,
                  radius_fine: int = 12, non_mask_len: int = 0, step: int = 512, mode: str = "softmax"):
    """masked_attention_efficient_c2f (local_attention.py:721-880), N=1.

    query (C,H,W), key (C,T,H,W), query_fine (Cf,sH,sW), key_fine (Cf,T,sH,sW),
    value (P,T,sH,sW) -> out (P,H,W), coarse argmax (T,HW), idx (HW,k), logit (HW,k).
    """
    C, H, W = query.shape
    T = key.shape[1]
    Cf, Hf, Wf = query_fine.shape
    P = value.shape[0]
    scale = key_fine.shape[2] // key.shape[2]                        # :769
    if normalize:                                                    # :771-775
        query, key = l2_normalize(query, 0), l2_normalize(key, 0)
        query_fine, key_fine = l2_normalize(query_fine, 0), l2_normalize(key_fine, 0)
    L = 2 * radius_fine + 1
    qf = query_fine[:, ::scale, ::scale].reshape(Cf, -1)             # :785
    kf_unf = F.unfold(key_fine.transpose(0, 1), kernel_size=L, padding=radius_fine,
                      stride=scale).reshape(T, Cf, L * L, H * W)     # :790
    v_unf = F.unfold(value.transpose(0, 1), kernel_size=L, padding=radius_fine,
                     stride=scale).reshape(T, P, L * L, H * W)       # :793
    qv, kv = query.reshape(C, -1), key.reshape(C, -1)
    HW = H * W
    outs, arg_all, idx_all, logit_all = [], [], [], []
    for p in range(0, HW, step):
        qi = torch.arange(p, min(HW, p + step))
        s = qi.numel()
        aff = torch.einsum("ci,cj->ij", kv, qv[:, qi]) / temperature            # :804-806
        if neighbor_range is not None:
            m = mask_slab(H, W, H, W, T, qi, neighbor_range, mask_mode, non_mask_len)
            aff = aff.masked_fill(~m, NEG_INF)                                   # :832
        aff = aff.reshape(T, HW, s).softmax(1)                                   # :835
        am = aff.argmax(1)                                                       # :837 (T,s)
        gi = am.view(T, 1, 1, s)
        k_sel = kf_unf.gather(3, gi.expand(T, Cf, L * L, s))                     # :841
        fine = (k_sel * qf[:, qi].view(1, Cf, 1, s)).sum(1) / temperature        # :847 (T,LL,s)
        fine = fine.reshape(T * L * L, s)
        v_sel = v_unf.gather(3, gi.expand(T, P, L * L, s))                       # :853
        v_sel = v_sel.permute(1, 0, 2, 3).reshape(P, T * L * L, s)               # :855
        val, idx = topk_canonical(fine, topk)                                    # :859
        g = v_sel.gather(1, idx.unsqueeze(0).expand(P, -1, -1))                  # :862
        w = val.softmax(0) if mode == "softmax" else val.clamp(min=0) ** 2       # :858-861
        outs.append((g * w.unsqueeze(0)).sum(1))                                 # :870
        arg_all.append(am); idx_all.append(idx.t()); logit_all.append(val.t())
    out = torch.cat(outs, 1).reshape(P, H, W)
    return out, torch.cat(arg_all, 1), torch.cat(idx_all, 0), torch.cat(logit_all, 0)


# ----------------------------------------------------------------------------
# A3  initial Gaussian labels,  A8 upsample,  A9 soft-argmax read-out
# ----------------------------------------------------------------------------
def gaussian_labels(points_xy: torch.Tensor, h: int, w: int, stride: int, sigma: float = 6.0):
    """vanilla_tracker.py:193-221.  points (P,2)=(x,y) -> full (P,h,w), feature-res (P,ceil(h/s),ceil(w/s))."""
    xs = torch.arange(w, dtype=torch.float32).view(1, 1, w)
    ys = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    cx = points_xy[:, 0].to(torch.float32).view(-1, 1, 1)
    cy = points_xy[:, 1].to(torch.float32).view(-1, 1, 1)
    g = torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma ** 2))
    return g, g[:, ::stride, ::stride]


def upsample_bilinear(label: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """vanilla_tracker.py:396-400: F.interpolate(bilinear, align_corners=False). (P,Hf,Wf)->(P,h,w)."""
    return F.interpolate(label.unsqueeze(0), size=(h, w), mode="bilinear", align_corners=False)[0]


def img2coord(maps: np.ndarray, topk: int = 5) -> np.ndarray:
    """vanilla_tracker.py:172-191.  maps (T,P,h,w) float32 -> coords (2,P,T) float64.

    Canonical tie order: stable ascending argsort, last `topk` (higher index wins ties).
    """
    T, P, h, w = maps.shape
    flat = maps.reshape(T, P, -1)
    order = np.argsort(flat, axis=-1, kind="stable")[..., -topk:]           # :181
    v = np.take_along_axis(flat, order, axis=-1)                             # :182
    v = v / (np.sum(v, keepdims=True, axis=-1) + 1e-9)                       # :183 (stays float32)
    x = order % w                                                            # :184
    y = order // w                                                           # :185
    coords = np.zeros((2, P, T), dtype=float)
    coords[0] = np.sum(x * v, axis=-1).T                                     # :187 (int64*f32 -> f64)
    coords[1] = np.sum(y * v, axis=-1).T
    coords[:, np.sum(flat.transpose(1, 0, 2), axis=-1) == 0] = -1            # :189
    return coords


def readout_ties(maps: np.ndarray, topk: int = 5, rel: float = 1e-5) -> np.ndarray:
    """(P,T) bool: the `topk`-th and (`topk`+1)-th largest values of a map are EQUAL up to `rel` of the map's maximum, i.e. which
    pixel img2coord's argsort keeps is unspecified (np.argsort, vanilla_tracker.py:181) or decided by the last bit of the
    arithmetic, and a comparison with the reference must skip that read-out.  Typical source: the bilinear upsample replicates
    border rows/columns, so a maximum at the border is a run of equal values."""
    T, P, h, w = maps.shape
    srt = np.sort(maps.reshape(T, P, -1), axis=-1)
    return ((srt[..., -topk] - srt[..., -topk - 1]) <= rel * np.abs(srt[..., -1])).T


# ----------------------------------------------------------------------------
# A1  ResNet-18 trunk (mmcv ConvModule naming)
# ----------------------------------------------------------------------------
class _CM(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x) if self.act else x


class _Block(nn.Module):
    """BasicBlock (resnet.py:16-116): conv3x3(stride)-BN-ReLU, conv3x3-BN, +identity, ReLU."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = _CM(cin, cout, 3, stride, 1, True)
        self.conv2 = _CM(cout, cout, 3, 1, 1, False)
        self.downsample = _CM(cin, cout, 1, stride, 0, False) if (stride != 1 or cin != cout) else None

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        return F.relu(self.conv2(self.conv1(x)) + idt)


class ResNet18(nn.Module):
    """resnet.py:329-638 for depth=18, pool_type in {'none','max'}; returns stage out_index.

    state_dict keys equal the reference's (conv1.conv.weight, layer1.0.conv1.bn.running_mean, ...).
    Like the reference it builds all four stages; unlike it, forward stops after out_index
    (the later stages do not influence the returned tensor, resnet.py:619-627).
    """

    def __init__(self, strides=(1, 2, 2, 2), out_index=3, pool_type="max"):
        super().__init__()
        self.conv1 = _CM(3, 64, 7, 2, 3, True)                       # resnet.py:457-466
        self.pool = nn.MaxPool2d(3, 2, 1) if pool_type == "max" else None
        cin = 64
        for i, s in enumerate(strides):
            planes = 64 * 2 ** i
            setattr(self, f"layer{i + 1}", nn.Sequential(_Block(cin, planes, s), _Block(planes, planes, 1)))
            cin = planes
        self.out_index = out_index

    def forward(self, x):
        x = self.conv1(x)
        if self.pool is not None:
            x = self.pool(x)
        for i in range(self.out_index + 1):
            x = getattr(self, f"layer{i + 1}")(x)
        return x


def seeded_resnet_state(seed: int = 0, strides=(1, 1, 1, 4), pool_type="none", trained_like: bool = False):
    """Synthetic weights per SURVEY.md section 8d: kaiming-normal(fan_out) convs, BN gamma=1 beta=0,
    running stats (0,1); the last BN of each block is NOT zeroed.
    trained_like (round 6): BatchNorm layers as a trained checkpoint has them -- gamma in [0.5, 1.5], beta ~ N(0, 0.1), running_mean
    ~ N(0, 0.2), running_var in [0.5, 2] -- drawn AFTER the convolutions from the same generator (the convolution weights of a seed do
    not change with the flag)."""
    g = torch.Generator().manual_seed(seed)
    net = ResNet18(strides, 3, pool_type)
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith("conv.weight"):
            fan_out = v.shape[0] * v.shape[2] * v.shape[3]
            sd[k] = torch.randn(v.shape, generator=g) * math.sqrt(2.0 / fan_out)
    if trained_like:
        for k, v in sd.items():
            if k.endswith("bn.weight"):
                sd[k] = 0.5 + torch.rand(v.shape, generator=g)
            elif k.endswith("bn.bias"):
                sd[k] = 0.1 * torch.randn(v.shape, generator=g)
            elif k.endswith("bn.running_mean"):
                sd[k] = 0.2 * torch.randn(v.shape, generator=g)
            elif k.endswith("bn.running_var"):
                sd[k] = 0.5 + 1.5 * torch.rand(v.shape, generator=g)
    return sd


# ----------------------------------------------------------------------------
# A2/A10  tracker driver
# ----------------------------------------------------------------------------
def key_slots(frame_idx: int, precede_frames: int = 5, with_first: bool = True):
    """Key/value frame list for query frame `frame_idx` (vanilla_tracker.py:346-362):
    [0] + [max(0,idx-p) .. idx-1]; frame 0 appears twice while idx <= p."""
    ks = list(range(max(0, frame_idx - precede_frames), frame_idx))
    return ([0] + ks) if with_first else ks


def forward_test_main(feats: torch.Tensor, query_xy: torch.Tensor, h: int, w: int, *, precede_frames=5,
                      topk=10, temperature=0.07, neighbor_range=30, mask_mode="circle", with_first=True,
                      with_first_neighbor=True, normalize=True, step=512, mode="softmax", return_all=False):
    """VanillaTracker.forward_test_main after feature extraction (vanilla_tracker.py:321-412).

    feats (T,C,Hf,Wf) encoder outputs, query_xy (P,2)=(x,y) at frame 0.
    Returns trajectories_pred (1,T,P,2) float64 [+ per-frame internals].
    """
    T, C, Hf, Wf = feats.shape
    stride = h // Hf                                                   # :197
    full0, lab0 = gaussian_labels(query_xy, h, w, stride)
    assert lab0.shape[-2:] == (Hf, Wf), (lab0.shape, Hf, Wf)
    P = lab0.shape[0]
    labels = [lab0]
    preds = [full0]
    idxs, logits = [], []
    for f in range(1, T):
        ks = key_slots(f, precede_frames, with_first)
        key = feats[ks].transpose(0, 1)                                # (C,T',Hf,Wf)
        val = torch.stack([labels[k] for k in ks], 1)                  # (P,T',Hf,Wf)
        idx, logit = affinity_topk(feats[f], key, topk, temperature, neighbor_range, mask_mode, None,
                                   normalize, 0 if with_first_neighbor else 1, step)
        out = propagate_topk(val.reshape(P, -1), idx, topk_weights(logit, mode)).reshape(P, Hf, Wf)
        labels.append(out)
        preds.append(upsample_bilinear(out, h, w))
        idxs.append(idx); logits.append(logit)
    seg = torch.stack(preds, 0).numpy()                                # (T,P,h,w)  :404
    coords = img2coord(seg)                                            # :406
    traj = torch.from_numpy(coords).permute(2, 1, 0).unsqueeze(0)      # :407
    if return_all:
        return traj, dict(labels=torch.stack(labels, 0), idx=idxs, logit=logits)
    return traj


def forward_test(encode, rgbs: torch.Tensor, query_points: torch.Tensor, trajectories, visibilities, main=None, **cfg):
    """VanillaTracker.forward_test regrouping by query time (vanilla_tracker.py:246-303).

    encode(frames (T,3,h,w)) -> feats (T,C,Hf,Wf).  rgbs (1,T,3,h,w), query_points (1,P,3)=(t,x,y).
    `main` = the per-group driver (default forward_test_main; hr_forward_test_main for HRVanillaTracker, which inherits
    this regrouping).
    """
    forward_test_main = main if main is not None else globals()["forward_test_main"]
    B, T, P = trajectories.shape[:3]
    h, w = rgbs.shape[-2:]
    if not cfg.get("with_first", False):                               # :246 (default False; the slot list's default is True, :353)
        traj = forward_test_main(encode(rgbs[0]), query_points[0, :, 1:], h, w, **cfg)
        return trajectories, visibilities, traj, torch.zeros_like(visibilities), query_points
    ts = torch.unique(query_points[:, :, 0])
    qp_r = torch.zeros_like(query_points)
    tr_r, vi_r = torch.zeros_like(trajectories), torch.zeros_like(visibilities)
    tp_r = torch.zeros_like(trajectories)
    K = 0
    for t in [int(v) for v in ts]:
        sel = query_points[0, :, 0] == t
        n = int(sel.sum())
        qp = query_points[:, sel].clone()
        qp_r[:, K:K + n] = qp
        traj = forward_test_main(encode(rgbs[0, t:]), qp[0, :, 1:], h, w, **cfg)    # :284
        tp_r[:, t:, K:K + n] = traj.to(tp_r.dtype)                                   # zero prefix :286
        tr_r[:, :, K:K + n] = trajectories[:, :, sel]
        vi_r[:, :, K:K + n] = visibilities[:, :, sel]
        K += n
    return tr_r, vi_r, tp_r, torch.zeros_like(visibilities), qp_r


def hr_forward_test_main(feats: torch.Tensor, query_xy: torch.Tensor, h: int, w: int, *, radius: int, precede_frames=5,
                         topk=10, temperature=1.0, with_first=True, normalize=True, return_all=False, save_mem=False):
    """HRVanillaTracker.forward_test_main ("backward warping") after feature extraction (vanilla_tracker.py:492-585):
    per frame a local (2R+1)^2 window over every key slot (mmcv Correlation, :547), top-k over K*(2R+1)^2 of the RAW
    correlation (:558), temperature then softmax (:563-564), labels gathered from the unfolded label maps (:550-561).
    `normalize` is the reference's `withnorm` key (:437), `save_mem` its `save_mem` key (:432).  feats (T,C,Hf,Wf), query_xy (P,2)=(x,y) at frame 0.
    Returns trajectories_pred (1,T,P,2) float64 [+ internals]."""
    T, C, Hf, Wf = feats.shape
    stride = h // Hf
    full0, lab0 = gaussian_labels(query_xy, h, w, stride)
    labels, preds, idxs, logits = [lab0], [full0], [], []
    for f in range(1, T):
        if save_mem:      # :537-545: ONE key frame (frame key_start, features re-extracted; no first-frame slot, whatever with_first says)
            ks = [max(0, f - precede_frames)]                                           # against the label maps of key_start..f-1 (:520-521):
            if f - ks[0] != 1:                                                          # more than one map cannot be reshaped to one key frame (:552)
                raise RuntimeError("save_mem=True pairs one key frame with frame - key_start label maps: runs only for precede_frames = 1")
        else:
            ks = key_slots(f, precede_frames, with_first)                               # :521-536
        out, idx, logit = local_corr_topk(feats[f], feats[ks], torch.stack([labels[k] for k in ks], 0), radius, topk,
                                          temperature, normalize)
        labels.append(out)
        preds.append(upsample_bilinear(out, h, w))                                      # :569-573
        idxs.append(idx); logits.append(logit)
    seg = torch.stack(preds, 0).numpy()
    coords = img2coord(seg)                                                             # :577-579
    traj = torch.from_numpy(coords).permute(2, 1, 0).unsqueeze(0)                       # :580
    if return_all:
        return traj, dict(labels=torch.stack(labels, 0), idx=idxs, logit=logits, ties=readout_ties(seg).T)
    return traj


def hr_forward_test_forward(feats: torch.Tensor, ref_yx: torch.Tensor, h: int, w: int, *, radius: int, precede_frames=5,
                            topk=10, temperature=1.0, normalize=True) -> torch.Tensor:
    """HRVanillaTracker.forward_test_forward ("forward warping", vanilla_tracker.py:591-645): the points are pushed through
    a chain of coordinate fields; the field of step f has QUERY = frame max(0, f - precede_frames) and KEY = frame f (:622-637)
    and is sampled bilinearly (align_corners=True) at the current coordinates / scale (:639).
    feats (T,C,Hf,Wf), ref_yx (2,P) = (y,x) rows -> (2,P,T) float64, rows (x,y)."""
    T, C, Hf, Wf = feats.shape
    scale = w // Wf                                                                     # :609
    coord = torch.flip(ref_yx, (0,)).float()                                            # :611
    coords = [coord]
    for f in range(1, T):
        start = max(0, f - precede_frames)                                              # :617
        field = get_coord(feats[start], feats[f], radius, topk, temperature, scale, normalize)     # :636 (2,Hf,Wf)
        g = (coord.clone() / scale).t()                                                 # (P,2) = (x,y) feature coordinates
        gx = g[:, 0] * 2.0 / max(Wf - 1, 1) - 1.0                                       # corr_lookup.py:61-63
        gy = g[:, 1] * 2.0 / max(Hf - 1, 1) - 1.0
        grid = torch.stack([gx, gy], -1).view(1, -1, 1, 2)
        coord = F.grid_sample(field.unsqueeze(0), grid, "bilinear", "zeros", True)[0, :, :, 0]     # (2,P)
        coords.append(coord)
    return torch.stack(coords, -1).double()                                             # :642-644


# ----------------------------------------------------------------------------
# Checker-side restatement of the PRODUCT's operand format fgvc_split_f16f6p (not a reference format: the reference computes
# <k, q> in f32, local_attention.py:331-371).  Byte-exact twin of fgvc_amd/csrc/pair_topk_v7.hpp `split_f16f6p_kernel`, and the
# arithmetic model of fgvc_pair_topk_f16f6's scores: 2^16 <k, q> = sum h_k h_q + 2^-8 (sum h6_k l6_q + sum l6_k h6_q).
# ----------------------------------------------------------------------------
def _e2m3_codes(y: np.ndarray) -> np.ndarray:
    """|y| <= 7.5 (float32) -> e2m3 codes (sign << 5 | 5 magnitude bits), round to nearest even, as the kernel's p6_code."""
    a = np.abs(y).astype(np.float32)
    inv = np.where(a < 2, np.float32(8), np.where(a < 4, np.float32(4), np.float32(2))).astype(np.float32)
    r = np.minimum(np.rint(a * inv) / inv, np.float32(7.5)).astype(np.float32)
    c = np.where(r < 2, 8 * r, np.where(r < 4, 8 + 4 * r, 16 + 2 * r)).astype(np.uint32)
    return c | np.where(y < 0, np.uint32(32), np.uint32(0))


def _e2m3_value(c: np.ndarray) -> np.ndarray:
    c = c.astype(np.int64)
    ex, m = (c >> 3) & 3, c & 7
    v = np.where(ex > 0, (1.0 + m / 8.0) * 2.0 ** (ex - 1), m / 8.0)
    return np.where((c >> 5) & 1, -v, v)


def _f6_scale_exp(m: np.ndarray) -> np.ndarray:
    """E8M0 exponent s with max / 2^s <= 7.5, the kernel's p6_scale_exp in float32 (an all-zero block: -40)."""
    m = m.astype(np.float32)
    f, e = np.frexp((m * (np.float32(1.0) / np.float32(7.5))).astype(np.float32))
    s = np.where(f > 0.5, e, e - 1).astype(np.int32)
    s = np.where((m * np.exp2(-s.astype(np.float32))).astype(np.float32) > np.float32(7.5), s + 1, s)
    return np.where(m > 0, np.maximum(s, -40), -40).astype(np.int32)


def f16f6p_channels(v: int, hi: int) -> np.ndarray:
    """the 32 channels of scale block (group v, lane half hi), element e = 8 m + i <-> channel 64 v + 16 m + 8 hi + i"""
    return np.array([64 * v + 16 * m + 8 * hi + i for m in range(4) for i in range(8)])


def f16f6p_encode(x: np.ndarray) -> np.ndarray:
    """(n, 256) float32 L2-normalised rows -> (n, 1024) uint8 rows of fgvc_split_f16f6p."""
    x = np.ascontiguousarray(x, np.float32)
    n = x.shape[0]
    xs = (x * np.float32(256.0)).astype(np.float32)
    h = xs.astype(np.float16)
    hf = h.astype(np.float32)
    lf = ((xs - hf) * np.float32(256.0)).astype(np.float32)
    rows = np.zeros((n, 1024), np.uint8)
    rows[:, :512] = h.view(np.uint8).reshape(n, 512)
    sh6 = (np.arange(32, dtype=np.uint64) * np.uint64(6))
    for v in range(4):
        for hi in range(2):
            ch = f16f6p_channels(v, hi)
            for vals, main, tail, sc in ((hf, 512, 640, 0), (lf, 704, 832, 4)):
                b = vals[:, ch]
                s = _f6_scale_exp(np.abs(b).max(1))
                codes = _e2m3_codes((b * np.exp2(-s.astype(np.float32))[:, None]).astype(np.float32)).astype(np.uint64)
                by = np.zeros((n, 24), np.uint8)                       # the 192-bit little-endian string, 64 bits (10 2/3 codes) at a time
                for k in range(3):
                    lo = np.zeros(n, np.uint64)
                    for e in range(32):
                        bit = 6 * e - 64 * k
                        if -6 < bit < 64:
                            lo |= (codes[:, e] << np.uint64(bit)) if bit >= 0 else (codes[:, e] >> np.uint64(-bit))
                    by[:, 8 * k: 8 * k + 8] = lo.view(np.uint8).reshape(n, 8)
                rows[:, main + 32 * v + 16 * hi: main + 32 * v + 16 * hi + 16] = by[:, :16]
                t0 = tail + 32 * (v >> 1) + 16 * hi + 8 * (v & 1)
                rows[:, t0: t0 + 8] = by[:, 16:]
                rows[:, 896 + 16 * hi + sc + v] = (s + 123).astype(np.uint8)
    return rows


def f16f6p_decode(rows: np.ndarray):
    """(n, 1024) uint8 rows -> (h, h6, l6) float64 (n, 256) in channel order: h = f16(256 x), h6 ~ h, l6 ~ 256 (256 x - h) as the
    matrix instruction sees them (FP6 code x 2^s)."""
    rows = np.ascontiguousarray(rows, np.uint8)
    n = rows.shape[0]
    h = rows[:, :512].copy().view(np.float16).astype(np.float64).reshape(n, 256)
    h6, l6 = np.zeros((n, 256)), np.zeros((n, 256))
    for v in range(4):
        for hi in range(2):
            ch = f16f6p_channels(v, hi)
            for out, main, tail, sc in ((h6, 512, 640, 0), (l6, 704, 832, 4)):
                t0 = tail + 32 * (v >> 1) + 16 * hi + 8 * (v & 1)
                by = np.concatenate([rows[:, main + 32 * v + 16 * hi: main + 32 * v + 16 * hi + 16], rows[:, t0: t0 + 8]], 1)
                w = np.ascontiguousarray(by).view(np.uint64).reshape(n, 3)
                codes = np.zeros((n, 32), np.int64)
                for e in range(32):
                    k, bit = (6 * e) // 64, (6 * e) % 64
                    c = w[:, k] >> np.uint64(bit)
                    if bit > 58:
                        c = c | (w[:, k + 1] << np.uint64(64 - bit))
                    codes[:, e] = (c & np.uint64(63)).astype(np.int64)
                s = rows[:, 896 + 16 * hi + sc + v].astype(np.int64) - 123
                out[:, ch] = _e2m3_value(codes) * np.exp2(s.astype(np.float64))[:, None]
    return h, h6, l6


def _pack_e2m3_24(codes: np.ndarray) -> np.ndarray:
    """(n, 32) e2m3 codes -> (n, 24) uint8: element e in bits [6 e, 6 e + 6) of the 192-bit little-endian string."""
    c = codes.astype(np.uint32).reshape(codes.shape[0], 8, 4)
    w = c[:, :, 0] | (c[:, :, 1] << 6) | (c[:, :, 2] << 12) | (c[:, :, 3] << 18)
    return np.stack([w & 255, (w >> 8) & 255, (w >> 16) & 255], -1).reshape(codes.shape[0], 24).astype(np.uint8)


def act_f16f6_rows(x: np.ndarray, scale_log2: int) -> np.ndarray:
    """Model of the encoder's f16 + FP6 activation rows (FGVC_ACT_F16F6; fgvc_amd/csrc/common.hpp: split_f16f6_chunk) -- (n, 32) float32
    values of one 32-channel chunk -> (n, 128) uint8: [h = f16(s x) 64 B | l6 main 16 B | h6 main 16 B | l6 tail 8 B, scale byte, 0 x 7 |
    h6 tail 8 B, scale byte, 0 x 7] with l6 = e2m3(f16(2^11 (s x - h)) / 2^sl) (byte 127 + sl - 11), h6 = e2m3(h / 2^sh) (byte 127 + sh),
    2^s the smallest power of two that keeps the block's largest magnitude at or below 7.5 (>= 2^-95 for an all-zero block).  The
    sign bit of a code whose magnitude rounds to zero follows the value's sign, as the hardware conversion leaves it."""
    x = np.ascontiguousarray(x, np.float32)
    n = x.shape[0]
    xs = np.clip((x * np.float32(2.0 ** scale_log2)).astype(np.float32), -65504, 65504).astype(np.float32)
    h = xs.astype(np.float16)
    hf = h.astype(np.float32)
    l16 = ((xs - hf) * np.float32(2048.0)).astype(np.float32).astype(np.float16).astype(np.float32)
    rows = np.zeros((n, 128), np.uint8)
    rows[:, :64] = h.view(np.uint8).reshape(n, 64)
    for vals, main, tail, sub in ((l16, 64, 96, 11), (hf, 80, 112, 0)):
        m = np.abs(vals).max(1)
        s = np.maximum(_f6_scale_exp(m), 32 - 127)
        s = np.where(m > 0, s, 32 - 127)
        by = _pack_e2m3_24(_e2m3_codes((vals * np.exp2(-s.astype(np.float32))[:, None]).astype(np.float32)))
        rows[:, main: main + 16] = by[:, :16]
        rows[:, tail: tail + 8] = by[:, 16:]
        rows[:, tail + 8] = (s + 127 - sub).astype(np.uint8)
    return rows


def act_f16f6_decode(rows: np.ndarray):
    """(n, 128) uint8 rows of act_f16f6_rows -> (h, h6, l6) float64 (n, 32): h = f16(s x), h6 ~ h, l6 ~ s x - h as the matrix
    instruction sees them."""
    rows = np.ascontiguousarray(rows, np.uint8)
    n = rows.shape[0]
    h = rows[:, :64].copy().view(np.float16).astype(np.float64).reshape(n, 32)
    out = []
    for main, tail in ((80, 112), (64, 96)):
        by = np.concatenate([rows[:, main: main + 16], rows[:, tail: tail + 8]], 1).astype(np.uint32).reshape(n, 8, 3)
        w = by[:, :, 0] | (by[:, :, 1] << 8) | (by[:, :, 2] << 16)
        codes = np.stack([(w >> (6 * i)) & 63 for i in range(4)], -1).reshape(n, 32)
        out.append(_e2m3_value(codes) * np.exp2(rows[:, tail + 8].astype(np.float64) - 127.0)[:, None])
    return h, out[0], out[1]


def f16f6_cosines(qrows: np.ndarray, krows: np.ndarray) -> np.ndarray:
    """the cosines fgvc_pair_topk_f16f6 computes from two sets of rows, in float64: (nk, nq)"""
    hq, h6q, l6q = f16f6p_decode(qrows)
    hk, h6k, l6k = f16f6p_decode(krows)
    return (hk @ hq.T + (h6k @ l6q.T + l6k @ h6q.T) / 256.0) / 65536.0


# ----------------------------------------------------------------------------
# tolerance-aware comparison of a top-k result against a dense score slab
# ----------------------------------------------------------------------------
def check_topk(dense: torch.Tensor, idx: torch.Tensor, score: torch.Tensor, k: int,
               tol: float = 1e-3, gap: float = 1e-5) -> dict:
    """Validate (idx, score) (S,k) against the dense slab `dense` (M,S) (ideally float64).

    * every reported score is within `tol` of dense[idx]                       (score parity)
    * every selected index is a legitimate top-k member: dense[idx] >= kth_dense - tol
    * on columns whose ranks 1..k+1 are separated by more than `gap` in `dense`,
      idx must equal the canonical dense top-k EXACTLY (bit-exact index parity)
    Returns counts; raises AssertionError on violation.
    """
    M, S = dense.shape
    kk = min(k + 1, M)
    dv, di = topk_canonical(dense, kk)
    dv, di = dv.t(), di.t()                                       # (S,kk)
    got = dense.t().gather(1, idx.clamp_min(0).long())            # (S,k)
    finite = torch.isfinite(dv[:, :k])
    err = (got - score.to(dense.dtype)).abs()
    err = torch.where(finite & torch.isfinite(got), err, torch.zeros_like(err))
    assert float(err.max()) <= tol, f"score error {float(err.max())} > {tol}"
    kth = dv[:, k - 1:k]
    legit = (got >= kth - tol) | ~finite
    assert bool(legit.all()), "an index outside the tolerance top-k set was selected"
    gaps = (dv[:, :-1] - dv[:, 1:]).abs()
    gaps = torch.where(torch.isfinite(gaps), gaps, torch.full_like(gaps, float("inf")))
    clear = (gaps.min(dim=1).values > gap) & finite.all(1)
    exact = (idx.long() == di[:, :k]).all(1)
    assert bool(exact[clear].all()), f"{int((~exact[clear]).sum())} clear-gap queries differ in index"
    return dict(queries=S, clear=int(clear.sum()), exact=int(exact.sum()), max_score_err=float(err.max()))
