"""`mmpt.models.common` operators with the reference's names, argument meaning and error behaviour,
executed by libfgvc_hip.so.  Tensors are NCHW like the reference's; layout changes happen here.

Reference signatures: mmpt/models/common/local_attention.py:20,191,267,392,721,1117;
affinity_utils.py:6,75; part of utils.py:59-78,202.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import ops
from ..ops import MaskSpec

__all__ = [
    "NeighborMask", "spatial_neighbor", "masked_attention_efficient", "masked_attention_efficient_v2",
    "masked_attention", "masked_attention_efficient_c2f", "masked_attention_efficient_correlation_v2",
    "compute_affinity", "coords_grid", "cat", "video2images", "images2video", "bilinear_sample",
]


class NeighborMask:
    """What `spatial_neighbor` returns here: the PREDICATE, not the (HW x HW) bool tensor
    (268 MB at 128x128, 659 MB at 120x214 -- affinity_utils.py:75-112).  The kernels evaluate the
    predicate analytically.  `.dense()` materialises the reference's tensor when something else needs it.
    """
    ndim = 2

    def __init__(self, height: int, width: int, neighbor_range, mode: str, device):
        self.height, self.width, self.neighbor_range, self.mode, self.device = height, width, neighbor_range, mode, device

    @property
    def shape(self):
        return (self.height * self.width, self.height * self.width)

    @property
    def spec(self) -> MaskSpec:
        return MaskSpec.from_neighbor_range(self.neighbor_range, self.mode)

    def dense(self) -> torch.Tensor:
        H, W = self.height, self.width
        ys = torch.arange(H, device=self.device).view(H, 1).expand(H, W).reshape(-1)
        xs = torch.arange(W, device=self.device).view(1, W).expand(H, W).reshape(-1)
        dy, dx = ys.view(-1, 1) - ys.view(1, -1), xs.view(-1, 1) - xs.view(1, -1)
        s = self.spec
        return (dy * dy + dx * dx <= s.r2max) & (dy.abs() <= s.ry) & (dx.abs() <= s.rx)

    def bool(self):
        return self.dense()


def spatial_neighbor(batches, height, width, neighbor_range, device, dtype, dim=1, mode="circle"):
    """affinity_utils.py:75-82 signature.  Returns a NeighborMask (see above)."""
    assert dim in [1, 2]
    assert mode in ["circle", "square"]
    return NeighborMask(height, width, neighbor_range, mode, device)


def cat(tensors: List[torch.Tensor], dim: int = 0):
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


def coords_grid(batch: int, xx, yy):
    """(batch, 2, H, W), channel 0 = x, channel 1 = y (local_attention.py:20-35)."""
    gy, gx = torch.meshgrid(yy, xx, indexing="ij")
    return torch.stack([gx, gy], 0).float()[None].repeat(batch, 1, 1, 1)


def bilinear_sample(feat, grid, mode="bilinear", padding_mode="zeros", align_corners=False, scale=True):
    """corr_lookup.py:31-65: F.grid_sample at pixel coordinates (x,y); `scale` maps them to [-1,1]
    (the reference rescales the caller's grid IN PLACE; this one works on a copy)."""
    import torch.nn.functional as F
    H, W = feat.shape[-2:]
    if grid.shape[-1] != 2:
        grid = grid.permute(0, 2, 3, 1)
    if scale:
        grid = torch.stack([grid[..., 0] * 2.0 / max(W - 1, 1) - 1.0, grid[..., 1] * 2.0 / max(H - 1, 1) - 1.0], -1)
    return F.grid_sample(feat, grid, mode, padding_mode, align_corners)


def video2images(imgs):
    """(N,C,T,H,W) -> (N*T,C,H,W)  (utils.py:59-67)."""
    n, c, t = imgs.shape[:3]
    return imgs.transpose(1, 2).reshape(n * t, c, *imgs.shape[3:])


def images2video(imgs, clip_len):
    """(N*T,C,H,W) -> (N,C,T,H,W)  (utils.py:70-78)."""
    nt, c = imgs.shape[:2]
    return imgs.reshape(nt // clip_len, clip_len, c, *imgs.shape[2:]).transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------------------------
def _check_common(query, key, value, mode, sim_mode, normalize=True):
    assert mode in ["softmax", "cosine"]
    assert query.size(0) == key.size(0) == value.size(0)
    if query.size(0) != 1:
        # the reference's index_select is only correct for N == 1 (local_attention.py:360-362,
        # enforced upstream by vanilla_tracker.py:134)
        raise NotImplementedError("fgvc_amd: batch size must be 1 (as in the reference's tracker)")
    if sim_mode not in ("dot_product", "l2-distance"):
        raise NotImplementedError(f"fgvc_amd: sim_mode={sim_mode!r} (the reference knows 'dot_product' and 'l2-distance')")
    if sim_mode == "l2-distance" and (not normalize or mode != "softmax"):
        # (2 k.q - |k|^2) / sqrt(C): with |k| = 1 the -|k|^2 term shifts every logit alike -- same ranking as the dot product, and
        # the softmax does not see the shift.  Un-normalised keys or the (shift-sensitive) cosine weights need another kernel.
        raise NotImplementedError("fgvc_amd: sim_mode='l2-distance' is on the accelerated path for normalize=True, mode='softmax'")


def _temperature(sim_mode, temperature, channels):
    """Divisor of the raw dot product: `temperature` (local_attention.py:321-323), or sqrt(C)/2 for 'l2-distance' (:324-327:
    (2 k.q - 1) / sqrt(C), `temperature` is not read on that branch)."""
    return (channels ** 0.5) / 2.0 if sim_mode == "l2-distance" else temperature


def _attention(query, key, value, spec: MaskSpec, dense_mask, temperature, topk, normalize, non_mask_len, mode):
    if key.ndim == 4:
        key, value = key.unsqueeze(2), value.unsqueeze(2)
    assert value.ndim == key.ndim == 5
    assert value.shape[2:] == key.shape[2:], f"{value.shape} {key.shape}"
    T = key.size(2)
    assert 0 <= non_mask_len < T
    C, Hq, Wq = query.shape[1:]
    Hk, Wk = key.shape[3:]
    P = value.size(1)
    same = (Hq, Wq) == (Hk, Wk)
    qf = ops.normalize_to_hwc(query.float(), normalize, pad=True)                                   # (1,HWq,C')
    kf = ops.normalize_to_hwc(key[0].transpose(0, 1).float().contiguous(), normalize, pad=True)    # (T,HWk,C')
    any_mask = dense_mask is not None or not spec.is_none
    if any_mask:
        assert same or dense_mask is not None
    labels = value[0].permute(1, 2, 3, 0).reshape(T, Hk * Wk, P).float().contiguous()
    if topk is None:        # weights over every unmasked key (local_attention.py:376-383)
        out = ops.dense_attend(qf[0], kf, labels, Hq, Wq, Hk, Wk, spec, temperature, mode, non_mask_len if any_mask else T,
                               dense_mask)
        return out.t().reshape(1, P, Hq, Wq).to(query.dtype)
    pairs = ops.make_pairs([(0, t, any_mask and t >= non_mask_len) for t in range(T)], query.device)
    pidx, pscore = ops.pair_topk_auto(qf, kf, pairs, Hq, Wq, Hk, Wk, spec, topk, normalized=bool(normalize),
                                      validate=False, dense_mask=dense_mask, all_masked=any_mask and non_mask_len == 0)
    slot_pair = torch.arange(T, dtype=torch.int32, device=query.device).view(1, T)
    idx, _, weight = ops.merge_topk(pidx, pscore, slot_pair, Hk * Wk, topk, temperature, mode, validate=False)
    out = ops.propagate_topk(labels, torch.arange(T, dtype=torch.int32, device=query.device), idx[0], weight[0],
                             Hq, Wq, Hk, Wk)
    return out.t().reshape(1, P, Hq, Wq).to(query.dtype)


def masked_attention_efficient(query, key, value, mask, temperature=1, topk=None, normalize=True, step=32,
                               non_mask_len=0, mode="softmax", sim_mode="dot_product"):
    """local_attention.py:267-389.  `step` is accepted and ignored: nothing is chunked because the
    (T*HW x step) slab never exists.  `mask`: None, a NeighborMask from spatial_neighbor(), or any
    (HkWk, HqWq) tensor."""
    _check_common(query, key, value, mode, sim_mode, normalize)
    spec, dense = MaskSpec.none(), None
    if isinstance(mask, NeighborMask):
        spec = mask.spec
    elif mask is not None:
        hk, wk = key.shape[-2:]
        assert tuple(mask.shape[-2:]) == (hk * wk, query.shape[2] * query.shape[3])
        dense = mask.reshape(hk * wk, -1).bool()
    return _attention(query, key, value, spec, dense, _temperature(sim_mode, temperature, query.shape[1]), topk, normalize,
                      non_mask_len, mode)


def masked_attention_efficient_v2(query, key, value, radius, temperature=1, topk=None, normalize=True, step=32,
                                  non_mask_len=0, mode="softmax", sim_mode="dot_product"):
    """local_attention.py:392-508: the disc `dist < radius` rebuilt per chunk there, analytic here.
    (The reference ignores non_mask_len and sim_mode in this variant, :453-470; so do we.)"""
    _check_common(query, key, value, mode, "dot_product", normalize)    # `sim_mode` is accepted and never read there (:453-455)
    return _attention(query, key, value, MaskSpec.circle(radius), None, temperature, topk, normalize, 0, mode)


def masked_attention(query, key, value, mask, temperature=1, topk=None, normalize=True, step=100):
    """local_attention.py:191-264 (whole-volume formulation): identical result to the efficient form."""
    return masked_attention_efficient(query, key, value, mask, temperature, topk, normalize, step)


def compute_affinity(src_img, dst_img, temperature=1.0, normalize=True, softmax_dim=None, mask=None,
                     precision: str = "f32"):
    """affinity_utils.py:6-30: (N, HWsrc, HWdst) dense affinity = the materialised correlation volume.
    `precision` ('f32' | 'f16f6' | 'f16f8' | 'bf16x3' | 'bf16') is an extension selecting the MFMA arithmetic ('f16f6' / 'f16f8':
    C == 256 and normalize=True; 'f16f6' is the fastest form inside the 1e-3 score bar)."""
    n = src_img.shape[0]
    outs = []
    for b in range(n):
        sf = ops.normalize_to_hwc(src_img[b:b + 1].float(), normalize, pad=True)[0]
        df = ops.normalize_to_hwc(dst_img[b:b + 1].float(), normalize, pad=True)[0]
        if precision in ("f16f8", "f16f6"):
            if sf.shape[-1] != 256 or not normalize:
                raise NotImplementedError(f"{precision} volume needs C == 256 and normalize=True")
            split = ops.split_f16f8 if precision == "f16f8" else ops.split_f16f6
            sf, df = split(sf), split(df)
        elif precision != "f32":
            if sf.shape[-1] % 64:
                raise NotImplementedError("bf16 volume needs C % 64 == 0")
            sf, df = ops.split_bf16(sf), ops.split_bf16(df)
        outs.append(ops.corr_volume(df, sf, temperature, precision))        # [src j][dst i]
    aff = torch.stack(outs, 0)
    if mask is not None:
        m = mask.dense() if isinstance(mask, NeighborMask) else mask.bool()
        aff.masked_fill_(~m, float("-inf"))
    if softmax_dim is not None:
        aff = aff.softmax(dim=softmax_dim)
    if mask is not None:
        aff[aff.isnan()] = 0
    return aff


def masked_attention_efficient_c2f(query, key, query_fine, key_fine, value, mask, temperature=1, topk=None,
                                   normalize=True, step=32, non_mask_len=0, mode="softmax",
                                   sim_mode="dot_product", radius_fine=12):
    """local_attention.py:721-880.  Coarse stage = fgvc_pair_topk_bf16x4 / fgvc_pair_topk_f32 with topk=1 per key slot (arg-max of
    the per-frame softmax, :835-837); fine stage = fgvc_c2f_refine_f32."""
    _check_common(query, key, value, mode, sim_mode, normalize)
    if mode != "softmax" or sim_mode != "dot_product" or topk is None:
        raise NotImplementedError("fgvc_amd c2f: mode='softmax', sim_mode='dot_product' and an integer topk only")
    if key.ndim == 4:
        key, value = key.unsqueeze(2), value.unsqueeze(2)
        key_fine = key_fine.unsqueeze(2) if key_fine.ndim == 4 else key_fine
    T = key.size(2)
    C, H, W = query.shape[1:]
    assert key.shape[3:] == (H, W), "c2f needs equal coarse query/key grids"
    scale = key_fine.shape[3] // H                                                                # :769
    P = value.size(1)
    spec = mask.spec if isinstance(mask, NeighborMask) else MaskSpec.none()
    dense = None if (mask is None or isinstance(mask, NeighborMask)) else mask.reshape(H * W, H * W).bool()
    any_mask = dense is not None or not spec.is_none
    dev = query.device
    qf = ops.normalize_to_hwc(query.float(), normalize, pad=True)
    kf = ops.normalize_to_hwc(key[0].transpose(0, 1).float().contiguous(), normalize, pad=True)
    pairs = ops.make_pairs([(0, t, any_mask and t >= non_mask_len) for t in range(T)], dev)
    cidx, _ = ops.pair_topk_auto(qf, kf, pairs, H, W, H, W, spec, 1, normalized=bool(normalize), validate=False,
                                 dense_mask=dense, all_masked=any_mask and non_mask_len == 0)          # C = 256, normalised: the bf16-pipe kernel; else fgvc_pair_topk_f32
    coarse = cidx[:, :, 0].clamp_min(0).contiguous()
    qfine = ops.normalize_to_hwc(query_fine.float(), normalize)[0]
    kfine = ops.normalize_to_hwc(key_fine[0].transpose(0, 1).float().contiguous(), normalize)
    vfine = value[0].permute(1, 2, 3, 0).reshape(T, -1, P).float().contiguous()
    out, _, _ = ops.c2f_refine(coarse, qfine, kfine, vfine, H, W, scale, radius_fine, topk, temperature)
    return out.t().reshape(1, P, H, W).to(query.dtype)


def masked_attention_efficient_correlation_v2(query_frame, key_frames, value, radius, corr_infer, feat_extractor,
                                              temperature=1, topk=None, normalize=True, sstep=32, tstep=5):
    """local_attention.py:1117-1250: local (2R+1)^2 window, zero padded, top-k over K*(2R+1)^2, temperature
    applied after the top-k.  `corr_infer`, `sstep`, `tstep` are accepted and unused (no chunking)."""
    assert query_frame.size(0) == 1
    query = feat_extractor(query_frame)
    K = key_frames.size(2)
    keys = feat_extractor(key_frames[:, :, :].transpose(1, 2).flatten(0, 1))
    C, H, W = query.shape[1:]
    P = value.size(1)
    qf = ops.normalize_to_hwc(query.float(), normalize, pad=True)
    kf = ops.normalize_to_hwc(keys.float().contiguous(), normalize, pad=True)
    idx, _, weight = ops.local_corr_topk(qf, kf, H, W, radius, topk, temperature, normalized=bool(normalize))
    labels = value[0].permute(1, 2, 3, 0).reshape(K, H * W, P).float().contiguous()
    out = ops.propagate_topk(labels, torch.arange(K, dtype=torch.int32, device=query.device), idx, weight, H, W, H, W,
                             window_L=2 * radius + 1)
    return out.t().reshape(1, P, H, W).to(query.dtype)
