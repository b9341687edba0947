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
    "compute_affinity", "propagate", "non_local_attention", "local_square_attention", "coords_grid", "cat", "video2images",
    "images2video", "bilinear_sample",
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


def _temperature(sim_mode, temperature, channels):
    """Divisor of the raw dot product: `temperature` (local_attention.py:321-323), or sqrt(C)/2 for 'l2-distance' (:324-327:
    (2 k.q - 1) / sqrt(C), `temperature` is not read on that branch)."""
    return (channels ** 0.5) / 2.0 if sim_mode == "l2-distance" else temperature


def _logit_shift(sim_mode, channels, normalize=True):
    """What 'l2-distance' subtracts from every logit beside the scaled dot product: |k|^2 / sqrt(C) = 1 / sqrt(C) for unit keys
    (un-normalised keys carry their own |k|^2 inside the product: _l2_augment)."""
    return 1.0 / (channels ** 0.5) if (sim_mode == "l2-distance" and normalize) else 0.0


def _l2_augment(query, key):
    """'l2-distance' on UN-normalised features, (2 k.q - |k|^2) / sqrt(C) (local_attention.py:324-327), as a plain dot product at
    temperature sqrt(C) / 2: one more channel, 1 on the query side and -|k|^2 / 2 on the key side.  (With normalised features the
    extra term is the same for every key and this is not needed.)"""
    q1 = torch.cat([query, torch.ones_like(query[:, :1])], 1)
    k1 = torch.cat([key, -0.5 * key.float().pow(2).sum(1, keepdim=True).to(key.dtype)], 1)
    return q1, k1


def _attention(query, key, value, spec: MaskSpec, dense_mask, temperature, topk, normalize, non_mask_len, mode, logit_shift=0.0):
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
        if logit_shift != 0.0 and mode != "softmax":
            raise NotImplementedError("fgvc_amd: sim_mode='l2-distance' with mode='cosine' needs topk (the dense form applies no shift)")
        out = ops.dense_attend(qf[0], kf, labels, Hq, Wq, Hk, Wk, spec, temperature, mode, non_mask_len if any_mask else T,
                               dense_mask)
        return out.t().reshape(1, P, Hq, Wq).to(query.dtype)
    pairs = ops.make_pairs([(0, t, any_mask and t >= non_mask_len) for t in range(T)], query.device)
    pidx, pscore = ops.pair_topk_auto(qf, kf, pairs, Hq, Wq, Hk, Wk, spec, topk, normalized=bool(normalize),
                                      validate=False, dense_mask=dense_mask, all_masked=any_mask and non_mask_len == 0)
    slot_pair = torch.arange(T, dtype=torch.int32, device=query.device).view(1, T)
    idx, logit, weight = ops.merge_topk(pidx, pscore, slot_pair, Hk * Wk, topk, temperature, mode, validate=False)
    if logit_shift != 0.0 and mode == "cosine":
        # 'l2-distance': the logit is the scaled dot product MINUS |k|^2 / sqrt(C); clamp(logit, 0)^2 sees that shift (:370-371).
        # Empty list entries carry -inf: weight 0 either way.
        weight = (logit - logit_shift).clamp_(min=0).square_()
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
    C = query.shape[1]
    if sim_mode == "l2-distance" and not normalize:
        if key.ndim == 4:
            key, value = key.unsqueeze(2), value.unsqueeze(2)
        query, key = _l2_augment(query, key)
    return _attention(query, key, value, spec, dense, _temperature(sim_mode, temperature, C), topk, normalize,
                      non_mask_len, mode, _logit_shift(sim_mode, C, normalize))


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


def propagate(img, affinity, topk=None):
    """affinity_utils.py:33-50: new_img[b] = img[b] (C x HW) @ affinity[b] (HW_src x HW_dst) for a GIVEN dense affinity (what
    compute_affinity returned); `topk`: per destination column subtract the k-th largest entry, clamp at 0, normalise by the sum
    (:36-44).  The affinity is streamed once by fgvc_dense_propagate_f32 (+ once by fgvc_dense_kth_f32 for the thresholds).
    The reference rewrites the caller's `affinity` in place on the topk branch ("to save memory"); this one leaves it untouched."""
    batches, channels, height, width = img.size()
    assert affinity.shape == (batches, height * width, height * width), "propagate: affinity must be (N, HW, HW)"
    outs = []
    for b in range(batches):
        labels = img[b].reshape(channels, height * width).t().float().contiguous()
        out = ops.dense_propagate(affinity[b].float().contiguous(), labels, topk)
        outs.append(out.t().reshape(channels, height, width))
    return torch.stack(outs, 0).to(img.dtype)


def non_local_attention(tar, refs, per_ref=True, flatten=True, temprature=1.0, mask=None, scaling=False, norm=False,
                        att_only=False, mode="dot"):
    """correlation.py:32-83: att[b, t, i, j] = <tar_i, ref_{t,j}> / temprature (sic), the dense volume per reference frame
    (fgvc_corr_volume_f32 with the roles swapped: the volume kernel writes [key][query], here key = target pixel i).
    Returns what the reference returns: `att` if att_only, else (batch size, softmaxed attention) -- its first return value is
    the variable `_` that `_, t, feat_dim, w_, h_ = refs.shape` bound (:46), i.e. the batch size; the transformed frames
    it computes are dropped there too.  mode='l2' is refused: the reference's own `.view` of a transposed tensor (:62)
    fails for more than one reference frame."""
    if isinstance(refs, (list, tuple)):
        refs = torch.stack(list(refs), 1)
    if mode != "dot":
        raise NotImplementedError("fgvc_amd non_local_attention: mode='dot' only (the reference's 'l2' branch cannot run for t > 1)")
    B, t, feat_dim = refs.shape[:3]
    atts = []
    for b in range(B):
        tf = ops.normalize_to_hwc(tar[b:b + 1].float(), bool(norm), pad=True)[0]                     # (HW_tar, C')
        rf = ops.normalize_to_hwc(refs[b].float().contiguous(), bool(norm), pad=True)                # (t, HW_ref, C')
        atts.append(torch.stack([ops.corr_volume(rf[k], tf, temprature, "f32") for k in range(t)], 0))   # [t][tar i][ref j]
    att = torch.stack(atts, 0)
    if scaling:
        att = att / torch.sqrt(torch.tensor(feat_dim).float()).to(att.device)
    if mask is not None:
        att.masked_fill_(~(mask.dense() if isinstance(mask, NeighborMask) else mask.bool()), float("-inf"))
    if att_only:
        return att.to(tar.dtype)
    if per_ref:
        return B, torch.softmax(att, dim=-1).to(tar.dtype)
    return B, torch.softmax(att.permute(0, 2, 1, 3).flatten(2), -1).to(tar.dtype)


def local_square_attention(query, key, value, kernel_size, temperature=1, topk=None, batch_as_context=False):
    """local_attention.py:38-103: zero-padded (kh x kw) window around every query pixel (F.unfold, padding k // 2), RAW dot products
    / temperature (no normalisation, no softmax: the attention values themselves weight the values), optionally only the `topk`
    largest.  batch_as_context: the windows of ALL key batch entries form one context (the reference's gather then needs a query
    batch of 1).  Kernels: fgvc_local_corr_topk_f32 + fgvc_propagate_topk_f32 with the logits as weights (topk), or the
    box-masked dense pass fgvc_dense_attend_f32 in raw mode (topk=None: zero-padded taps contribute 0)."""
    assert query.ndim == key.ndim == 4
    assert query.shape[1:] == key.shape[1:]
    assert value.shape[2:] == key.shape[2:], f"{value.shape} {key.shape}"
    assert value.shape[0] == key.shape[0]
    ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
    if ks[0] % 2 == 0 or ks[1] % 2 == 0:
        raise ValueError("local_square_attention: even kernel sizes do not survive the reference's own reshape (F.unfold yields (H+1)(W+1) columns)")
    C, H, W = query.shape[1:]
    P = value.shape[1]
    Ry, Rx = ks[0] // 2, ks[1] // 2
    if batch_as_context:
        assert query.shape[0] == 1, "batch_as_context: the reference's gather needs a query batch of 1"
        groups = [(0, list(range(key.shape[0])))]
    else:
        assert query.shape[0] == key.shape[0]
        groups = [(b, [b]) for b in range(query.shape[0])]
    outs = []
    for qb, kbs in groups:
        qf = ops.normalize_to_hwc(query[qb:qb + 1].float(), False, pad=True)                        # raw rows, zero-padded channels
        kf = ops.normalize_to_hwc(key[kbs].float().contiguous(), False, pad=True)                   # (K, HW, C')
        labels = value[kbs].permute(0, 2, 3, 1).reshape(len(kbs), H * W, P).float().contiguous()
        if topk is None:
            out = ops.dense_attend(qf[0], kf, labels, H, W, H, W, MaskSpec(ry=Ry, rx=Rx), temperature, "raw")
        else:
            if Ry != Rx:
                raise NotImplementedError("fgvc_amd local_square_attention: topk with a non-square window")
            idx, logit, _ = ops.local_corr_topk(qf, kf, H, W, Ry, topk, temperature, normalized=False)
            out = ops.propagate_topk(labels, torch.arange(len(kbs), dtype=torch.int32, device=query.device), idx, logit, H, W, H, W,
                                     window_L=2 * Ry + 1)
        outs.append(out.t().reshape(P, H, W))
    return torch.stack(outs, 0).to(query.dtype)


def masked_attention_efficient_c2f(query, key, query_fine, key_fine, value, mask, temperature=1, topk=None,
                                   normalize=True, step=32, non_mask_len=0, mode="softmax",
                                   sim_mode="dot_product", radius_fine=12):
    """local_attention.py:721-880.  Coarse stage = fgvc_pair_topk_f16x3 / fgvc_pair_topk_f32 with topk=1 per key slot (arg-max of
    the per-frame softmax, :835-837); fine stage = fgvc_c2f_refine_f32."""
    _check_common(query, key, value, mode, "dot_product", normalize)    # `sim_mode` is accepted and never read there (:733, :805, :847)
    if topk is None:
        # the reference's own branch is `pass` followed by `output[...] = cur_output` with cur_output unbound (:866-870): it raises
        raise NotImplementedError("masked_attention_efficient_c2f: topk=None cannot run in the reference either (local_attention.py:866-870)")
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
    out, _, _ = ops.c2f_refine(coarse, qfine, kfine, vfine, H, W, scale, radius_fine, topk, temperature, mode)
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
