"""Trackers behind the reference's registry names and call contract
(mmpt/models/trackers/base.py:24-60, vanilla_tracker.py:25-412, :417-585).

`model(test_mode=True, rgbs=..., query_points=..., trajectories=..., visibilities=...)` returns the
reference's 5-tuple.  Everything after the encoder runs in libfgvc_hip.so through fgvc_amd.engine:
no feature map, label map or score slab visits the host (the reference parks features on the CPU
and re-uploads them per frame, :145-147, :347-352, and ships T*P*h*w floats back for a numpy
argsort, :404-406).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import engine, ops
from .builder import build_backbone, build_components
from .config import ConfigDict
from .registry import MODELS


class EncoderOverflow(RuntimeError):
    """An activation left the f16 range of its calibrated scale (ResNet.check_overflow): the pass that raised it is invalid."""


class BaseModel(nn.Module):
    """base.py:24-60: holds train_cfg/test_cfg, dispatches on test_mode."""

    def __init__(self, train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg if test_cfg is not None else ConfigDict()

    def init_weights(self):
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()

    def forward_train(self, *a, **k):
        raise NotImplementedError("fgvc_amd accelerates the inference path only")

    def forward_test(self, *a, **k):
        raise NotImplementedError

    def forward(self, test_mode=False, **kwargs):
        if not test_mode:
            return self.forward_train(**kwargs)
        try:
            return self.forward_test(**kwargs)
        except EncoderOverflow:
            # The f16 arithmetic of the encoder stores activations at per-tensor scales derived from the weights (canonical frames, 2^8 of
            # headroom); a video with larger activations than that overflows them.  The check that raised has dropped the scales and
            # asked for 2^4 more headroom: run the video once more (up to three times: 2^22 in all).  The video AFTER it starts from the
            # canonical scales again -- no result depends on the videos before it.  Results are never returned from an overflowed pass.
            bb = getattr(self, "backbone", None)
            try:
                for attempt in range(3):
                    self.overflow_retries = getattr(self, "overflow_retries", 0) + 1
                    try:
                        return self.forward_test(**kwargs)
                    except EncoderOverflow:
                        if attempt == 2 or getattr(bb, "calibration", None) != "canonical":
                            raise
            finally:
                if hasattr(bb, "end_overflow_retry"):
                    bb.end_overflow_retry()


@MODELS.register_module()
class BaseTracker(BaseModel):
    def __init__(self, backbone, head=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.backbone = build_backbone(backbone)
        self.head = build_components(head) if head is not None else None
        self.register_buffer("iteration", torch.tensor(0, dtype=torch.float))

    def extract_feat(self, imgs):
        x = self.backbone(imgs)
        if self.head is not None:
            x = self.head(x)
        return x


@MODELS.register_module()
class VanillaTracker(BaseTracker):
    """Label propagation with a dense (disc-masked) affinity and top-k softmax."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        g = self.test_cfg.get
        self.stride_sample = g("stride_sample", False)
        self.feat_channels = None          # the encoder's (un-padded) channel count, known after the first get_feats_hwc()

    # ---- A1/A2: encoder, every frame exactly once, features stay on the device ------------------
    def extract_feat(self, imgs):
        x = self.backbone(imgs)
        if self.stride_sample:
            x = x[:, :, ::self.stride_sample, ::self.stride_sample]
        if self.head is not None:
            x = self.head(x)
        return x

    @torch.no_grad()
    def get_feats_hwc(self, frames: torch.Tensor, split: bool = False, out: Optional[torch.Tensor] = None):
        """frames (T,3,h,w) -> normalised channels-last (T, HfWf, C'), Hf, Wf.
        batch_step frames per encoder call (vanilla_tracker.py:135-147).  split=True: the bank comes back in the pair kernel's operand
        format -- (T, HfWf, 2, C') int16 in the format engine_config().pair_split_fmt names, or (T, HfWf, 4, 256) int16 = split_f16f6x()
        rows (2 KiB per pixel: the f16 + FP6 row and the exact f32 channels behind it; engine_config().bank_fmt, the default) --
        wherever the engine's split pair kernel applies (one pass less; engine.run_affinity takes either form), f32 otherwise.
        `out`: rows of the caller's own feature bank (T, ...) of the shape / dtype this call produces; the encoder then writes there
        and `out` itself is returned (clip sharding: the frames a rank encodes land in its local bank without a copy).  A mismatch
        falls back to a fresh tensor -- compare the result with `out` by identity."""
        step = int(self.test_cfg.get("batch_step", 5))
        norm = bool(self.test_cfg.get("with_norm", True))
        chunks = []
        in_place = out is not None
        Hf = Wf = None
        if self.test_cfg.get("channels_last", False):
            # MIOpen's fastest f32 kernels on gfx950 are NHWC; feeding NHWC avoids its transposes
            frames = frames.contiguous(memory_format=torch.channels_last)
        fast = (hasattr(self.backbone, "forward_hwc") and self.head is None and not self.stride_sample
                and len(getattr(self.backbone, "out_indices", ())) == 1)
        split_if = None
        split_fmt = "bf16"
        if split and fast:
            cfg = self.engine_config()
            split_fmt = cfg.bank_fmt                 # "f16f6x" (2 KiB rows: + the exact f32 channels) where the refining merge runs
            if cfg.pair_precision in ("auto", "split"):
                split_if = lambda C, H, W: ops.split_path_ok(C, H, W, cfg.topk, cfg.with_norm, None, cfg.mask,
                                                             cfg.with_first_neighbor or not cfg.with_first)
        for i in range(0, frames.shape[0], step):
            if fast:       # backbone writes normalised channels-last rows itself (no NCHW round trip)
                o = out[i:i + step] if out is not None else None
                f, Hf, Wf = self.backbone.forward_hwc(frames[i:i + step], norm, split_if=split_if, split_fmt=split_fmt, out=o)
                self.feat_channels = f.shape[-1]                      # (this path never pads)
                chunks.append(f)
                in_place = in_place and o is not None and f is o
                continue
            f = self.extract_feat(frames[i:i + step])
            if isinstance(f, (tuple, list)):
                f = f[0]
            Hf, Wf = f.shape[-2:]
            self.feat_channels = f.shape[1]                           # before the zero padding to a kernel width
            chunks.append(ops.normalize_to_hwc(f.float(), norm, pad=True))
            in_place = False
        if in_place and chunks:
            return out, Hf, Wf
        return (chunks[0] if len(chunks) == 1 else torch.cat(chunks, 0)), Hf, Wf       # (cat of one tensor is a copy)

    def engine_config(self) -> engine.TrackerConfig:
        """The test_cfg as the engine reads it.  Unless test_cfg names `pair_split_fmt` itself, the pair kernel follows the encoder's
        arithmetic: an f16f8 trunk (features +-3e-5 logit from the reference's) goes with fgvc_pair_topk_f16f6 (+-6e-5, half the matrix
        work) where that kernel applies -- every key slot masked, the mask within 64 key blocks of a query tile --, any other trunk
        (set_arith('f16x3') / 'bf16x3') with the 1e-7-grade fgvc_pair_topk_f16x3."""
        cfg = engine.TrackerConfig.from_test_cfg(self.test_cfg)
        if "pair_split_fmt" not in self.test_cfg and getattr(self.backbone, "arith", None) in ("f16f8", "f16f6"):
            all_masked = cfg.with_first_neighbor or not cfg.with_first
            if all_masked and ops.pair_blocks_reached(cfg.mask) <= ops.PAIR_F16F6_MAX_BLOCKS and cfg.topk <= 10 and cfg.with_norm:
                cfg.pair_split_fmt = "f16f6"
        return cfg

    def _check_kernels(self):
        """Fail closed, per video: the pair kernel's LDS protocol waits with bounded spins, and a workgroup whose wait gave up writes
        poison lists (NaN trajectories downstream) and raises a device flag.  The flag is read -- and cleared -- here, at the point
        where the reference synchronises anyway (`.cpu().numpy()` of the label maps, vanilla_tracker.py:404).  `check_kernels=False`
        in test_cfg (an extension key) skips the device synchronisation; the poison still marks the results."""
        if not self.test_cfg.get("check_kernels", True):
            return
        if ops.pair_f16x3_timed_out():
            raise RuntimeError("fgvc_pair_topk_f16x3: a bounded wait of the kernel's LDS protocol timed out; this video's results are invalid")
        st = getattr(self, "_refine_stats", None)
        if st is not None:
            # the refining merge assumes |fgvc_pair_topk_f16f6 score - exact| <= pair_refine_eps; every candidate it re-scores is a sample
            # of that error, and the largest one seen in this video is held against the bound (measured: 5e-6 of 2e-5)
            self._refine_stats = None
            worst, eps = max(ops.refine_max_error(s_) for s_ in st), float(self.engine_config().pair_refine_eps)
            if worst > eps:
                raise RuntimeError(f"fgvc_merge_refine_topk_f32: the pair kernel's score error reached {worst:.2e} on this video, beyond the bound "
                                   f"{eps:.2e} the exact re-scoring assumes; raise test_cfg.pair_refine_eps or set pair_split_fmt='f16'")
        if hasattr(self.backbone, "check_overflow") and self.backbone.check_overflow():
            raise EncoderOverflow("fgvc_amd ResNet: an activation left the f16 range of its calibrated scale (f16 arithmetic of the encoder); "
                               "this video's results are invalid -- the scales were dropped and the next call re-calibrates on its own "
                               "frames (or call backbone.calibrate(frames), or backbone.set_arith('bf16x3'))")

    # ---- A10: regrouping by query time ----------------------------------------------------------
    @torch.no_grad()
    def forward_test(self, rgbs, query_points, trajectories, visibilities, save_image=False, save_path=None,
                     iteration=None):
        """rgbs (1,T,3,h,w), query_points (1,P,3)=(t,x,y), trajectories (1,T,P,2), visibilities (1,T,P)."""
        if not rgbs.is_cuda:
            raise RuntimeError("fgvc_amd.VanillaTracker runs on the GPU only (no CPU fallback)")
        assert rgbs.shape[0] == 1, "batch size must be 1 (vanilla_tracker.py:134)"
        cfg = self.engine_config()
        T, h, w = rgbs.shape[1], rgbs.shape[-2], rgbs.shape[-1]
        dev = rgbs.device
        qp = query_points[0]
        if not cfg.regroup:
            # single group that starts at frame 0 regardless of the query times (vanilla_tracker.py:302-303)
            feats, Hf, Wf = self.get_feats_hwc(rgbs[0], split=True)
            plan = engine.plan_clip(T, [0], cfg)
            tk = engine.run_affinity(feats, Hf, Wf, plan, cfg, channels=self.feat_channels)
            self._refine_stats = [tk.refine_stats] if tk.refine_stats is not None else None
            _, coords = engine.run_propagation(tk, 0, qp[:, 1:].to(dev, torch.float32), Hf, Wf, h, w, cfg)
            traj_pred = coords.unsqueeze(0)                                   # float64, like torch.from_numpy(...)
            self._check_kernels()
            return trajectories, visibilities, traj_pred, torch.zeros_like(visibilities), query_points
        t_min = int(qp[:, 0].min().item())
        # frames before the earliest query time are never used by any group
        feats, Hf, Wf = self.get_feats_hwc(rgbs[0, t_min:], split=True)
        qp_rel = qp.clone()
        qp_rel[:, 0] -= t_min
        stats = []
        traj, order = engine.track_points(feats, Hf, Wf, h, w, qp_rel, cfg, channels=self.feat_channels, stats_out=stats)   # (T-t_min, P, 2) f64, regrouped
        self._refine_stats = stats or None
        order = order.to(dev)
        traj_pred = torch.zeros_like(trajectories)
        traj_pred[0, t_min:] = traj.to(traj_pred.dtype)
        self._check_kernels()
        return (trajectories[:, :, order], visibilities[:, :, order], traj_pred,
                torch.zeros_like(visibilities), query_points[:, order])

    @torch.no_grad()
    def forward_test_main(self, rgbs, query_points, trajectories, visibilities):
        """vanilla_tracker.py:305-412: all points are propagated from frame 0 of `rgbs`."""
        cfg = self.engine_config()
        T, h, w = rgbs.shape[1], rgbs.shape[-2], rgbs.shape[-1]
        feats, Hf, Wf = self.get_feats_hwc(rgbs[0], split=True)
        plan = engine.plan_clip(T, [0], cfg)
        tk = engine.run_affinity(feats, Hf, Wf, plan, cfg, channels=self.feat_channels)
        self._refine_stats = [tk.refine_stats] if tk.refine_stats is not None else None
        pts = query_points[0, :, 1:].to(rgbs.device, torch.float32)
        _, coords = engine.run_propagation(tk, 0, pts, Hf, Wf, h, w, cfg)
        self._check_kernels()
        return trajectories, visibilities, coords.unsqueeze(0), torch.zeros_like(visibilities), query_points


@MODELS.register_module()
class HRVanillaTracker(VanillaTracker):
    """Single-scale local-window variant (vanilla_tracker.py:417-660): mmcv.ops.Correlation(max_displacement=R)
    + F.unfold + top-k is one call to fgvc_local_corr_topk_{f16x3,f32} per frame.

    Keys read here exactly as the reference reads them: `neighbor_range` (default 24, -> R), `withnorm` (sic, :437; NOT
    `with_norm`), `topk` (10), `temperature` (default 1, :563), `precede_frames`, `with_first` (slot 0, default True, :534;
    regrouping, default False, :246), `batch_step`.  `dilations` is passed by the reference as the dilation of the
    Correlation KERNEL (:426-428), which has a single tap (kernel_size=1): any value gives the same result, so it is accepted
    and has no effect (mmcv arithmetic: "parity unpinned").  `save_mem=True` (:432, :537-545: one key frame, the previous one, and no
    first-frame slot) runs as in the reference for `precede_frames = 1` and is refused otherwise (the reference's branch then pairs one
    key frame with several label maps and fails at its reshape, :552)."""

    def __init__(self, stride=2, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.stride = stride
        g = self.test_cfg.get
        self.infer_radius = g("neighbor_range", 24) // 2
        self.infer_dilations = g("dilations", 1)
        self.grid_size_hr = 2 * self.infer_radius + 1
        self.save_mem = bool(g("save_mem", False))                      # :432
        if self.save_mem and int(g("precede_frames", 5)) != 1:
            raise NotImplementedError("fgvc_amd: HRVanillaTracker save_mem=True needs precede_frames = 1: the reference's branch pairs ONE key "
                                      "frame with the label maps of all preceding frames (vanilla_tracker.py:520-545) and fails at its "
                                      "reshape (:552) for any other value")

    def _feats_hwc(self, frames: torch.Tensor):
        """frames (T,3,h,w) -> channels-last rows (T, HfWf, C'), L2-normalised iff `withnorm` (:437-439), Hf, Wf."""
        step = int(self.test_cfg.get("batch_step", 5))
        norm = bool(self.test_cfg.get("withnorm", True))
        chunks = []
        for i in range(0, frames.shape[0], step):
            f = self.extract_feat(frames[i:i + step])
            if isinstance(f, (tuple, list)):
                f = f[0]
            Hf, Wf = f.shape[-2:]
            chunks.append(ops.normalize_to_hwc(f.float(), norm, pad=True))
        return (chunks[0] if len(chunks) == 1 else torch.cat(chunks, 0)), Hf, Wf, norm

    def _sweep(self, feats, Hf, Wf, norm, h, w, pts):
        """Labels + read-out for the points `pts` (P,2)=(x,y) given at the first frame of the bank `feats` (T,HW,C')."""
        g = self.test_cfg.get
        T, dev, P = feats.shape[0], feats.device, pts.shape[0]
        labels = torch.zeros((T, Hf * Wf, P), device=dev)
        ops.gaussian_labels(pts, Hf, Wf, h // Hf, 6.0, out=labels[0])
        R, k, tau = self.infer_radius, int(g("topk", 10)), float(g("temperature", 1))
        pre, with_first = int(g("precede_frames", 5)), bool(g("with_first", True))
        for f in range(1, T):
            # save_mem (:537-545): the single key frame f - 1 (features re-extracted there, the same values here), no first-frame slot
            ks = [f - 1] if self.save_mem else engine.key_slots(f, 0, pre, with_first)
            # `normalized` must be the flag that controlled the normalisation of the bank: the bf16-pipe kernel's fixed-point
            # keys assume |q.k| <= 1 and raw ResNet dot products are not (they go through fgvc_local_corr_topk_f32)
            idx, _, weight = ops.local_corr_topk(feats[f:f + 1], feats[ks], Hf, Wf, R, k, tau, normalized=norm)
            ops.propagate_topk(labels, torch.tensor(ks, dtype=torch.int32, device=dev), idx, weight, Hf, Wf, Hf, Wf,
                               window_L=2 * R + 1, out=labels[f])
        return ops.softargmax_top5(labels, Hf, Wf, h, w, gauss_points=pts)

    @torch.no_grad()
    def forward_test_main(self, rgbs, query_points, trajectories, visibilities):
        """"backward warping" (:492-585): labels of frame f = top-k softmax over the (2R+1)^2 windows of its key slots."""
        h, w = rgbs.shape[-2], rgbs.shape[-1]
        feats, Hf, Wf, norm = self._feats_hwc(rgbs[0])
        coords = self._sweep(feats, Hf, Wf, norm, h, w, query_points[0, :, 1:].to(rgbs.device, torch.float32))
        self._check_kernels()
        vis = torch.zeros_like(visibilities) if visibilities is not None else None
        return trajectories, visibilities, coords.unsqueeze(0), vis, query_points

    def _coord_field(self, qrow, krow, H, W, scale, norm):
        idx, _, weight = ops.local_corr_topk(qrow, krow, H, W, self.infer_radius, int(self.test_cfg.get("topk", 10)),
                                             float(self.test_cfg.get("temperature", 1)), normalized=norm)
        return ops.topk_coord(idx, weight, H, W, self.infer_radius, scale).t().reshape(1, 2, H, W)

    @torch.no_grad()
    def get_coord(self, query_feat, key_feats, shape, scale):
        """vanilla_tracker.py:445-488: dense forward-warping field.  query_feat (1,C,H,W), key_feats (1,C,H,W) ->
        (1,2,H,W) expected (x,y) image coordinate of each query pixel's match in the key frame."""
        norm = bool(self.test_cfg.get("withnorm", True))
        H, W = query_feat.shape[-2:]
        qf = ops.normalize_to_hwc(query_feat.float(), norm, pad=True)
        kf = ops.normalize_to_hwc(key_feats.float(), norm, pad=True)
        return self._coord_field(qf, kf[:1], H, W, scale, norm)

    @torch.no_grad()
    def forward_test_forward(self, imgs, ref_seg_map=None, img_meta=None, ref=None, save_image=False, save_path=None,
                             iteration=None):
        """"forward warping" (vanilla_tracker.py:591-660): push the points `ref` (B,2,P) = rows (y,x) through the chain of
        frame-to-frame coordinate fields (query = frame max(0, f - precede_frames), key = frame f).  imgs (B,1,3,T,h,w).
        Returns what the reference returns: a list over the batch of float64 numpy arrays (2,P,T), rows (x,y)."""
        from .common import bilinear_sample
        h, w = imgs.shape[-2:]
        imgs = imgs.reshape((-1,) + imgs.shape[2:])                                    # :599
        assert imgs.shape[0] == 1, "batch size must be 1 (get_feats, vanilla_tracker.py:134)"
        T = imgs.shape[2]
        feats, Hf, Wf, norm = self._feats_hwc(imgs[0].transpose(0, 1))
        scale = w // Wf                                                                # :609
        coord = torch.flip(ref, (1,)).float()                                          # :611 (B,2,P) -> rows (x,y)
        coords = [coord]
        pre = int(self.test_cfg.get("precede_frames", 5))
        for f in range(1, T):
            start = max(0, f - pre)
            field = self._coord_field(feats[start:start + 1], feats[f:f + 1], Hf, Wf, scale, norm)
            coord = bilinear_sample(field, coord.clone().unsqueeze(-1) / scale, align_corners=True).squeeze(-1)   # :639
            coords.append(coord)
        return list(torch.stack(coords, -1).cpu().numpy().astype(float))              # :642-660

    @torch.no_grad()
    def forward_test(self, rgbs, query_points, trajectories, visibilities, **kw):
        if not self.test_cfg.get("with_first", False):
            return self.forward_test_main(rgbs, query_points, trajectories, visibilities)
        # inherited regrouping (vanilla_tracker.py:246-299): one sweep per distinct query time over the tail of the clip; the
        # frames are encoded ONCE (the reference re-encodes rgbs[:, t:] per group, :284 -- same features)
        h, w = rgbs.shape[-2], rgbs.shape[-1]
        times = query_points[0, :, 0].to(torch.int64)
        t_min = int(times.min())
        feats, Hf, Wf, norm = self._feats_hwc(rgbs[0, t_min:])
        order, col = [], 0
        traj_pred = torch.zeros_like(trajectories)
        for t in sorted(set(times.tolist())):
            sel = (times == t).nonzero().flatten()
            pts = query_points[0, sel, 1:].to(rgbs.device, torch.float32)
            coords = self._sweep(feats[t - t_min:], Hf, Wf, norm, h, w, pts)
            traj_pred[0, t:, col:col + sel.numel()] = coords.to(traj_pred.dtype)
            order.extend(sel.tolist())
            col += sel.numel()
        order = torch.tensor(order, device=rgbs.device)
        self._check_kernels()
        return (trajectories[:, :, order], visibilities[:, :, order], traj_pred, torch.zeros_like(visibilities),
                query_points[:, order])
