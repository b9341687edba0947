"""TAP-Vid sample format (mmpt/datasets/tapvid.py:122-174) on synthetic clips, plus the strided per-rank sampler
(mmpt/datasets/samplers/distributed_sampler.py:53).  The real TAP-Vid / JHMDB files are not available offline;
`SyntheticTapVid` produces tensors with exactly the shapes, dtypes and conventions the model consumes:
    rgbs (1,T,3,h,w) float32 (stands for Lab-normalised frames), query_points (1,P,3) = (t,x,y) float32,
    trajectories (1,T,P,2) float32, visibilities (1,T,P) float32.
"""
from __future__ import annotations

import glob
import io
import os
import pickle

import numpy as np
import torch


class SyntheticTapVid:
    """Moving-texture clips with known point tracks: every frame is the first frame shifted by an integer
    (dx,dy) per frame, so ground-truth trajectories are exact and a tracker's accuracy is measurable."""

    def __init__(self, n_videos=4, frames=8, size=(256, 256), points=8, query_mode="first", seed=0, device="cpu"):
        self.n, self.T, self.h, self.w, self.P = n_videos, frames, size[0], size[1], points
        self.query_mode, self.seed, self.device = query_mode, seed, device

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000 + i)
        T, h, w, P = self.T, self.h, self.w, self.P
        pad = 2 * T
        base = torch.nn.functional.interpolate(torch.randn(1, 3, (h + 2 * pad) // 8 + 1, (w + 2 * pad) // 8 + 1, generator=g),
                                               size=(h + 2 * pad, w + 2 * pad), mode="bilinear", align_corners=False)[0]
        base = base + 0.25 * torch.randn(base.shape, generator=g)
        vx, vy = int(torch.randint(-2, 3, (1,), generator=g)), int(torch.randint(-2, 3, (1,), generator=g))
        rgbs = torch.stack([base[:, pad - vy * t: pad - vy * t + h, pad - vx * t: pad - vx * t + w] for t in range(T)], 0)
        t0 = torch.zeros(P) if self.query_mode == "first" else torch.randint(0, max(1, T // 2), (P,), generator=g).float()
        margin = 2 * T + 8
        x0 = torch.rand(P, generator=g) * (w - 2 * margin) + margin
        y0 = torch.rand(P, generator=g) * (h - 2 * margin) + margin
        ts = torch.arange(T).view(T, 1).float()
        traj = torch.stack([x0.view(1, P) + vx * (ts - t0.view(1, P)) + 0 * ts, y0.view(1, P) + vy * (ts - t0.view(1, P))], -1)
        # express the query at its own time: position at t0 is (x0,y0)
        qp = torch.stack([t0, x0, y0], -1)
        vis = (ts >= t0.view(1, P)).float()
        d = self.device
        return dict(rgbs=rgbs.unsqueeze(0).to(d), query_points=qp.unsqueeze(0).to(d),
                    trajectories=traj.unsqueeze(0).to(d), visibilities=vis.unsqueeze(0).to(d))


class StridedLoader:
    """indices[rank::world], one video per step (samples_per_gpu=1, tools/test.py:127)."""

    def __init__(self, dataset, rank=0, world=1):
        self.dataset, self.rank, self.world = dataset, rank, world
        self.total = len(dataset)

    def __iter__(self):
        for i in range(self.rank, len(self.dataset), self.world):
            yield self.dataset[i]

    def __len__(self):
        return len(range(self.rank, len(self.dataset), self.world))


# ---- TAP-Vid input contract (configs/eval/base_data.py:1-7: RGB2LAB + Normalize(mean=[50,0,0], std=[50,127,127])) -------
_M_RGB2XYZ = ((0.412453, 0.357580, 0.180423), (0.212671, 0.715160, 0.072169), (0.019334, 0.119193, 0.950227))
_WHITE_D65 = (0.950456, 1.0, 1.088754)


def rgb_to_lab(rgb: torch.Tensor) -> torch.Tensor:
    """sRGB in [0,1], (..., 3, h, w) float -> CIE L*a*b* (D65), L in [0,100], a/b in about [-127,127]: the conversion
    `cv2.cvtColor(img01_float32, cv2.COLOR_RGB2Lab)` performs in the reference (augmentation.py:1372-1391), with OpenCV's
    documented constants.  Parity with OpenCV is UNPINNED here (cv2 is not installed; its float path evaluates the
    transfer curve and the cube root through spline tables, documented accuracy ~1e-3): the known answers in
    tests/test_host.py are the CIE values of the sRGB primaries."""
    x = rgb.to(torch.float32)
    lin = torch.where(x > 0.04045, ((x + 0.055) / 1.055).clamp_min(0) ** 2.4, x / 12.92)
    r, g, b = lin.unbind(-3)
    xyz = [(m[0] * r + m[1] * g + m[2] * b) / w for m, w in zip(_M_RGB2XYZ, _WHITE_D65)]
    f = [torch.where(t > 0.008856, t.clamp_min(1e-12) ** (1.0 / 3.0), 7.787 * t + 16.0 / 116.0) for t in xyz]
    L = torch.where(xyz[1] > 0.008856, 116.0 * f[1] - 16.0, 903.3 * xyz[1])
    return torch.stack([L, 500.0 * (f[0] - f[1]), 200.0 * (f[1] - f[2])], -3)


def preprocess_tapvid_frames(frames_uint8: torch.Tensor, size=(256, 256)) -> torch.Tensor:
    """(T, h0, w0, 3) uint8 RGB -> (1, T, 3, h, w) float32 network input: bilinear resize to `size`, /255, RGB->Lab,
    (x - [50,0,0]) / [50,127,127]  (configs/eval/base_data.py:1-7; tapvid.py:106-108 scales the points by the same size)."""
    x = frames_uint8.permute(0, 3, 1, 2).to(torch.float32)
    if tuple(x.shape[-2:]) != tuple(size):
        x = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)
    lab = rgb_to_lab((x / 255.0).clamp(0, 1))
    mean = torch.tensor([50.0, 0.0, 0.0], device=lab.device).view(1, 3, 1, 1)
    std = torch.tensor([50.0, 127.0, 127.0], device=lab.device).view(1, 3, 1, 1)
    return ((lab - mean) / std).unsqueeze(0)


# ---- TAP-Vid files (mmpt/datasets/tapvid.py:36-174, tapvid_evaluation_datasets.py:284-395) ---------------------------------
def _queries_first(occluded: np.ndarray, points: np.ndarray):
    """One query per track at its first visible frame; tracks that are never visible are dropped
    (tapvid_evaluation_datasets.py:352-395).  occluded (P,T) bool, points (P,T,2) = (x,y).  Returns (queries (Q,3) = (t,y,x),
    points (Q,T,2), occluded (Q,T))."""
    keep = (~occluded).any(axis=1)
    points, occluded = points[keep], occluded[keep]
    t0 = np.argmax(~occluded, axis=1)
    rows = np.arange(points.shape[0])
    q = np.stack([t0.astype(points.dtype), points[rows, t0, 1], points[rows, t0, 0]], axis=-1)
    return q, points, occluded


def _queries_strided(occluded: np.ndarray, points: np.ndarray, stride: int = 5):
    """A query at every `stride`-th frame for every track visible there; tracks are repeated per query
    (tapvid_evaluation_datasets.py:284-349)."""
    qs, ps, os_ = [], [], []
    for t in range(0, occluded.shape[1], stride):
        vis = ~occluded[:, t]
        qs.append(np.stack([np.full(int(vis.sum()), t, dtype=points.dtype), points[vis, t, 1], points[vis, t, 0]], axis=-1))
        ps.append(points[vis])
        os_.append(occluded[vis])
    return np.concatenate(qs, 0), np.concatenate(ps, 0), np.concatenate(os_, 0)


class TapVidPickles:
    """TAP-Vid videos from pickles in the sample format the model consumes (the reference's TAPVidDataset, tapvid.py:36-174).
    `root`: a directory of `*.pkl` files with one video each (what the reference globs, tapvid.py:66), or ONE pickle holding
    {name: video} or [video, ...] (the published tapvid_davis.pkl).  A video = dict(video (T,H,W,3) uint8 frames -- or an
    array of encoded JPEG byte strings, tapvid.py:91-99 --, points (P,T,2) = (x,y) in [0,1], occluded (P,T) bool).
    Frames go through the TAP-Vid input contract (`preprocess_tapvid_frames`: resize to `input_size`, RGB->Lab, normalise),
    points are scaled to `input_size` pixels (tapvid.py:106-108), queries are sampled by `query_mode` 'first' | 'strided'."""

    def __init__(self, root: str, query_mode: str = "first", input_size=(256, 256), device="cpu"):
        if query_mode not in ("first", "strided"):
            raise ValueError(f"Unknown query mode {query_mode}.")                                   # tapvid.py:115
        self.query_mode, self.input_size, self.device = query_mode, tuple(input_size), device
        self.videos = []                       # (path, key): key = None for one-video files
        files = sorted(glob.glob(os.path.join(root, "*.pkl"))) if os.path.isdir(root) else [root]
        for f in files:
            with open(f, "rb") as fh:
                obj = pickle.load(fh)
            if isinstance(obj, dict) and "video" in obj:
                self.videos.append((f, None))
            elif isinstance(obj, dict):
                self.videos.extend((f, k) for k in obj)
            else:
                self.videos.extend((f, i) for i in range(len(obj)))
        self._open = (None, None)

    def __len__(self):
        return len(self.videos)

    def _raw(self, i):
        f, key = self.videos[i]
        if self._open[0] != f:
            with open(f, "rb") as fh:
                self._open = (f, pickle.load(fh))
        obj = self._open[1]
        return obj if key is None else obj[key]

    @staticmethod
    def _frames(video) -> torch.Tensor:
        if len(video) and isinstance(video[0], (bytes, bytearray)):                                  # JPEG bytes
            from PIL import Image
            video = np.stack([np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in video])
        return torch.from_numpy(np.ascontiguousarray(np.asarray(video, dtype=np.uint8)))

    def __getitem__(self, i):
        sample = self._raw(i)
        frames = self._frames(sample["video"])                                                      # (T,H,W,3) uint8
        h, w = self.input_size
        points = np.asarray(sample["points"], dtype=np.float32) * np.array([w, h], dtype=np.float32)   # tapvid.py:108
        occluded = np.asarray(sample["occluded"]).astype(bool)
        q, points, occluded = (_queries_first if self.query_mode == "first" else _queries_strided)(occluded, points)
        query_points = torch.from_numpy(q[:, [0, 2, 1]].astype(np.float32))                         # (t,y,x) -> (t,x,y), :132-133
        traj = torch.from_numpy(points).permute(1, 0, 2).contiguous()                               # (T,P,2)
        vis = ~torch.from_numpy(occluded).permute(1, 0).contiguous()                                # (T,P)
        P = query_points.shape[0]
        qt = query_points[:, 0].long()
        # Kubric reports query points on the crop boundary as invisible (tapvid.py:135-150)
        for p in range(P):
            if not vis[qt[p], p]:
                x, y = float(query_points[p, 1]), float(query_points[p, 2])
                xb, yb = min(abs(x), abs(x - (w - 1))) < 1e-3, min(abs(y), abs(y - (h - 1))) < 1e-3
                xin, yin = 0 <= x <= w - 1, 0 <= y <= h - 1
                if (xb and yin) or (xin and yb) or (xb and yb):
                    vis[qt[p], p] = True
        assert bool(vis[qt, torch.arange(P)].all()), "Query points must be visible"                # tapvid.py:160-161
        assert torch.allclose(query_points[:, 1:], traj[qt, torch.arange(P)], atol=1.0)             # tapvid.py:164-168
        rgbs = preprocess_tapvid_frames(frames.to(self.device), self.input_size)                    # (1,T,3,h,w)
        d = self.device
        return dict(rgbs=rgbs, query_points=query_points.unsqueeze(0).to(d), trajectories=traj.unsqueeze(0).to(d),
                    visibilities=vis.float().unsqueeze(0).to(d))


# ---- JHMDB (mmpt/datasets/jhmdb_dataset.py:72-141) -> the tracker's sample format -------------------------------------------------
class JhmdbPoses:
    """JHMDB pose videos in the sample format VanillaTracker.forward_test consumes.  The reference's JHMDB dataset yields
    `imgs` / `ref_seg_map` (jhmdb_dataset.py:112-141), which the released tracker does not accept;
    this is the adaptation the release lacks: the 15 joints of the FIRST frame become query points at t = 0.

    Files as the reference reads them (:72-96): `<list_path>/<split>_list.txt` with lines "<anno.mat> <frames dir>", both relative
    to `root`; frames `*.png`; the .mat holds `pos_img` (2, 15, T) = (x; y), 1-based (:125 "magic -1").  Frames are resized to
    `input_size` (test_pipeline_jhmdb: 320 x 320, keep_ratio=False) and go through the RGB->Lab contract; joints are scaled
    the same way.  `pose_prediction(sample, traj_pred)` maps a prediction back to the (2, 15, T) original-resolution array that
    `metrics.jhmdb_pck` (jhmdb_dataset.py:174-256) scores against `sample["gt_poses"]`."""
    NUM_KEYPOINTS = 15

    def __init__(self, root: str, list_path: str = None, split: str = "val", input_size=(320, 320), device="cpu"):
        self.root, self.input_size, self.device = root, tuple(input_size), device
        self.samples = []
        with open(os.path.join(list_path or root, f"{split}_list.txt")) as f:
            for line in f:
                if not line.strip():
                    continue
                anno, vname = line.strip("\n").split()
                frames = sorted(glob.glob(os.path.join(root, vname, "*.png")))
                if frames:
                    self.samples.append(dict(frames_path=frames, anno_path=os.path.join(root, anno), video_path=os.path.join(root, vname)))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        import scipy.io as sio
        from PIL import Image
        s = self.samples[i]
        frames = torch.from_numpy(np.stack([np.array(Image.open(p).convert("RGB")) for p in s["frames_path"]]))   # (T,h0,w0,3)
        T, h0, w0 = frames.shape[:3]
        gt = np.asarray(sio.loadmat(s["anno_path"])["pos_img"], dtype=np.float64) - 1.0                          # (2,15,Tg)
        Tg = gt.shape[-1]
        h, w = self.input_size
        scale = np.array([w / w0, h / h0]).reshape(2, 1, 1)
        gts = gt * scale
        n = min(T, Tg)                                                                                           # :205
        traj = torch.from_numpy(gts[:, :, :n]).permute(2, 1, 0).float().contiguous()                             # (T,15,2)
        qp = torch.cat([torch.zeros(self.NUM_KEYPOINTS, 1), traj[0]], 1)                                        # (15,3) = (0,x,y)
        rgbs = preprocess_tapvid_frames(frames[:n].to(self.device), self.input_size)
        d = self.device
        return dict(rgbs=rgbs, query_points=qp.unsqueeze(0).to(d), trajectories=traj.unsqueeze(0).to(d),
                    visibilities=torch.ones(1, n, self.NUM_KEYPOINTS, device=d)), dict(gt_poses=gt[:, :, :n], original_shape=(h0, w0))

    def pose_prediction(self, meta, traj_pred) -> np.ndarray:
        """traj_pred (1,T,15,2) at the network input size -> (2,15,T) at the video's own resolution."""
        h0, w0 = meta["original_shape"]
        h, w = self.input_size
        p = np.asarray(traj_pred.cpu() if hasattr(traj_pred, "cpu") else traj_pred, dtype=np.float64)[0]            # (T,15,2)
        return p.transpose(2, 1, 0) * np.array([w0 / w, h0 / h]).reshape(2, 1, 1)


def jhmdb_evaluate(model, dataset: "JhmdbPoses"):
    """Run the tracker over a JhmdbPoses dataset and score PCK@0.1..0.5 as the reference's pck_evaluate does."""
    from . import metrics
    preds, gts = [], []
    for i in range(len(dataset)):
        sample, meta = dataset[i]
        out = model(test_mode=True, **sample)
        assert torch.equal(out[4], sample["query_points"])              # every query is at t = 0: one group, order kept
        pred = out[2]
        preds.append(dataset.pose_prediction(meta, pred))
        gts.append(meta["gt_poses"])
    return metrics.jhmdb_pck(preds, gts)


# SMAL joints BADJA annotates (badja_dataset.py:71-82 `SMALJointInfo.annotated_classes`): 20 of the 37 per frame
BADJA_ANNOTATED = (8, 9, 10, 12, 13, 14, 15, 18, 19, 20, 22, 23, 24, 25, 28, 31, 32, 33, 35, 36)


class BadjaPoses:
    """BADJA animal videos in the sample format VanillaTracker.forward_test consumes.  Like the reference's JHMDB dataset, its
    BadjaDataset yields `imgs` / `ref_seg_map` (badja_dataset.py:355-410), which the released tracker does not accept (SURVEY.md
    section 8b): the joints of the FIRST frame become query points at t = 0.

    Files as the reference reads them (:152-229): `<list_path>/joint_annotations/*.json`, each a list of records with `image_path`,
    `segmentation_path` (their first 6 characters are dropped and the rest joined to `root`), `joints` (37 x (y, x)) and `visibility`;
    a video = the frames `JPEGImages/Full-Resolution/<animal>/%05d.jpg` from the first to the last annotated number, silhouettes
    `Annotations/Full-Resolution/<animal>/%05d.png`; frames without a record have no joints (:206-211).  Videos under `extra_videos`
    are skipped (:177); the reference's IGNORE_ANIMALS list is ONE string by a missing comma (:38-41) and ignores nothing -- same here.
    Frames are resized to `size` = (320, 512) (:355-362; the released pipeline's Resize(-1, 320) then changes nothing) and go through
    the RGB->Lab contract; joints are scaled the same way (:364-366, :489-494).  Image resizing is PIL's here, OpenCV's there: the
    adapter is restated from the file, "parity unpinned" (no BADJA data or mmcv offline)."""

    def __init__(self, root: str, list_path: str = None, size=(320, 512), length: int = -1, device="cpu"):
        import json
        self.root, self.size, self.length, self.device = root, tuple(size), int(length), device
        self.videos = []
        adir = os.path.join(list_path or root, "joint_annotations")
        for name in sorted(os.listdir(adir)):
            with open(os.path.join(adir, name)) as f:
                records = json.load(f)
            first, last = records[0]["segmentation_path"], records[-1]["segmentation_path"]
            if "extra_videos" in first:
                continue
            animal = first.split("/")[-2]
            lo, hi = int(first.split("/")[-1].split(".")[0]), int(last.split("/")[-1].split(".")[0])
            by_file = {os.path.join(root, r["image_path"][6:]): r for r in records}
            frames, segs, joints, vis = [], [], [], []
            for fr in range(lo, hi + 1):
                img = os.path.join(root, "JPEGImages/Full-Resolution/%s/%05d.jpg" % (animal, fr))
                r = by_file.get(img)
                frames.append(img)
                segs.append(os.path.join(root, r["segmentation_path"][6:]) if r else
                            os.path.join(root, "Annotations/Full-Resolution/%s/%05d.png" % (animal, fr)))
                joints.append(np.asarray(r["joints"], dtype=np.float64)[list(BADJA_ANNOTATED)] if r else None)
                vis.append(np.asarray(r["visibility"])[list(BADJA_ANNOTATED)] if r else None)
            if frames:
                self.videos.append(dict(name=animal, frames=frames, segs=segs, joints=joints, visibles=vis))

    def __len__(self):
        return len(self.videos)

    def __getitem__(self, i):
        from PIL import Image
        v = self.videos[i]
        n = len(v["frames"]) if self.length == -1 else min(self.length, len(v["frames"]))
        imgs = [Image.open(p).convert("RGB") for p in v["frames"][:n]]
        w0, h0 = imgs[0].size
        h, w = self.size
        sy, sx = h / h0, w / w0
        frames = torch.from_numpy(np.stack([np.asarray(im) for im in imgs]))                       # (T,h0,w0,3)
        # silhouette -> the frame's size, then the network size.  The reference's first resize is cv2.resize(sil, (w, h), cv2.INTER_NEAREST)
        # (badja_dataset.py:255, :276): the flag sits in the `dst` position, so OpenCV runs its default, INTER_LINEAR -- when a mask and
        # its frame differ in size the `seg > 0` area (the PCK threshold) is the bilinear one.  Same size: both are the identity.
        segs = [np.asarray(Image.open(p).resize((w0, h0), Image.BILINEAR).resize((w, h), Image.NEAREST)) for p in v["segs"][:n]]
        joints = [None if j is None else j * np.array([sy, sx]) for j in v["joints"][:n]]          # (J,2) = (y,x) at the network size
        visibles = list(v["visibles"][:n])
        assert joints[0] is not None, "BADJA: the first frame of a video carries the query joints (badja_dataset.py:364)"
        J = joints[0].shape[0]
        traj = torch.zeros(n, J, 2)
        vis = torch.zeros(n, J)
        for t in range(n):
            if joints[t] is not None:
                traj[t] = torch.from_numpy(joints[t][:, ::-1].copy()).float()                      # (x,y)
                vis[t] = torch.from_numpy((np.asarray(visibles[t]) > 0).astype(np.float32))
        qp = torch.cat([torch.zeros(J, 1), traj[0]], 1)                                            # (J,3) = (0,x,y)
        rgbs = preprocess_tapvid_frames(frames.to(self.device), self.size)
        d = self.device
        return (dict(rgbs=rgbs, query_points=qp.unsqueeze(0).to(d), trajectories=traj.unsqueeze(0).to(d), visibilities=vis.unsqueeze(0).to(d)),
                dict(joints=joints, visibles=visibles, segs=segs, original_shape=(h0, w0), name=v["name"]))

    @staticmethod
    def pose_prediction(traj_pred) -> np.ndarray:
        """traj_pred (1,T,J,2) = (x,y) at the network size -> (2,J,T), the layout pck_evaluate reads (:526-527)."""
        p = np.asarray(traj_pred.cpu() if hasattr(traj_pred, "cpu") else traj_pred, dtype=np.float64)[0]
        return p.transpose(2, 1, 0)


def badja_evaluate(model, dataset: "BadjaPoses"):
    """Run the tracker over a BadjaPoses dataset and score PCK@0.1..0.4 as the reference's pck_evaluate does (BASELINE.md quotes its
    PCK@0.2)."""
    from . import metrics
    preds, js, vs, ss = [], [], [], []
    for i in range(len(dataset)):
        sample, meta = dataset[i]
        out = model(test_mode=True, **sample)
        assert torch.equal(out[4], sample["query_points"])              # every query is at t = 0: one group, order kept
        preds.append(dataset.pose_prediction(out[2]))
        js.append(meta["joints"]); vs.append(meta["visibles"]); ss.append(meta["segs"])
    return metrics.badja_pck(preds, js, vs, ss)
