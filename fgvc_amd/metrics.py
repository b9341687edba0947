"""Evaluation arithmetic of the path's consumers -- host-side numpy, not a kernel.

* tapvid_metrics      : TAP-Vid occlusion accuracy / pts-within-threshold / Jaccard
                        (mmpt/datasets/tapvid_evaluation_datasets.py:106-249)
* trajectory_summary  : per-point summary used by TAPVidDataset.tapvid_evaluate
                        (mmpt/datasets/flyingthingsplus/utils/figures.py:179-296; docstring known answers :225-246)
* jhmdb_pck           : PCK@alpha with the 0.6 * ||bbox of visible GT joints|| normaliser
                        (mmpt/datasets/jhmdb_dataset.py:144-152, :174-256)
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence

import numpy as np

TAPVID_THRESHOLDS = (1, 2, 4, 8, 16)


def tapvid_metrics(query_points: np.ndarray, gt_occluded: np.ndarray, gt_tracks: np.ndarray,
                   pred_occluded: np.ndarray, pred_tracks: np.ndarray, query_mode: str = "first",
                   extra_thresholds: Iterable[float] = ()) -> Dict[str, np.ndarray]:
    """query_points (b,n,3)=(t,y,x) [only t is used]; gt/pred_occluded (b,n,T) bool; tracks (b,n,T,2)=(x,y).
    Returns per-video arrays of shape (b,), values in [0,1]."""
    b, n, T = gt_occluded.shape
    qt = np.round(query_points[..., 0]).astype(np.int64)
    evaluate = np.ones((b, n, T), dtype=bool)
    evaluate[np.arange(b)[:, None], np.arange(n)[None, :], qt] = False         # never score the query frame
    if query_mode == "first":
        # the reference indexes gt_occluded[i] (shape (n,T)) with np.where(...)[0][0]: the first POINT that has a
        # visible frame, and blanks evaluation_points[i, :that] -- reproduced as is
        for i in range(b):
            first = np.where(gt_occluded[i] == 0)[0][0]
            evaluate[i, :first] = False
    elif query_mode != "strided":
        raise ValueError("Unknown query mode " + query_mode)
    out = {}
    out["occlusion_accuracy"] = ((pred_occluded == gt_occluded) & evaluate).sum((1, 2)) / evaluate.sum()
    visible, pred_visible = ~gt_occluded.astype(bool), ~pred_occluded.astype(bool)
    d2 = ((pred_tracks - gt_tracks) ** 2).sum(-1)
    n_vis = (visible & evaluate).sum((1, 2))
    fracs, jacs = [], []
    for th in TAPVID_THRESHOLDS:
        within = d2 < th ** 2
        correct = within & visible
        frac = (correct & evaluate).sum((1, 2)) / n_vis
        tp = (correct & pred_visible & evaluate).sum((1, 2))
        fp = (((~visible) & pred_visible) | ((~within) & pred_visible)) & evaluate
        jac = tp / (n_vis + fp.sum((1, 2)))
        out[f"pts_within_{th}"], out[f"jaccard_{th}"] = frac, jac
        fracs.append(frac)
        jacs.append(jac)
    for th in extra_thresholds:
        out[f"pts_within_{th}"] = ((d2 < th ** 2) & visible & evaluate).sum((1, 2)) / n_vis
    out["average_jaccard"] = np.mean(np.stack(jacs, 1), 1)
    out["average_pts_within_thresh"] = np.mean(np.stack(fracs, 1), 1)
    return out


def _ade(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.linalg.norm(a - b, axis=-1).mean()) if len(a) else float("nan")


def trajectory_summary(traj_gt, traj_pred, vis_gt, vis_pred, query_point, query_mode: str = "first") -> Dict[str, float]:
    """One point: traj (T,2), vis (T,) bool, query_point (3,)=(t,x,y).  TAP-Vid numbers are x100 like the reference's
    compute_summary (figures.py:289); ADEs in pixels."""
    traj_gt, traj_pred = np.asarray(traj_gt, np.float64), np.asarray(traj_pred, np.float64)
    vis_gt, vis_pred = np.asarray(vis_gt).astype(bool), np.asarray(vis_pred).astype(bool)
    s = {"ade": _ade(traj_gt, traj_pred), "ade_visible": _ade(traj_gt[vis_gt], traj_pred[vis_gt]),
         "n_timesteps": len(traj_gt), "n_timesteps_visible": int(vis_gt.sum())}
    m = tapvid_metrics(np.asarray(query_point, np.float64)[None, None], ~vis_gt[None, None], traj_gt[None, None],
                       ~vis_pred[None, None], traj_pred[None, None], query_mode)
    s.update({k: float(v[0]) * 100 for k, v in m.items()})
    return s


def tapvid_evaluate(results: Sequence, query_mode: str = "first") -> Dict[str, float]:
    """results: list of the tracker's 5-tuples (trajectories (1,T,P,2), visibilities (1,T,P), trajectories_pred,
    visibilities_pred, query_points (1,P,3)) as produced by single_gpu_test / multi_gpu_test
    (tapvid.py:198-312).  Mean over all points of all videos."""
    rows = []
    for traj, vis, tp, vp, qp in results:
        traj, vis, tp, vp, qp = (np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in (traj, vis, tp, vp, qp))
        for p in range(traj.shape[2]):
            rows.append(trajectory_summary(traj[0, :, p], tp[0, :, p], vis[0, :, p] > 0.5, vp[0, :, p] > 0.5,
                                           qp[0, p], query_mode))
    keys = rows[0].keys() if rows else []
    return {k: float(np.nanmean([r[k] for r in rows])) for k in keys}


def jhmdb_pck(pred_poses: Sequence[np.ndarray], gt_poses: Sequence[np.ndarray],
              alphas: Sequence[float] = (0.1, 0.2, 0.3, 0.4, 0.5)) -> Dict[str, float]:
    """pred_poses / gt_poses: per video (2, J, T) arrays (x;y).  A joint counts where the PREDICTION's x > 0
    (jhmdb_dataset.py:219); distance / (0.6 * ||bbox of those joints' GT||); PCK = % of distances <= alpha per joint,
    then the mean over joints."""
    J = gt_poses[0].shape[1]
    dists = [[] for _ in range(J)]
    for pred, gt in zip(pred_poses, gt_poses):
        T = min(pred.shape[-1], gt.shape[-1])
        pred, gt = pred[..., :T], gt[..., :T]
        seen = pred[0] > 0                                                     # (J,T)
        hi = np.where(seen[None], gt, -1.0).max(axis=1)                        # (2,T)
        lo = np.where(seen[None], gt, 1e6).min(axis=1)
        box = 0.6 * np.linalg.norm(hi - lo, axis=0)                            # (T,)
        d = np.linalg.norm(pred - gt, axis=0) / box[None]
        for j in range(J):
            dists[j].extend(d[j, seen[j]].tolist())
    out = {}
    for a in alphas:
        per_joint = [100.0 * np.mean(np.asarray(dj) <= a) for dj in dists if len(dj)]
        out[f"PCK@{a}"] = float(np.mean(per_joint))
    return out
