"""Evaluation arithmetic of the path's consumers -- host-side numpy, not a kernel.

* tapvid_metrics      : TAP-Vid occlusion accuracy / pts-within-threshold / Jaccard
                        (mmpt/datasets/tapvid_evaluation_datasets.py:106-249)
* trajectory_summary  : per-point summary used by TAPVidDataset.tapvid_evaluate
                        (mmpt/datasets/flyingthingsplus/utils/figures.py:179-296; docstring known answers :225-246)
* jhmdb_pck           : PCK@alpha with the 0.6 * ||bbox of visible GT joints|| normaliser
                        (mmpt/datasets/jhmdb_dataset.py:144-152, :174-256)
* tapvid_summaries / save_results : the per-point records and the on-disk files of TAPVidDataset.tapvid_evaluate
                        (mmpt/datasets/tapvid.py:198-350): summaries<dataset>.json, results_df<dataset>.csv, results_list<dataset>.pkl
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Sequence

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


SUMMARY_EXTRA_THRESHOLDS = (0.01, 0.05, *[0.1 * (i + 1) for i in range(10)], *[(i + 1) for i in range(10)])   # figures.py:281-285


def _visible_chain(vis: np.ndarray, t: int):
    """Slice of the frames around query time t that are visible without interruption (figures.py:112-176)."""
    assert vis[t], "Query point must be visible"
    occ = np.nonzero(~vis)[0]
    after, before = occ[occ > t], occ[occ < t]
    return slice(int(before[-1]) + 1 if len(before) else 0, int(after[0]) if len(after) else len(vis))


def trajectory_summary(traj_gt, traj_pred, vis_gt, vis_pred, query_point, query_mode: str = "first", idx: str = None,
                       extra_thresholds: Iterable[float] = ()) -> Dict[str, float]:
    """One point: traj (T,2), vis (T,) bool, query_point (3,)=(t,x,y).  The record of the reference's compute_summary
    (figures.py:179-296): TAP-Vid numbers x100 (:289), ADEs in pixels, `idx` = "<iter>--<video_idx>--<point_idx_in_video>"."""
    traj_gt, traj_pred = np.asarray(traj_gt, np.float64), np.asarray(traj_pred, np.float64)
    vis_gt, vis_pred = np.asarray(vis_gt).astype(bool), np.asarray(vis_pred).astype(bool)
    s = {}
    if idx is not None:
        s["idx"] = idx
    s.update({"ade": _ade(traj_gt, traj_pred), "ade_visible": _ade(traj_gt[vis_gt], traj_pred[vis_gt])})
    t = int(query_point[0])
    if 0 <= t < len(vis_gt) and vis_gt[t]:
        ch = _visible_chain(vis_gt, t)
        s["ade_visible_chain"] = _ade(traj_gt[ch], traj_pred[ch])
        n_chain = ch.stop - ch.start
    else:
        s["ade_visible_chain"], n_chain = float("nan"), 0
    s.update({"n_timesteps": len(traj_gt), "n_timesteps_visible": int(vis_gt.sum()), "n_timesteps_visible_chain": n_chain})
    m = tapvid_metrics(np.asarray(query_point, np.float64)[None, None], ~vis_gt[None, None], traj_gt[None, None],
                       ~vis_pred[None, None], traj_pred[None, None], query_mode, extra_thresholds)
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


def tapvid_summaries(results: Sequence, query_mode: str = "first", input_size=(256, 256), size=(256, 256)):
    """The per-point records TAPVidDataset.tapvid_evaluate builds (tapvid.py:235-268): one per point of every video, coordinates
    scaled from the network input size back to the evaluation size (`size` = (h, w); :241-245), idx "<video>_<point>" fields
    iter / video_idx / point_idx_in_video as there.  Returns (summaries, results_list)."""
    summaries, results_list = [], []
    sx, sy = size[1] / input_size[1], size[0] / input_size[0]
    for vid, (traj, vis, tp, vp, qp) in enumerate(results):
        traj, vis, tp, vp, qp = (np.asarray(x.cpu() if hasattr(x, "cpu") else x, dtype=np.float64) for x in (traj, vis, tp, vp, qp))
        scale = np.array([sx, sy])
        for n in range(traj.shape[2]):
            rec = {"idx": f"{vid}_{n}", "iter": vid, "video_idx": 0, "point_idx_in_video": n,
                   "trajectory_gt": traj[0, :, n] * scale, "trajectory_pred": tp[0, :, n] * scale,
                   "visibility_gt": vis[0, :, n] > 0.5, "visibility_pred": vp[0, :, n] > 0.5, "query_point": qp[0, n]}
            results_list.append(rec)
            summaries.append(trajectory_summary(rec["trajectory_gt"], rec["trajectory_pred"], rec["visibility_gt"], rec["visibility_pred"],
                                                rec["query_point"], query_mode, idx=f"{vid}--0--{n}",
                                                extra_thresholds=SUMMARY_EXTRA_THRESHOLDS))
    return summaries, results_list


def save_results(summaries, results_list, output_dir: str, metadata: Dict) -> Dict[str, str]:
    """The files of the reference's save_results (tapvid.py:316-350), same names and formats: summaries<dataset>.json (list of
    records), results_df<dataset>.csv (pandas DataFrame.from_records(summaries).to_csv) and, when `results_list` is non-empty,
    results_list<dataset>.pkl.  (The figures the reference draws from the data frame are not produced.)  Returns the paths."""
    import json
    import os
    import pickle

    import pandas as pd
    dataset = metadata["dataset"]
    os.makedirs(output_dir, exist_ok=True)
    paths = {"summaries": os.path.join(output_dir, f"summaries{dataset}.json"),
             "results_df": os.path.join(output_dir, f"results_df{dataset}.csv")}
    with open(paths["summaries"], "w", encoding="utf8") as f:
        json.dump(summaries, f)
    pd.DataFrame.from_records(summaries).to_csv(paths["results_df"])
    if len(results_list) > 0:
        paths["results_list"] = os.path.join(output_dir, f"results_list{dataset}.pkl")
        with open(paths["results_list"], "wb") as f:
            pickle.dump(results_list, f)
    return paths


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


def badja_pck(pred_poses: Sequence[np.ndarray], joints: Sequence[Sequence[Optional[np.ndarray]]],
              visibles: Sequence[Sequence[Optional[np.ndarray]]], segs: Sequence[Sequence[np.ndarray]],
              ratios: Sequence[float] = (0.1, 0.2, 0.3, 0.4)) -> Dict[str, float]:
    """BADJA PCK as BadjaDataset.pck_evaluate computes it (badja_dataset.py:451-571).  Per video: pred_poses (2, J, T) = (x; y) at
    the evaluation size; joints[t] (J, 2) = (y, x) at that size or None for an unlabelled frame (then nothing is counted, :504-508);
    visibles[t] (J,); segs[t] the silhouette at that size.  A VISIBLE joint is correct at ratio r when its distance to the ground
    truth is < r * sqrt(number of silhouette pixels of that frame) (:541-546, strict).  Returns PCK@r over all counted joints of all
    videos (%), and "PCK@0.2 per-video mean": the mean of the per-video PCK@0.2 values (the number :552-557 / :578 write out --
    under the label 'PCK@0.1 AVG'; NaN as soon as one video has no visible labelled joint, as there)."""
    counts = {r: [] for r in ratios}
    per_video = []
    for pred, js, vs, ss in zip(pred_poses, joints, visibles, segs):
        mine = {r: [] for r in ratios}
        T = min(pred.shape[-1], len(js))
        for t in range(T):
            if js[t] is None:
                continue
            thr0 = math.sqrt(float((np.asarray(ss[t]) > 0).sum()))
            for j in range(js[t].shape[0]):
                if not vs[t][j] > 0:
                    continue
                d = math.sqrt((float(js[t][j, 1]) - float(pred[0, j, t])) ** 2 + (float(js[t][j, 0]) - float(pred[1, j, t])) ** 2)
                for r in ratios:
                    ok = d < r * thr0
                    counts[r].append(ok)
                    mine[r].append(ok)
        if 0.2 in mine:      # (a video without one visible labelled joint: the reference's np.mean([]) is NaN and stays in the average, :552-557)
            per_video.append(100.0 * float(np.mean(mine[0.2])) if mine[0.2] else float("nan"))
    out = {f"PCK@{r}": (100.0 * float(np.mean(counts[r])) if counts[r] else float("nan")) for r in ratios}
    out["PCK@0.2 per-video mean"] = float(np.mean(per_video)) if per_video else float("nan")
    return out
