"""CPU: evaluation arithmetic against the reference's function output (golden) and the
reference's docstring known answers (figures.py:225-246)."""
import os

import numpy as np
import torch

from fgvc_amd import metrics as M


def test_tapvid_metrics_match_reference_golden(golden):
    g = golden("tapvid_metrics")
    for mode in ("first", "strided"):
        m = M.tapvid_metrics(g["query_points"], g["gt_occluded"], g["gt_tracks"], g["pred_occluded"],
                             g["pred_tracks"], mode, extra_thresholds=[0.5, 3])
        keys = [k.split("__", 1)[1] for k in g if k.startswith(mode + "__")]
        assert set(keys) == set(m)
        for k in keys:
            assert np.allclose(m[k], g[f"{mode}__{k}"], rtol=0, atol=1e-12), (mode, k)


def test_summary_known_answers():
    """the doctest of compute_summary (values in the docstring are fractions; the code multiplies by 100)."""
    s = M.trajectory_summary([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0]], [[0.0, 0.0], [2.0, 2.0], [3.0, 3.0]],
                             [True, True, False], [True, True, True], [0, 0.0, 0.0], "first")
    assert abs(s["ade"] - 0.9428090453147888) < 1e-6 and abs(s["ade_visible"] - 0.7071067690849304) < 1e-6
    assert s["n_timesteps"] == 3 and s["n_timesteps_visible"] == 2
    exp = {"occlusion_accuracy": 0.5, "jaccard_1": 0.0, "jaccard_2": 0.5, "jaccard_4": 0.5, "jaccard_8": 0.5,
           "jaccard_16": 0.5, "average_jaccard": 0.4, "pts_within_1": 0.0, "pts_within_2": 1.0, "pts_within_4": 1.0,
           "pts_within_8": 1.0, "pts_within_16": 1.0, "average_pts_within_thresh": 0.8}
    for k, v in exp.items():
        assert abs(s[k] - 100 * v) < 1e-9, (k, s[k])


def test_jhmdb_pck_matches_reference_golden(golden):
    """F3: metrics.jhmdb_pck against what the GENUINE jhmdb_dataset_rgb.pck_evaluate returned (jhmdb_dataset.py:174-256, lifted by AST in
    tests/golden/gen_golden.py: gen_jhmdb_pck) for three synthetic videos -- joints marked invisible by the prediction, a clip longer
    than its annotation, a nearly still pose.  PCK@0.1 is the accuracy figure BASELINE.json names."""
    g = golden("jhmdb_pck")
    n = int(g["n_videos"])
    got = M.jhmdb_pck([g[f"pred{v}"] for v in range(n)], [g[f"gt{v}"] for v in range(n)])
    assert set(got) == {f"PCK@{a}" for a in (0.1, 0.2, 0.3, 0.4, 0.5)}
    for a in (0.1, 0.2, 0.3, 0.4, 0.5):
        assert abs(got[f"PCK@{a}"] - float(g[f"PCK_at_{a}"])) < 1e-9, (a, got[f"PCK@{a}"], float(g[f"PCK_at_{a}"]))
    assert 5.0 < got["PCK@0.1"] < got["PCK@0.5"] < 100.0                                # (a fixture that discriminates)


def test_badja_pck_matches_reference_golden(golden):
    """F3: metrics.badja_pck against the GENUINE BadjaDataset.pck_evaluate (badja_dataset.py:438-583, lifted by AST: gen_badja_pck) on two
    synthetic animals: silhouette-area thresholds, (y, x) joints, visibility flags, an unlabelled frame; and the per-video mean of
    PCK@0.2 that the reference writes out as 'PCK@0.1 AVG'."""
    g = golden("badja_pck")
    n = int(g["n_videos"])
    preds, joints, vis, segs = [], [], [], []
    for v in range(n):
        lab = g[f"labelled{v}"]
        preds.append(g[f"pred{v}"])
        joints.append([g[f"joints{v}"][t] if lab[t] else None for t in range(len(lab))])
        vis.append([g[f"visible{v}"][t] if lab[t] else None for t in range(len(lab))])
        segs.append(list(g[f"segs{v}"]))
    got = M.badja_pck(preds, joints, vis, segs)
    for r in (0.1, 0.2, 0.3, 0.4):
        assert abs(got[f"PCK@{r}"] - float(g[f"PCK_at_{r}"])) < 1e-9, (r, got[f"PCK@{r}"], float(g[f"PCK_at_{r}"]))
    assert abs(got["PCK@0.2 per-video mean"] - float(g["per_video_mean_pck02"])) < 1e-9
    assert 5.0 < got["PCK@0.1"] < got["PCK@0.4"] < 100.0


def test_jhmdb_pck():
    rng = np.random.default_rng(1)
    gt = [rng.random((2, 15, 9)) * 100 + 10 for _ in range(3)]
    assert all(abs(v - 100.0) < 1e-12 for v in M.jhmdb_pck([g.copy() for g in gt], gt).values())
    # hand-checkable case: one joint pair, bbox diagonal 50 -> normaliser 30; errors 3 and 12 -> PCK@0.1 = 50, @0.5 = 100
    g = np.array([[[10.0], [40.0]], [[10.0], [50.0]]])            # (2,J=2,T=1): joints (10,10) and (40,50)
    p = g.copy()
    p[0, 0, 0] += 3.0
    p[1, 1, 0] += 12.0
    r = M.jhmdb_pck([p], [g])
    assert abs(r["PCK@0.1"] - 50.0) < 1e-9 and abs(r["PCK@0.5"] - 100.0) < 1e-9
    # a joint predicted at x <= 0 is ignored, also in the bbox normaliser (jhmdb_dataset.py:219-231)
    g3 = np.array([[[10.0], [40.0], [70.0]], [[10.0], [50.0], [20.0]]])
    p3 = g3.copy()
    p3[:, 2, 0] = -1
    assert abs(M.jhmdb_pck([p3], [g3])["PCK@0.1"] - 100.0) < 1e-9


def _write_fake_jhmdb(root, n_videos=2, T=6, size=(96, 128), seed=0):
    """A JHMDB-format tree (list file, PNG frames, pos_img .mat, 1-based) holding rigidly translating textures whose 15 'joints'
    move with the texture: the tracker must follow them."""
    import numpy as np
    import scipy.io as sio
    from PIL import Image
    rng = np.random.default_rng(seed)
    h, w = size
    lines = []
    for v in range(n_videos):
        pad = 3 * T
        base = rng.integers(0, 255, (h + 2 * pad, w + 2 * pad, 3)).astype(np.uint8)
        base = np.asarray(Image.fromarray(base).resize((w + 2 * pad, h + 2 * pad), Image.BILINEAR))       # keep it an image
        small = Image.fromarray(base).resize(((w + 2 * pad) // 6, (h + 2 * pad) // 6), Image.BILINEAR)
        base = np.asarray(small.resize((w + 2 * pad, h + 2 * pad), Image.BICUBIC))                         # smooth texture
        vx, vy = int(rng.integers(-2, 3)), int(rng.integers(-2, 3))
        vdir = os.path.join(root, "JHMDB", "Rename_Images", f"vid{v}")
        os.makedirs(vdir, exist_ok=True)
        for t in range(T):
            Image.fromarray(base[pad - vy * t: pad - vy * t + h, pad - vx * t: pad - vx * t + w]).save(os.path.join(vdir, f"{t + 1:05d}.png"))
        x0 = rng.uniform(20, w - 20, 15)
        y0 = rng.uniform(20, h - 20, 15)
        ts = np.arange(T)
        pos = np.stack([x0[:, None] + vx * ts[None], y0[:, None] + vy * ts[None]], 0) + 1.0               # (2,15,T), 1-based
        adir = os.path.join(root, "JHMDB", "joint_positions", f"vid{v}")
        os.makedirs(adir, exist_ok=True)
        sio.savemat(os.path.join(adir, "joint_positions.mat"), {"pos_img": pos})
        lines.append(f"JHMDB/joint_positions/vid{v}/joint_positions.mat JHMDB/Rename_Images/vid{v}")
    with open(os.path.join(root, "val_list.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def test_jhmdb_adapter_sample_format_and_pck(tmp_path):
    """F3: JHMDB files -> (rgbs, query_points, trajectories, visibilities); predictions map back to (2,15,T) at the video's
    own resolution; a perfect prediction scores PCK 100, one shifted by 0.2 x the normaliser scores 0 at 0.1 and 100 at 0.3."""
    from fgvc_amd import datasets, metrics
    _write_fake_jhmdb(str(tmp_path))
    ds = datasets.JhmdbPoses(str(tmp_path), split="val", input_size=(64, 80))
    assert len(ds) == 2
    sample, meta = ds[0]
    assert sample["rgbs"].shape == (1, 6, 3, 64, 80) and sample["query_points"].shape == (1, 15, 3)
    assert sample["trajectories"].shape == (1, 6, 15, 2) and sample["visibilities"].shape == (1, 6, 15)
    assert float(sample["query_points"][0, :, 0].abs().max()) == 0.0
    assert torch.allclose(sample["query_points"][0, :, 1:], sample["trajectories"][0, 0])
    assert meta["original_shape"] == (96, 128) and meta["gt_poses"].shape == (2, 15, 6)
    # joints were written 1-based at (x0, y0): back at the original resolution they are 0-based pixel coordinates
    back = ds.pose_prediction(meta, sample["trajectories"])
    assert np.allclose(back, meta["gt_poses"], atol=1e-4)
    assert metrics.jhmdb_pck([back], [meta["gt_poses"]])["PCK@0.1"] == 100.0
    gt = meta["gt_poses"]
    box = 0.6 * np.linalg.norm(gt.max(axis=1) - gt.min(axis=1), axis=0)                                  # (T,)
    shifted = gt + np.stack([0.2 * box, np.zeros_like(box)], 0)[:, None, :]
    r = metrics.jhmdb_pck([shifted], [gt])
    assert r["PCK@0.1"] == 0.0 and r["PCK@0.3"] == 100.0


def test_summaries_and_result_files(tmp_path):
    """F4: the per-point records and files of TAPVidDataset.tapvid_evaluate / save_results (tapvid.py:235-350)."""
    import json
    import pickle
    import pandas as pd
    from fgvc_amd import metrics
    s = metrics.trajectory_summary([[0, 0], [1, 1], [2, 2]], [[0, 0], [2, 2], [3, 3]], [1, 1, 0], [1, 1, 1], [0, 0, 0], idx="123--2--31")
    # docstring known answers of the reference's compute_summary (figures.py:225-246); TAP-Vid numbers are x100 (:289)
    want = {"idx": "123--2--31", "ade": 0.9428090453147888, "ade_visible": 0.7071067690849304, "ade_visible_chain": 0.7071067690849304,
            "n_timesteps": 3, "n_timesteps_visible": 2, "n_timesteps_visible_chain": 2, "occlusion_accuracy": 50.0, "jaccard_1": 0.0,
            "jaccard_2": 50.0, "jaccard_16": 50.0, "average_jaccard": 40.0, "pts_within_1": 0.0, "pts_within_2": 100.0,
            "average_pts_within_thresh": 80.0}
    for k, v in want.items():
        assert s[k] == v if not isinstance(v, float) else abs(s[k] - v) < 1e-5, (k, s[k], v)
    g = torch.Generator().manual_seed(3)
    results = []
    for _ in range(2):
        traj = torch.rand(1, 5, 3, 2, generator=g) * 256
        vis = torch.ones(1, 5, 3)
        results.append((traj, vis, traj + 1.5, torch.zeros(1, 5, 3), torch.cat([torch.zeros(1, 3, 1), traj[:, 0]], -1)))
    summaries, results_list = metrics.tapvid_summaries(results, "first", input_size=(256, 256), size=(256, 256))
    assert len(summaries) == 6 and summaries[4]["idx"] == "1--0--1" and results_list[4]["idx"] == "1_1"
    assert "pts_within_0.01" in summaries[0] and "pts_within_10" in summaries[0]
    paths = metrics.save_results(summaries, results_list, str(tmp_path / "out"), {"dataset": "davis", "query_mode": "first"})
    assert sorted(os.path.basename(p) for p in paths.values()) == ["results_dfdavis.csv", "results_listdavis.pkl", "summariesdavis.json"]
    assert json.load(open(paths["summaries"])) == json.loads(json.dumps(summaries))
    df = pd.read_csv(paths["results_df"], index_col=0)
    assert list(df.columns) == list(summaries[0].keys()) and len(df) == 6
    assert len(pickle.load(open(paths["results_list"], "rb"))) == 6


def _write_fake_badja(root, n_videos=2, T=6, size=(120, 160), seed=0, skip_label=3):
    """A BADJA-format tree (joint_annotations/*.json, JPEG frames, PNG silhouettes) of rigidly translating textures whose 37 'SMAL
    joints' move with the texture; frame `skip_label` of every video has no annotation record (as most BADJA frames have none)."""
    import json
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    h, w = size
    os.makedirs(os.path.join(root, "joint_annotations"), exist_ok=True)
    for v in range(n_videos):
        pad = 3 * T
        base = rng.integers(0, 255, ((h + 2 * pad) // 6, (w + 2 * pad) // 6, 3)).astype(np.uint8)
        base = np.asarray(Image.fromarray(base).resize((w + 2 * pad, h + 2 * pad), Image.BICUBIC))
        vx, vy = int(rng.integers(-2, 3)), int(rng.integers(-2, 3))
        animal = f"animal{v}"
        os.makedirs(os.path.join(root, "JPEGImages/Full-Resolution", animal), exist_ok=True)
        os.makedirs(os.path.join(root, "Annotations/Full-Resolution", animal), exist_ok=True)
        x0, y0 = rng.uniform(20, w - 20, 37), rng.uniform(20, h - 20, 37)
        records = []
        for t in range(T):
            fr = 7 + t                                                   # frame numbers need not start at 0
            Image.fromarray(base[pad - vy * t: pad - vy * t + h, pad - vx * t: pad - vx * t + w]).save(
                os.path.join(root, "JPEGImages/Full-Resolution", animal, f"{fr:05d}.jpg"), quality=97)
            sil = np.zeros((h, w), np.uint8)
            sil[h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 255             # area = h w / 4
            Image.fromarray(sil).save(os.path.join(root, "Annotations/Full-Resolution", animal, f"{fr:05d}.png"))
            if t == skip_label:
                continue
            vis = np.ones(37, int)
            vis[9] = 0                                                   # SMAL joint 9 (the 2nd annotated one) is never visible
            records.append(dict(image_path=f"BADJA/JPEGImages/Full-Resolution/{animal}/{fr:05d}.jpg",
                                segmentation_path=f"BADJA/Annotations/Full-Resolution/{animal}/{fr:05d}.png",
                                joints=np.stack([y0 + vy * t, x0 + vx * t], 1).tolist(), visibility=vis.tolist()))
        with open(os.path.join(root, "joint_annotations", f"{animal}.json"), "w") as f:
            json.dump(records, f)


def test_badja_adapter_sample_format_and_pck(tmp_path):
    """BADJA files -> (rgbs, query_points, trajectories, visibilities) + what the metric needs; PCK as badja_dataset.py:451-571:
    visible joints of labelled frames only, distance < ratio * sqrt(silhouette area), strict; the per-video PCK@0.2 mean beside it."""
    from fgvc_amd import datasets, metrics
    _write_fake_badja(str(tmp_path))
    ds = datasets.BadjaPoses(str(tmp_path), size=(60, 80))
    assert len(ds) == 2
    sample, meta = ds[0]
    J = len(datasets.BADJA_ANNOTATED)
    assert J == 20 and sample["rgbs"].shape == (1, 6, 3, 60, 80) and sample["query_points"].shape == (1, J, 3)
    assert sample["trajectories"].shape == (1, 6, J, 2) and sample["visibilities"].shape == (1, 6, J)
    assert float(sample["query_points"][0, :, 0].abs().max()) == 0.0
    assert torch.allclose(sample["query_points"][0, :, 1:], sample["trajectories"][0, 0])
    assert meta["joints"][3] is None and float(sample["visibilities"][0, 3].sum()) == 0.0          # the unlabelled frame
    assert float(sample["visibilities"][0, 0, 1]) == 0.0 and float(sample["visibilities"][0, 0].sum()) == J - 1
    assert meta["segs"][0].shape == (60, 80) and int((meta["segs"][0] > 0).sum()) == 30 * 40
    # a perfect prediction: every visible joint of every labelled frame is correct at every ratio
    perfect = ds.pose_prediction(sample["trajectories"])
    assert perfect.shape == (2, J, 6)
    r = metrics.badja_pck([perfect], [meta["joints"]], [meta["visibles"]], [meta["segs"]])
    assert r["PCK@0.1"] == r["PCK@0.4"] == r["PCK@0.2 per-video mean"] == 100.0
    # shifted in x by 0.25 * sqrt(area): wrong at 0.1 and 0.2, right at 0.3 and 0.4; exactly AT the threshold is wrong (strict <)
    thr = np.sqrt(30 * 40)
    shifted = perfect.copy(); shifted[0] += 0.25 * thr
    r = metrics.badja_pck([shifted], [meta["joints"]], [meta["visibles"]], [meta["segs"]])
    assert r["PCK@0.1"] == r["PCK@0.2"] == 0.0 and r["PCK@0.3"] == r["PCK@0.4"] == 100.0
    # the invisible joint and the unlabelled frame are not counted: moving them far away changes nothing
    far = perfect.copy(); far[:, 1, :] += 1000.0; far[:, :, 3] -= 1000.0
    assert metrics.badja_pck([far], [meta["joints"]], [meta["visibles"]], [meta["segs"]])["PCK@0.1"] == 100.0
    # two videos: overall PCK pools the joints, the per-video number averages the videos
    s1, m1 = ds[1]
    p1 = ds.pose_prediction(s1["trajectories"]); p1[0, :10] += 0.25 * thr                            # half of video 1's joints off at 0.2
    r = metrics.badja_pck([perfect, p1], [meta["joints"], m1["joints"]], [meta["visibles"], m1["visibles"]], [meta["segs"], m1["segs"]])
    n_vis = 5 * (J - 1)                                                                              # counted joints per video
    wrong = 5 * 9                                                                                    # joints 0..9 minus the invisible one, 5 frames
    assert abs(r["PCK@0.2"] - 100.0 * (2 * n_vis - wrong) / (2 * n_vis)) < 1e-9
    assert abs(r["PCK@0.2 per-video mean"] - 0.5 * (100.0 + 100.0 * (n_vis - wrong) / n_vis)) < 1e-9
