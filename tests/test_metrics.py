"""CPU: evaluation arithmetic against the reference's function output (golden) and the
reference's docstring known answers (figures.py:225-246)."""
import numpy as np

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
