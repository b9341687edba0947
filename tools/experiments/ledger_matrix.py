"""Which half costs the reference's lists?  

Every encoder arithmetic x every pair kernel on the two 256 x 256 fixtures of the genuine forward_test: the merged top-10 lists of the
512 sampled queries against what the reference's own `topk` returned (tests/test_oracle.py: ledger_topk, _cfg0_compare_topk).
Pair kernels: fgvc_pair_topk_f16f6, fgvc_pair_topk_f16x3, fgvc_pair_topk_f32, and "f16f6+rescore": the candidates of the f16f6 lists
re-scored in float64 from the encoder's f32 rows (what an exact re-scoring pass behind the f16f6 kernel could reach at best).

    python tools/experiments/ledger_matrix.py [out.json]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fgvc_amd import engine, ops  # noqa: E402
from oracle import fgvc_oracle as O  # noqa: E402
from tests.test_oracle import _clip8, ledger_topk  # noqa: E402
from tests.test_gpu_api import _cfg0_ledger, _tracker, T  # noqa: E402

dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")
CFG = dict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, step=512, with_first=True, with_first_neighbor=True, batch_step=4)


def rescore(pl, feats, plan, row, sample, HW, k=10):
    """pair lists (approximate scores) -> merged top-k of the sampled queries with every candidate re-scored in float64 from `feats`"""
    slots = plan.slot_pair[row]
    f64 = feats.double()
    qf = None
    cand_idx, cand_sc = [], []
    for t, pid in enumerate(slots):
        if pid < 0:
            continue
        qframe, kframe, _ = plan.pairs[pid]
        ci = pl.idx[pid][sample].long()                       # (S, k) key pixels
        q = f64[qframe][sample]                               # (S, C)
        kk = f64[kframe][ci.clamp(min=0)]                     # (S, k, C)
        sc = torch.einsum("sc,skc->sk", q, kk)
        sc = torch.where(ci >= 0, sc, torch.full_like(sc, float("-inf")))
        cand_idx.append(ci + t * HW)
        cand_sc.append(sc)
    ci, sc = torch.cat(cand_idx, 1), torch.cat(cand_sc, 1)
    # canonical order: score desc, index asc
    order = torch.sort(ci, dim=1, stable=True)[1]
    ci, sc = ci.gather(1, order), sc.gather(1, order)
    order = torch.sort(sc, dim=1, descending=True, stable=True)[1][:, :k]
    return ci.gather(1, order).cpu().numpy(), (sc.gather(1, order) / 0.07).float().cpu().numpy()


def main():
    out = {}
    for fixture in ("tracker_8x256x256", "tracker_cfg0_2x256x256"):
        g = dict(np.load(os.path.join(GOLD, fixture + ".npz"), allow_pickle=True))
        model = _tracker(dev, "VanillaTracker", (1, 1, 1, 4), CFG, int(g["seed"]))
        if fixture == "tracker_8x256x256":
            rgbs, nT, last = _clip8(g).to(dev), 8, 7
        else:
            rgbs, nT, last = (T(g["rgbs_i8"]).float() / 32.0).unsqueeze(0).to(dev), 2, 1
        sample = T(g["sample"]).long().to(dev)
        rep = {}
        for arith in ("f16f6", "f16f8", "bf16x3", "f16x3", "miopen_f32"):
            if arith == "miopen_f32":                          # every convolution through MIOpen in f32 (ResNet.use_split_conv = False)
                model.backbone.use_split_conv = False
                model.backbone.reset_split_cache()
            else:
                model.backbone.set_arith(arith)
            feats, Hf, Wf = model.get_feats_hwc(rgbs[0], split=False)
            assert feats.dtype == torch.float32 and feats.shape[-1] == 256, (feats.dtype, feats.shape)
            HW = Hf * Wf
            for pk in ("f16f6", "f16x3", "f32", "f16f6+rescore"):
                ecfg = engine.TrackerConfig.from_test_cfg(model.test_cfg)
                ecfg.pair_precision = "f32" if pk == "f32" else "split"
                ecfg.pair_split_fmt = "f16" if pk == "f16x3" else "f16f6"
                plan = engine.plan_clip(nT, [0], ecfg)
                row = plan.out_rows[(0, last)]
                pl = engine.run_pairs(feats, Hf, Wf, plan, ecfg)
                if pk == "f16f6+rescore":
                    idx, logit = rescore(pl, feats, plan, row, sample, HW)
                else:
                    tk = engine.merge_pairs(pl, ecfg)
                    idx, logit = tk.idx[row][sample].cpu().numpy(), tk.logit[row][sample].cpu().numpy()
                assert not ops.pair_f16x3_timed_out()
                if fixture == "tracker_8x256x256":
                    led = ledger_topk(g, idx, logit)
                    led = {k: v for k, v in led.items()}
                else:
                    led = _cfg0_ledger(g, idx)
                rep[f"enc={arith} pair={pk}"] = led
                brief = {k: v for k, v in led.items() if k != "mismatches"}
                print(fixture, f"enc={arith:10s} pair={pk:14s}", brief, flush=True)
        out[fixture] = rep
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_ledger_matrix.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__" and "traj" not in sys.argv:
    main()


def traj_matrix():
    """tracker_4x64x64 (32 x 32 features): trajectory error of the clear read-outs per (encoder arithmetic, pair kernel)"""
    from tests.test_gpu_api import _readout_gaps, _traj_err
    g = dict(np.load(os.path.join(GOLD, "tracker_4x64x64.npz"), allow_pickle=True))
    cfg = dict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, step=512, with_first=True, with_first_neighbor=True)
    gaps = _readout_gaps(g, dict(cfg), 64, 64)
    rgbs, qp, traj, vis = (T(g[n]).to(dev) for n in ("rgbs", "query_points", "trajectories", "visibilities"))
    for pk in ("f16f6", "f16", "f32"):
        c = dict(cfg)
        if pk == "f32":
            c["pair_precision"] = "f32"
        else:
            c["pair_split_fmt"] = pk
        model = _tracker(dev, "VanillaTracker", (1, 1, 1, 4), c, int(g["seed"]))
        for arith in ("f16f6", "f16f8", "bf16x3", "f16x3"):
            model.backbone.set_arith(arith)
            outs = model(test_mode=True, rgbs=rgbs, query_points=qp, trajectories=traj, visibilities=vis)
            m, near = _traj_err(outs[2], T(g["out_traj_pred"]), gaps, 1.0, arith, skip=[(1, 2)])
            print(f"tracker_4x64x64 enc={arith:7s} pair={pk:6s} clear read-outs max {m:.3e} px; near-tie {near}", flush=True)


if __name__ == "__main__" and "traj" in sys.argv:
    traj_matrix()
