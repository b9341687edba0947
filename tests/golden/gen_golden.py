"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

    python tests/golden/gen_golden.py

Imports the genuine operator files from /root/reference through
oracle/ref_import.py, feeds them seeded inputs and stores inputs + the
reference's outputs as small .npz fixtures next to this script.  The fixtures
are data only; no reference source travels.  While a reference operator runs,
`torch.Tensor.topk` is spied on so the (values, indices) the reference's own
top-k call produced are recorded too.

Recorded torch version: see `meta.json` (tie order of torch.topk and GEMM rounding
may differ between torch versions; the reference pins torch 1.9.1).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import fgvc_oracle as O  # noqa: E402  (only for seeded_resnet_state = input generation)


class TopkSpy:
    def __enter__(self):
        self.calls = []
        self._orig = torch.Tensor.topk
        spy = self

        def topk(t, *a, **k):
            r = spy._orig(t, *a, **k)
            spy.calls.append((r[0].clone(), r[1].clone()))
            return r

        torch.Tensor.topk = topk
        return self

    def __exit__(self, *exc):
        torch.Tensor.topk = self._orig
        return False


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"  {name}.npz  " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in out.items()))


def rnd(g, *shape):
    return torch.randn(*shape, generator=g)


def main():
    ref = ref_import.load()
    torch.set_num_threads(8)
    meta = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}

    # ---- A4 spatial_neighbor ------------------------------------------------
    for (H, W, nr, mode) in [(8, 12, 6, "circle"), (16, 16, 30, "circle"), (9, 7, 5, "square"), (20, 24, 14, "circle")]:
        m = ref.spatial_neighbor(1, H, W, neighbor_range=nr, device="cpu", dtype=torch.float32, mode=mode)
        m = m.reshape(H * W, H * W)
        save(f"mask_{mode}_{H}x{W}_r{nr}", H=H, W=W, nr=nr, packed=np.packbits(m.numpy().astype(np.uint8), axis=None),
             count=int(m.sum()))

    # ---- A5 masked_attention_efficient (+ v2 twin, + reference's own top-k) --
    cases = [
        # name,      C,  T, H,  W,  P, nr, k,  step, nml, mode
        ("mae_s8x12", 16, 2, 8, 12, 3, 6, 5, 32, 0, "softmax"),
        ("mae_s16x16", 64, 6, 16, 16, 5, 30, 10, 128, 0, "softmax"),
        ("mae_s32x32", 64, 3, 32, 32, 2, 14, 10, 512, 0, "softmax"),
        ("mae_s20x24_nml1", 32, 3, 20, 24, 4, 14, 10, 100, 1, "softmax"),
        ("mae_s12x20_cos", 32, 2, 12, 20, 3, 10, 4, 64, 0, "cosine"),
        ("mae_s16x24_c256", 256, 3, 16, 24, 5, 30, 10, 512, 0, "softmax"),
    ]
    for i, (name, C, T, H, W, P, nr, k, step, nml, mode) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + i)
        q, key, v = rnd(g, 1, C, H, W), rnd(g, 1, C, T, H, W), torch.rand(1, P, T, H, W, generator=g)
        if name == "mae_s16x16":          # duplicate frame 0 in slots 0 and 1 like the tracker does (:353-362)
            key[:, :, 1] = key[:, :, 0]
            v[:, :, 1] = v[:, :, 0]
        mask = ref.spatial_neighbor(1, H, W, neighbor_range=nr, device="cpu", dtype=torch.float32)
        with TopkSpy() as spy:
            out = ref.masked_attention_efficient(q, key, v, mask, temperature=0.07, topk=k, step=step,
                                                 non_mask_len=nml, mode=mode)
        tv = torch.cat([c[0][0] for c in spy.calls], dim=1)       # (k, HW)
        ti = torch.cat([c[1][0] for c in spy.calls], dim=1)
        out2 = ref.masked_attention_efficient_v2(q, key, v, nr // 2, temperature=0.07, topk=k, step=step,
                                                 non_mask_len=0, mode=mode) if nml == 0 else out
        save(name, query=q, key=key, value=v, nr=nr, topk=k, step=step, non_mask_len=nml,
             mode=np.array(mode), temperature=0.07, out=out, out_v2=out2,
             ref_topk_val=tv.t().contiguous(), ref_topk_idx=ti.t().contiguous().to(torch.int32))
        meta["cases"][name] = dict(C=C, T=T, H=H, W=W, P=P, nr=nr, topk=k)

    # ---- no-mask (full-frame) variant ---------------------------------------
    g = torch.Generator().manual_seed(200)
    q, key, v = rnd(g, 1, 32, 10, 14), rnd(g, 1, 32, 2, 10, 14), torch.rand(1, 3, 2, 10, 14, generator=g)
    with TopkSpy() as spy:
        out = ref.masked_attention_efficient(q, key, v, None, temperature=0.07, topk=10, step=64)
    save("mae_nomask_10x14", query=q, key=key, value=v, topk=10, temperature=0.07, out=out,
         ref_topk_val=torch.cat([c[0][0] for c in spy.calls], 1).t().contiguous(),
         ref_topk_idx=torch.cat([c[1][0] for c in spy.calls], 1).t().contiguous().to(torch.int32))

    # ---- A5'' dense volume formulations --------------------------------------
    g = torch.Generator().manual_seed(300)
    q, key = rnd(g, 1, 32, 9, 11), rnd(g, 1, 32, 2, 9, 11)
    v = torch.rand(1, 3, 2, 9, 11, generator=g)
    mask = ref.spatial_neighbor(1, 9, 11, neighbor_range=8, device="cpu", dtype=torch.float32)
    out = ref.masked_attention(q, key, v, mask, temperature=0.07, topk=5, step=40)
    aff = ref.compute_affinity(key[:, :, 0], q, temperature=0.07)          # (1, HWsrc, HWdst)
    att = ref.non_local_attention(q, key.transpose(1, 2), temprature=0.07, norm=True, att_only=True)
    save("dense_9x11", query=q, key=key, value=v, nr=8, out_masked_attention=out,
         compute_affinity=aff[0], non_local_att=att[0])

    # ---- A6 coarse-to-fine ----------------------------------------------------
    g = torch.Generator().manual_seed(400)
    H, W, s, T, C, Cf, P, Rf = 8, 10, 4, 2, 32, 16, 3, 3
    q, key = rnd(g, 1, C, H, W), rnd(g, 1, C, T, H, W)
    qf, kf = rnd(g, 1, Cf, H * s, W * s), rnd(g, 1, Cf, T, H * s, W * s)
    v = torch.rand(1, P, T, H * s, W * s, generator=g)
    mask = ref.spatial_neighbor(1, H, W, neighbor_range=8, device="cpu", dtype=torch.float32)
    with TopkSpy() as spy:
        out = ref.masked_attention_efficient_c2f(q, key, qf, kf, v, mask, temperature=0.07, topk=5, step=32,
                                                 radius_fine=Rf)
    out_cos = ref.masked_attention_efficient_c2f(q, key, qf, kf, v, mask, temperature=0.07, topk=5, step=32, radius_fine=Rf,
                                                 mode="cosine")                      # clamp(affinity, 0)^2 weights (:860-861)
    save("c2f_8x10", query=q, key=key, query_fine=qf, key_fine=kf, value=v, nr=8, topk=5, radius_fine=Rf,
         temperature=0.07, out=out, out_cos=out_cos,
         ref_topk_val=torch.cat([c[0][0] for c in spy.calls], 1).t().contiguous(),
         ref_topk_idx=torch.cat([c[1][0] for c in spy.calls], 1).t().contiguous().to(torch.int32))

    # ---- A7' local-window correlation (torch-only twin of mmcv Correlation) ---
    g = torch.Generator().manual_seed(500)
    H, W, K, C, P, R = 10, 12, 3, 32, 4, 3
    qframe, kframes = rnd(g, 1, C, H, W), rnd(g, 1, C, K, H, W)
    v = torch.rand(1, P, K, H, W, generator=g)
    with TopkSpy() as spy:
        out = ref.masked_attention_efficient_correlation_v2(qframe, kframes, v, R, None, lambda x: x,
                                                            temperature=0.07, topk=6, sstep=50, tstep=2)
    save("localcorr_10x12", query=qframe, key=kframes, value=v, radius=R, topk=6, temperature=0.07, out=out,
         ref_topk_val=torch.cat([c[0][0] for c in spy.calls], 1).t().contiguous(),
         ref_topk_idx=torch.cat([c[1][0] for c in spy.calls], 1).t().contiguous().to(torch.int32))

    # ---- A1-A3, A8-A10 the whole tracker ---------------------------------------
    cfg = ref.ConfigDict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, step=512,
                         with_first=True, with_first_neighbor=True)
    model = ref.builder.build_model(
        dict(type="VanillaTracker", backbone=dict(type="ResNet", depth=18, strides=(1, 1, 1, 4),
                                                  out_indices=(2,), pool_type="none")),
        train_cfg=None, test_cfg=cfg)
    sd = O.seeded_resnet_state(seed=7, strides=(1, 1, 1, 4), pool_type="none")
    missing = model.backbone.load_state_dict(sd, strict=True)
    model.eval()
    g = torch.Generator().manual_seed(600)
    T, h, w = 4, 64, 64
    rgbs = rnd(g, 1, T, 3, h, w)
    qp = torch.tensor([[[0., 10., 20.], [1., 33.5, 12.25], [0., 50., 40.], [2., 5., 60.], [1., 20., 20.]]])
    P = qp.shape[1]
    traj_gt = torch.rand(1, T, P, 2, generator=g) * 64
    vis_gt = (torch.rand(1, T, P, generator=g) > 0.3).float()
    with ref_import.cuda_as_cpu(), torch.no_grad():
        outs = model(test_mode=True, rgbs=rgbs, query_points=qp, trajectories=traj_gt, visibilities=vis_gt)
        feats = model.backbone(rgbs[0])
        # un-regrouped main path for the t=0 points only
        main = model.forward_test_main(rgbs, qp[:, [0, 2]], torch.zeros(1, T, 2, 2), torch.zeros(1, T, 2))
    wsum = float(sum(v.double().abs().sum() for k, v in sd.items() if v.dtype.is_floating_point))
    save("tracker_4x64x64", rgbs=rgbs, query_points=qp, trajectories=traj_gt, visibilities=vis_gt,
         seed=7, weight_abs_sum=wsum,
         out_trajectories=outs[0], out_visibilities=outs[1], out_traj_pred=outs[2], out_vis_pred=outs[3],
         out_query_points=outs[4], feats_sub=feats[:, ::16, ::4, ::4], feats_abs_sum=float(feats.double().abs().sum()),
         main_traj_pred=main[2])

    # ---- img2coord and gaussian maps on their own ------------------------------
    g = torch.Generator().manual_seed(700)
    maps = torch.rand(3, 4, 12, 16, generator=g).numpy().astype(np.float32)
    maps[1, 2] = 0.0
    coords = model.img2coord(maps, 4)
    with ref_import.cuda_as_cpu():
        grid, _, stride = model.get_coords_grid((1, 24, 32), (12, 16))
        gfull, gres = model.draw_gaussion_map_online(torch.tensor([[[3.0, 4.0], [10.5, 7.25]]]), grid, stride=stride)
    save("readout_small", maps=maps, coords=coords, gauss_points=np.array([[3.0, 4.0], [10.5, 7.25]], np.float32),
         gauss_full=gfull[0], gauss_res=gres[0], stride=stride)

    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print(f"total fixture bytes: {tot}")



def gen_tapvid_metrics():
    """tests/golden/tapvid_metrics.npz: the reference's compute_tapvid_metrics (numpy only) is lifted out of its
    module by AST (the module imports mediapy/absl, absent here) and run on seeded inputs."""
    import ast
    from typing import Iterable, Mapping
    src = open(os.path.join(ref_import.REF_ROOT, "mmpt/datasets/tapvid_evaluation_datasets.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "compute_tapvid_metrics"][0]
    ns = {"np": np, "Iterable": Iterable, "Mapping": Mapping}
    exec(compile(ast.Module([fn], []), "ref", "exec"), ns)
    rng = np.random.default_rng(0)
    b, n, T = 3, 7, 12
    gt_occ = rng.random((b, n, T)) < 0.3
    gt_occ[:, :, 0] = False
    qp = np.zeros((b, n, 3))
    qp[..., 0] = rng.integers(0, 4, (b, n))
    for i in range(b):
        for j in range(n):
            gt_occ[i, j, int(qp[i, j, 0])] = False
    gt = rng.random((b, n, T, 2)) * 64
    pred = gt + rng.normal(0, 3, (b, n, T, 2))
    pred_occ = rng.random((b, n, T)) < 0.3
    outs = {}
    for mode in ("first", "strided"):
        m = ns["compute_tapvid_metrics"](qp, gt_occ, gt, pred_occ, pred, mode, additional_pck_thresholds=[0.5, 3])
        outs.update({f"{mode}__{k}": np.asarray(v) for k, v in m.items()})
    save("tapvid_metrics", query_points=qp, gt_occluded=gt_occ, gt_tracks=gt, pred_occluded=pred_occ,
         pred_tracks=pred, **outs)


def gen_hr_tracker():
    """tests/golden/hr_tracker_5x48x64.npz: the genuine HRVanillaTracker driver loops (vanilla_tracker.py:417-660: backward
    warping forward_test_main, the query-time regrouping it inherits, forward warping forward_test_forward) run around the
    mmcv.ops.Correlation stand-in of oracle/ref_import.py (its arithmetic stays "parity unpinned")."""
    ref = ref_import.load()
    g = torch.Generator().manual_seed(800)
    T, h, w = 5, 48, 64
    rgbs = torch.randn(1, T, 3, h, w, generator=g)
    sd = O.seeded_resnet_state(seed=11, strides=(1, 2, 1, 1), pool_type="none")
    out = {}
    # "savemem": save_mem=True pairs ONE key frame (frame key_start, features re-extracted, no first-frame slot) with the label maps
    # of key_start..frame-1 (vanilla_tracker.py:521-545): it runs only when that is one map, i.e. precede_frames = 1
    for tag, extra in (("norm", {}), ("raw", dict(withnorm=False, temperature=4.0)), ("nofirst", dict(with_first=False)),
                       ("savemem", dict(save_mem=True, precede_frames=1))):
        cfg = ref.ConfigDict({**dict(precede_frames=2, topk=6, temperature=0.07, neighbor_range=8, with_first=True,
                                     batch_step=2), **extra})
        model = ref.builder.build_model(
            dict(type="HRVanillaTracker", backbone=dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,),
                                                        pool_type="none")), train_cfg=None, test_cfg=cfg)
        model.backbone.load_state_dict(sd, strict=True)
        model.eval()
        qp0 = torch.tensor([[[0., 10., 20.], [0., 33.5, 12.25], [0., 50., 40.]]])
        with ref_import.cuda_as_cpu(), torch.no_grad():
            main = model.forward_test_main(rgbs, qp0, torch.zeros(1, T, 3, 2), torch.zeros(1, T, 3))
        out[f"main_{tag}"] = main[2]
        if tag == "norm":
            qp = torch.tensor([[[0., 10., 20.], [2., 33.5, 12.25], [0., 50., 40.], [1., 5., 30.]]])
            traj_gt = torch.rand(1, T, 4, 2, generator=g) * 48
            vis_gt = (torch.rand(1, T, 4, generator=g) > 0.3).float()
            ref_yx = torch.tensor([[[20., 12.25, 40.], [10., 33.5, 50.]]])                  # (1, 2, P) = (y, x)
            with ref_import.cuda_as_cpu(), torch.no_grad():
                outs = model(test_mode=True, rgbs=rgbs, query_points=qp, trajectories=traj_gt, visibilities=vis_gt)
                fwd = model.forward_test_forward(rgbs.transpose(1, 2).unsqueeze(1), None, None, ref_yx)
                q, k = model.backbone(rgbs[0, :1]), model.backbone(rgbs[0, 1:2])
                coord = model.get_coord(q, k, (h, w), w // q.shape[-1])
            out.update(query_points=qp, trajectories=traj_gt, visibilities=vis_gt, out_trajectories=outs[0],
                       out_visibilities=outs[1], out_traj_pred=outs[2], out_vis_pred=outs[3], out_query_points=outs[4],
                       ref_yx=ref_yx, forward_coords=np.stack(fwd, 0), coord_field=coord)
    save("hr_tracker_5x48x64", rgbs=rgbs, seed=11, query_points0=qp0, **out)


def gen_extra_modes():
    """tests/golden/mae_l2_12x16.npz, mae_dense_softmax_10x12.npz: the branches of masked_attention_efficient the shipped
    configs do not take -- sim_mode='l2-distance' (local_attention.py:324-327) and topk=None (dense softmax / cosine weights
    over every unmasked key, :376-383)."""
    ref = ref_import.load()
    g = torch.Generator().manual_seed(900)
    C, T, H, W, P, nr, k = 48, 3, 12, 16, 4, 10, 8
    q, key, v = rnd(g, 1, C, H, W), rnd(g, 1, C, T, H, W), torch.rand(1, P, T, H, W, generator=g)
    mask = ref.spatial_neighbor(1, H, W, neighbor_range=nr, device="cpu", dtype=torch.float32)
    with TopkSpy() as spy:
        out = ref.masked_attention_efficient(q, key, v, mask, temperature=0.07, topk=k, step=64, sim_mode="l2-distance")
    # ... with the cosine weights clamp(logit, 0)^2 (:370-371): the logit is (2 k.q - |k|^2) / sqrt(C), so the -|k|^2 shift matters here
    out_cos = ref.masked_attention_efficient(q, key, v, mask, temperature=0.07, topk=k, step=64, sim_mode="l2-distance", mode="cosine")
    # ... and on UN-normalised features, where |k|^2 differs from key to key and changes the ranking (top-k and topk=None forms)
    out_raw = ref.masked_attention_efficient(q, key, v, mask, temperature=0.07, topk=k, step=64, sim_mode="l2-distance", normalize=False)
    out_raw_dense = ref.masked_attention_efficient(q, key, v, mask, temperature=0.07, topk=None, step=64, sim_mode="l2-distance",
                                                   normalize=False)
    save("mae_l2_12x16", query=q, key=key, value=v, nr=nr, topk=k, temperature=0.07, out=out, out_cos=out_cos, out_raw=out_raw,
         out_raw_dense=out_raw_dense,
         ref_topk_val=torch.cat([c[0][0] for c in spy.calls], 1).t().contiguous(),
         ref_topk_idx=torch.cat([c[1][0] for c in spy.calls], 1).t().contiguous().to(torch.int32))
    g = torch.Generator().manual_seed(901)
    C, T, H, W, P, nr = 32, 2, 10, 12, 3, 8
    q, key, v = rnd(g, 1, C, H, W), rnd(g, 1, C, T, H, W), torch.rand(1, P, T, H, W, generator=g)
    mask = ref.spatial_neighbor(1, H, W, neighbor_range=nr, device="cpu", dtype=torch.float32)
    kw = dict(temperature=0.07, topk=None, step=50)
    save("mae_dense_softmax_10x12", query=q, key=key, value=v, nr=nr, temperature=0.07,
         out=ref.masked_attention_efficient(q, key, v, mask, **kw),
         out_nml1=ref.masked_attention_efficient(q, key, v, mask, non_mask_len=1, **kw),
         out_nomask=ref.masked_attention_efficient(q, key, v, None, **kw),
         out_cos=ref.masked_attention_efficient(q, key, v, mask, mode="cosine", **kw),
         out_l2=ref.masked_attention_efficient(q, key, v, mask, sim_mode="l2-distance", **kw))


def gen_tv_keymap():
    """tests/golden/tv_keymap.json: which torchvision ResNet-18 key the genuine ResNet._load_torchvision_checkpoint
    (resnet.py:525-563) copies into each of its own parameters / buffers.  Every tensor of a torchvision-shaped state dict is
    filled with its own index; the loaded model's tensors then name their sources."""
    ref = ref_import.load()
    net = ref.ResNet(depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none")
    own = net.state_dict()
    tv = {}
    def add(name, shape):
        tv[name] = torch.full(shape, float(len(tv) + 1))
    add("conv1.weight", own["conv1.conv.weight"].shape)
    for b in ("weight", "bias", "running_mean", "running_var"):
        add(f"bn1.{b}", (64,))
    tv["bn1.num_batches_tracked"] = torch.tensor(len(tv) + 1)
    for L in range(1, 5):
        for blk in range(2):
            for c in (1, 2):
                add(f"layer{L}.{blk}.conv{c}.weight", own[f"layer{L}.{blk}.conv{c}.conv.weight"].shape)
                n = own[f"layer{L}.{blk}.conv{c}.bn.weight"].shape
                for b in ("weight", "bias", "running_mean", "running_var"):
                    add(f"layer{L}.{blk}.bn{c}.{b}", n)
            if f"layer{L}.{blk}.downsample.conv.weight" in own:
                add(f"layer{L}.{blk}.downsample.0.weight", own[f"layer{L}.{blk}.downsample.conv.weight"].shape)
                n = own[f"layer{L}.{blk}.downsample.bn.weight"].shape
                for b in ("weight", "bias", "running_mean", "running_var"):
                    add(f"layer{L}.{blk}.downsample.1.{b}", n)
    add("fc.weight", (1000, 512)); add("fc.bias", (1000,))
    ids = {float(v.flatten()[0]): k for k, v in tv.items()}
    import logging
    net._load_torchvision_checkpoint(tv, strict=False, logger=logging.getLogger("gen"))
    keymap = {}
    for k, v in net.state_dict().items():
        src = ids.get(float(v.flatten()[0].item())) if v.numel() else None
        keymap[k] = src if (src is not None and bool((v == v.flatten()[0]).all())) else None
    with open(os.path.join(HERE, "tv_keymap.json"), "w") as f:
        json.dump(keymap, f, indent=0, sort_keys=True)
    print(f"  tv_keymap.json  {sum(v is not None for v in keymap.values())} of {len(keymap)} own tensors filled from the checkpoint")


def gen_dense_api():
    """tests/golden/dense_api_10x12.npz: the three dense / window operators of the A5'' row that have no shipped caller --
    `propagate` (affinity_utils.py:33-50, with and without topk), `non_local_attention` (correlation.py:32-83: att_only, per_ref,
    pooled, scaling, mask) and `local_square_attention` (local_attention.py:38-103: whole window, topk, batch_as_context)."""
    ref = ref_import.load()
    g = torch.Generator().manual_seed(1100)
    N, C, H, W, P, T = 2, 32, 10, 12, 5, 3
    HW = H * W
    src, dst = rnd(g, N, C, H, W), rnd(g, N, C, H, W)
    img = torch.rand(N, P, H, W, generator=g)
    aff = ref.compute_affinity(src, dst, temperature=0.07, softmax_dim=1)                 # (N, HW src, HW dst), columns sum to 1
    out = {"src": src, "dst": dst, "img": img,
           "prop": ref.propagate(img, aff.clone()), "prop_top7": ref.propagate(img, aff.clone(), topk=7),
           "prop_raw": ref.propagate(img, ref.compute_affinity(src, dst, temperature=1.0))}
    wide = torch.rand(1, 40, H, W, generator=g)                                            # more than 32 channels: two kernel passes
    out.update(img_wide=wide, prop_wide_top3=ref.propagate(wide, aff[:1].clone(), topk=3))
    Hn, Wn, Tn = 8, 10, 2                                                                  # (a smaller grid: these outputs are dense)
    tar, refs = rnd(g, 1, C, Hn, Wn), rnd(g, 1, Tn, C, Hn, Wn)
    mask = ref.spatial_neighbor(1, Hn, Wn, neighbor_range=6, device="cpu", dtype=torch.float32).reshape(Hn * Wn, Hn * Wn)
    out.update(tar=tar, refs=refs, nl_mask_nr=6,
               nl_att=ref.non_local_attention(tar, refs, temprature=0.07, norm=True, att_only=True),
               nl_att_scaled_masked=ref.non_local_attention(tar, refs, temprature=2.0, scaling=True, mask=mask, att_only=True),
               nl_per_ref=ref.non_local_attention(tar, [refs[:, t] for t in range(Tn)], temprature=0.07, norm=True)[1],
               nl_pooled=ref.non_local_attention(tar, refs, per_ref=False, temprature=0.07, norm=True)[1],
               nl_first=int(ref.non_local_attention(tar, refs, temprature=0.07, norm=True)[0]))
    q, k = rnd(g, N, C, H, W) * 0.3, rnd(g, N, C, H, W) * 0.3
    v = torch.rand(N, P, H, W, generator=g)
    out.update(lq=q, lk=k, lv=v,
               lsa_all=ref.local_square_attention(q, k, v, 5, temperature=0.5),
               lsa_rect=ref.local_square_attention(q, k, v, (3, 7), temperature=0.5),
               lsa_top4=ref.local_square_attention(q, k, v, 5, temperature=0.5, topk=4),
               lsa_ctx_top6=ref.local_square_attention(q[:1], k, v, 7, temperature=0.5, topk=6, batch_as_context=True),
               lsa_ctx_all=ref.local_square_attention(q[:1], k, v, 3, temperature=0.5, batch_as_context=True))
    save("dense_api_10x12", **out)


def gen_tracker_cfg0():
    """tests/golden/tracker_cfg0_2x256x256.npz: BASELINE.json configs[0] / the reference's only shipped eval geometry
    (configs/eval/res18_d1_eval.py:8,12-22): the genuine VanillaTracker.forward_test on 2 x 256 x 256 frames, ResNet-18 with
    strides (1,1,1,4) -> 128 x 128 x 256 features, neighbor_range 30, top-10, with_first (frame 0 sits in key slots 0 AND 1 of
    frame 1).  Stored: the frames (values on a 1/32 grid, as int8), the trajectories, and for 512 sampled query pixels of frame 1
    what the reference's own `topk` call returned (spied) plus a float64 top-(k + 2) of the same rows computed from the
    REFERENCE's features -- the gaps that say which queries' ranks are clear of f32 rounding."""
    ref = ref_import.load()
    cfg = ref.ConfigDict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, step=512,
                         with_first=True, with_first_neighbor=True)
    model = ref.builder.build_model(
        dict(type="VanillaTracker", backbone=dict(type="ResNet", depth=18, strides=(1, 1, 1, 4),
                                                  out_indices=(2,), pool_type="none")),
        train_cfg=None, test_cfg=cfg)
    sd = O.seeded_resnet_state(seed=13, strides=(1, 1, 1, 4), pool_type="none")
    model.backbone.load_state_dict(sd, strict=True)
    model.eval()
    g = torch.Generator().manual_seed(1300)
    T, h, w, P = 2, 256, 256, 8
    # video-like frames: a smooth field that moves by (3, -2) pixels between the frames, plus noise; values on a 1/32 grid in [-4, 4)
    base = torch.nn.functional.interpolate(torch.randn(1, 3, 40, 40, generator=g), size=(h + 16, w + 16), mode="bicubic", align_corners=False)[0]
    frames = torch.stack([base[:, 8:8 + h, 8:8 + w], base[:, 10:10 + h, 5:5 + w]], 0) * 1.2 + 0.25 * torch.randn(T, 3, h, w, generator=g)
    rgbs_i8 = torch.clamp(torch.round(frames * 32), -128, 127).to(torch.int8)
    rgbs = (rgbs_i8.float() / 32.0).unsqueeze(0)
    qp = torch.cat([torch.zeros(P, 1), torch.rand(P, 2, generator=g) * 200 + 28], 1).unsqueeze(0)
    traj_gt = torch.rand(1, T, P, 2, generator=g) * 256
    vis_gt = (torch.rand(1, T, P, generator=g) > 0.3).float()
    with ref_import.cuda_as_cpu(), torch.no_grad(), TopkSpy() as spy:
        outs = model(test_mode=True, rgbs=rgbs, query_points=qp, trajectories=traj_gt, visibilities=vis_gt)
    with torch.no_grad():
        feats = model.backbone(rgbs[0])                                   # (2, 256, 128, 128)
    HW, k = 128 * 128, 10
    calls = [c for c in spy.calls if c[0].shape[1] == k]
    tv = torch.cat([c[0][0] for c in calls], dim=1)                     # (k, HW): the chunks of frame 1's one attention call
    ti = torch.cat([c[1][0] for c in calls], dim=1)
    assert tv.shape == (k, HW), tv.shape
    sample = torch.cat([torch.tensor([0, 127, HW - 128, HW - 1, 64 * 128 + 64]), torch.randint(0, HW, (507,), generator=g)])
    # float64 top-(k + 2) of the sampled rows from the reference's features: key slots = [frame 0, frame 0] (with_first, idx <= 5)
    fn = torch.nn.functional.normalize(feats.double(), dim=1).flatten(2)          # (2, C, HW)
    aff = (fn[0].t() @ fn[1][:, sample]) / 0.07                                     # (HW keys of frame 0, n)
    ky, kx = torch.arange(HW) // 128, torch.arange(HW) % 128
    inside = ((ky.view(-1, 1) - ky[sample].view(1, -1)) ** 2 + (kx.view(-1, 1) - kx[sample].view(1, -1)) ** 2).double().sqrt() < 15
    aff = aff.masked_fill(~inside, float("-inf"))
    dv, di = aff.topk(7, dim=0)                                                     # 7 DISTINCT pixels: each appears in both slots
    wsum = float(sum(v.double().abs().sum() for kk, v in sd.items() if v.dtype.is_floating_point))
    save("tracker_cfg0_2x256x256", rgbs_i8=rgbs_i8, query_points=qp, trajectories=traj_gt, visibilities=vis_gt, seed=13,
         weight_abs_sum=wsum, out_traj_pred=outs[2], out_query_points=outs[4],
         sample=sample.to(torch.int32), ref_topk_val=tv.t()[sample].contiguous(), ref_topk_idx=ti.t()[sample].contiguous().to(torch.int32),
         f64_distinct_val=dv.t().contiguous(), f64_distinct_idx=di.t().contiguous().to(torch.int32),
         feats_sub=feats[:, ::16, ::8, ::8], feats_abs_sum=float(feats.double().abs().sum()))


def gen_tracker_8f():
    """tests/golden/tracker_8x256x256.npz (round 4): the genuine VanillaTracker.forward_test on EIGHT 256 x 256 frames at the reference's
    eval geometry (128 x 128 x 256 features, radius 15, top-10, precede_frames 5, with_first) -- the last frame's attention call merges
    SIX DISTINCT key frames (0, 2..6), which the two-frame fixture (frame 0 in both slots: one pair) cannot exercise.  The frames come
    from tests/golden/clips.py (integer arithmetic: not stored).  Stored: the trajectories, and for 512 sampled query pixels of frame 7
    what the reference's own `topk` returned (spied) plus a float64 top-12 of the same rows from the REFERENCE's features (the gaps that
    say which queries' ranks are clear of rounding), and a sub-sample of the features."""
    from tests.golden import clips
    ref = ref_import.load()
    cfg = ref.ConfigDict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, step=512,
                         with_first=True, with_first_neighbor=True)
    model = ref.builder.build_model(
        dict(type="VanillaTracker", backbone=dict(type="ResNet", depth=18, strides=(1, 1, 1, 4),
                                                  out_indices=(2,), pool_type="none")),
        train_cfg=None, test_cfg=cfg)
    seed = 17
    sd = O.seeded_resnet_state(seed=seed, strides=(1, 1, 1, 4), pool_type="none")
    model.backbone.load_state_dict(sd, strict=True)
    model.eval()
    Tn, h, w, P = 8, 256, 256, 8
    rgbs = (torch.from_numpy(clips.moving_texture(Tn, h, w, seed=1700)).float() / 32.0).unsqueeze(0)
    g = torch.Generator().manual_seed(1700)
    qp = torch.cat([torch.zeros(P, 1), torch.rand(P, 2, generator=g) * 180 + 38], 1).unsqueeze(0)
    traj_gt = torch.rand(1, Tn, P, 2, generator=g) * 256
    vis_gt = (torch.rand(1, Tn, P, generator=g) > 0.3).float()
    with ref_import.cuda_as_cpu(), torch.no_grad(), TopkSpy() as spy:
        outs = model(test_mode=True, rgbs=rgbs, query_points=qp, trajectories=traj_gt, visibilities=vis_gt)
    with torch.no_grad():
        feats = model.backbone(rgbs[0])                                   # (8, 256, 128, 128)
    HW, k = 128 * 128, 10
    calls = [c for c in spy.calls if c[0].shape[1] == k]
    per_frame = HW // 512
    assert len(calls) == (Tn - 1) * per_frame, len(calls)                # frames 1..7, 32 chunks of 512 queries each
    last = calls[-per_frame:]
    tv = torch.cat([c[0][0] for c in last], dim=1)                       # (k, HW)
    ti = torch.cat([c[1][0] for c in last], dim=1)
    sample = torch.cat([torch.tensor([0, 127, HW - 128, HW - 1, 64 * 128 + 64]), torch.randint(0, HW, (507,), generator=g)])
    slots = [0, 2, 3, 4, 5, 6]                                           # key slots of frame 7 (vanilla_tracker.py:353-362)
    fn = torch.nn.functional.normalize(feats.double(), dim=1).flatten(2)  # (8, C, HW)
    ky, kx = torch.arange(HW) // 128, torch.arange(HW) % 128
    inside = ((ky.view(-1, 1) - ky[sample].view(1, -1)) ** 2 + (kx.view(-1, 1) - kx[sample].view(1, -1)) ** 2).double().sqrt() < 15
    aff = torch.cat([((fn[s].t() @ fn[7][:, sample]) / 0.07).masked_fill(~inside, float("-inf")) for s in slots], 0)   # (6 HW, n)
    dv, di = aff.topk(12, dim=0)
    wsum = float(sum(v.double().abs().sum() for kk, v in sd.items() if v.dtype.is_floating_point))
    save("tracker_8x256x256", clip_seed=1700, query_points=qp, trajectories=traj_gt, visibilities=vis_gt, seed=seed,
         weight_abs_sum=wsum, out_traj_pred=outs[2], out_query_points=outs[4],
         sample=sample.to(torch.int32), ref_topk_val=tv.t()[sample].contiguous(), ref_topk_idx=ti.t()[sample].contiguous().to(torch.int32),
         f64_val=dv.t().contiguous(), f64_idx=di.t().contiguous().to(torch.int32),
         feats_sub=feats[:, ::16, ::8, ::8], feats_abs_sum=float(feats.double().abs().sum()))


def gen_tracker_all_queries(trained: bool):
    """tests/golden/tracker_8x256x256_all.npz / tracker_trained_8x256x256.npz (round 6): the genuine VanillaTracker.forward_test on eight
    256 x 256 frames, as gen_tracker_8f, with what the reference's own `topk` returned for ALL 16 384 queries of frame 7 (key pixel as
    (slot, pixel): uint8 + uint16), the float64 top-10 of the same rows from the REFERENCE's features, and per query the smallest
    distance between float64 ranks 1 .. 11 -- the number that says at which resolution a list is decidable.
    trained = False: the weights and clip of tracker_8x256x256.npz (kaiming convolutions, unit BatchNorm statistics, a texture in [-4, 4)).
    trained = True: BatchNorm layers as a trained checkpoint has them (gamma in [0.5, 1.5], beta, running mean and variance: oracle
    seeded_resnet_state(trained_like=True)) and frames inside the range of the reference's Lab normalisation (clips.lab_like)."""
    from tests.golden import clips
    ref = ref_import.load()
    cfg = ref.ConfigDict(precede_frames=5, topk=10, temperature=0.07, neighbor_range=30, step=512,
                         with_first=True, with_first_neighbor=True)
    model = ref.builder.build_model(
        dict(type="VanillaTracker", backbone=dict(type="ResNet", depth=18, strides=(1, 1, 1, 4),
                                                  out_indices=(2,), pool_type="none")),
        train_cfg=None, test_cfg=cfg)
    seed, clip_seed = (23, 2300) if trained else (17, 1700)
    sd = O.seeded_resnet_state(seed=seed, strides=(1, 1, 1, 4), pool_type="none", trained_like=trained)
    model.backbone.load_state_dict(sd, strict=True)
    model.eval()
    Tn, h, w, P = 8, 256, 256, 8
    clip = clips.lab_like(Tn, h, w, seed=clip_seed) if trained else clips.moving_texture(Tn, h, w, seed=clip_seed)
    rgbs = (torch.from_numpy(clip).float() / 32.0).unsqueeze(0)
    g = torch.Generator().manual_seed(clip_seed)
    qp = torch.cat([torch.zeros(P, 1), torch.rand(P, 2, generator=g) * 180 + 38], 1).unsqueeze(0)
    traj_gt = torch.rand(1, Tn, P, 2, generator=g) * 256
    vis_gt = (torch.rand(1, Tn, P, generator=g) > 0.3).float()
    with ref_import.cuda_as_cpu(), torch.no_grad(), TopkSpy() as spy:
        outs = model(test_mode=True, rgbs=rgbs, query_points=qp, trajectories=traj_gt, visibilities=vis_gt)
    with torch.no_grad():
        feats = model.backbone(rgbs[0])                                   # (8, 256, 128, 128)
    HW, k = 128 * 128, 10
    calls = [c for c in spy.calls if c[0].shape[1] == k]
    per_frame = HW // 512
    assert len(calls) == (Tn - 1) * per_frame, len(calls)
    last = calls[-per_frame:]
    tv = torch.cat([c[0][0] for c in last], dim=1).t().contiguous()      # (HW, k)
    ti = torch.cat([c[1][0] for c in last], dim=1).t().contiguous()
    slots = [0, 2, 3, 4, 5, 6]
    fn = torch.nn.functional.normalize(feats.double(), dim=1).flatten(2)  # (8, C, HW)
    keys = torch.stack([fn[s] for s in slots], 0)                          # (6, C, HW)
    ky, kx = torch.arange(HW) // 128, torch.arange(HW) % 128
    f64_idx = torch.empty(HW, k, dtype=torch.int64)
    gap = torch.empty(HW, dtype=torch.float64)
    for q0 in range(0, HW, 512):
        qs = torch.arange(q0, q0 + 512)
        inside = ((ky.view(-1, 1) - ky[qs].view(1, -1)) ** 2 + (kx.view(-1, 1) - kx[qs].view(1, -1)) ** 2).double().sqrt() < 15
        aff = (torch.einsum("sck,cq->skq", keys, fn[7][:, qs]) / 0.07).masked_fill(~inside.unsqueeze(0), float("-inf")).reshape(6 * HW, 512)
        dv, di = aff.topk(k + 1, dim=0)
        f64_idx[qs] = di[:k].t()
        gap[qs] = (dv[:-1] - dv[1:]).min(0).values
    wsum = float(sum(v.double().abs().sum() for kk, v in sd.items() if v.dtype.is_floating_point))
    name = "tracker_trained_8x256x256" if trained else "tracker_8x256x256_all"
    extra = dict(query_points=qp, trajectories=traj_gt, visibilities=vis_gt, out_traj_pred=outs[2], out_query_points=outs[4],
                 feats_sub=feats[:, ::16, ::8, ::8]) if trained else {}
    save(name, clip_seed=clip_seed, seed=seed, trained_like=int(trained), weight_abs_sum=wsum, feats_abs_sum=float(feats.double().abs().sum()),
         ref_slot=(ti // HW).to(torch.uint8), ref_pix=(ti % HW).to(torch.int32).numpy().astype("uint16"), ref_val=tv,
         f64_slot=(f64_idx // HW).to(torch.uint8), f64_pix=(f64_idx % HW).to(torch.int32).numpy().astype("uint16"),
         gap=gap.float(), **extra)


def _lift_methods(relpath, cls_name, names, ns):
    """The named methods of a reference class, lifted out of their module by AST (the modules import cv2 / mmcv / tensorboard, absent
    here) into a bare class of the same name: the methods' own statements run unchanged, on the stand-ins `ns` gives their globals."""
    import ast
    src = open(os.path.join(ref_import.REF_ROOT, relpath)).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == cls_name][0]
    body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), [n.name for n in body]
    new = ast.ClassDef(name=cls_name, bases=[], keywords=[], body=body, decorator_list=[])
    mod = ast.Module([new], [])
    ast.fix_missing_locations(mod)
    exec(compile(mod, "ref:" + relpath, "exec"), ns)
    return ns[cls_name]


class _NP:
    """numpy as the reference's numpy 1.x: `np.float` still exists"""
    float = float

    def __getattr__(self, k):
        return getattr(np, k)


def gen_jhmdb_pck():
    """tests/golden/jhmdb_pck.npz: the genuine jhmdb_dataset_rgb.pck_evaluate / compute_pck (jhmdb_dataset.py:144-256), lifted by AST,
    on three synthetic videos: ground-truth joints from .mat files written here (`pos_img`, 1-based, as the dataset ships them),
    predictions = ground truth + noise with some joints marked invisible (x <= 0) and one clip longer than its annotation.
    Stand-ins (no arithmetic): mmcv.imread / imwrite / ProgressBar, cv2.line, terminal_is_available."""
    import tempfile
    import types
    import scipy.io as sio
    mm = types.SimpleNamespace(imread=lambda p_: np.zeros((8, 8, 3), np.uint8), imwrite=lambda *a, **k: True, ProgressBar=lambda n: None)
    ns = {"np": _NP(), "sio": sio, "os": os, "osp": os.path, "mmcv": mm, "cv2": types.SimpleNamespace(line=lambda *a, **k: None),
          "terminal_is_available": lambda: False}
    Cls = _lift_methods("mmpt/datasets/jhmdb_dataset.py", "jhmdb_dataset_rgb", ["compute_pck", "pck_evaluate", "vis_pose"], ns)
    Cls.NUM_KEYPOINTS = 15
    Cls.PALETTE = [[0, 0, 0]] * 16
    rng = np.random.default_rng(15)
    with tempfile.TemporaryDirectory() as tmp:
        ds = Cls.__new__(Cls)
        ds.video_dir, ds.filename_tmpl, ds.samples = tmp, "{:05}.png", []
        gts, preds = [], []
        for v, (T_gt, T_pred) in enumerate(((9, 9), (14, 17), (6, 6))):
            gt = rng.random((2, 15, T_gt)) * 180 + 15
            if v == 2:
                gt[:, :, :] = gt[:, :, :1] + rng.normal(0, 2, gt.shape)          # a nearly still pose: a small box, a large normalised error
            sio.savemat(os.path.join(tmp, f"v{v}.mat"), {"pos_img": gt + 1.0})
            pred = np.zeros((2, 15, T_pred))
            pred[:, :, :T_gt] = gt + rng.normal(0, 30, gt.shape)
            pred[:, :, T_gt:] = 50.0
            hide = rng.random((15, T_pred)) < 0.15
            pred[:, hide] = -1.0                                                   # img2coord's mark of an empty label map (:170)
            ds.samples.append(dict(anno_path=os.path.join(tmp, f"v{v}.mat"), num_frames=T_pred, video_path=os.path.join(tmp, f"v{v}"),
                                   frames_path=[os.path.join(tmp, f"v{v}", f"{i:05}.png") for i in range(T_pred)]))
            gts.append(gt)
            preds.append(pred)
        out = os.path.join(tmp, "out")
        os.makedirs(out)
        res = ds.pck_evaluate([p_.copy() for p_ in preds], out)
    arrs = {}
    for v in range(3):
        arrs[f"gt{v}"], arrs[f"pred{v}"] = gts[v], preds[v]
    save("jhmdb_pck", n_videos=3, **arrs, **{k.replace("@", "_at_"): np.float64(v_) for k, v_ in res.items()})


def gen_badja_pck():
    """tests/golden/badja_pck.npz: the genuine BadjaDataset.pck_evaluate (badja_dataset.py:438-583), lifted by AST, on two synthetic
    animals (silhouettes, (y, x) joints, visibility flags, unlabelled frames).  Stand-ins (no arithmetic): SummaryWriter /
    pips_vis.Summ_writer (never used: vis_traj is off), mmcv.imresize at the identity size, torch's .cuda(), get_video."""
    import math
    import tempfile
    import types
    mm = types.SimpleNamespace(imresize=lambda x, size, interpolation=None: (x if (x.shape[1], x.shape[0]) == tuple(size) else 1 / 0),
                               ProgressBar=lambda n: None)
    ns = {"np": _NP(), "os": os, "osp": os.path, "mmcv": mm, "math": math, "torch": torch, "terminal_is_available": lambda: False,
          "SummaryWriter": lambda *a, **k: None, "pips_vis": types.SimpleNamespace(Summ_writer=lambda **k: None)}
    Cls = _lift_methods("mmpt/datasets/badja_dataset.py", "BadjaDataset", ["pck_evaluate"], ns)
    rng = np.random.default_rng(20)
    H, W, J = 48, 64, 20
    videos = []
    for v, T in enumerate((7, 5)):
        frames = [rng.integers(0, 255, (H, W, 3)).astype(np.uint8) for _ in range(T)]
        segs, joints, vis = [], [], []
        for t in range(T):
            sg = np.zeros((H, W), np.uint8)
            sg[8 + v:30 + t, 10:40 + 2 * t] = 255
            segs.append(sg)
            unlabelled = (t == 3 and v == 0)
            joints.append(None if unlabelled else np.stack([rng.random(J) * (H - 1), rng.random(J) * (W - 1)], 1))
            vis.append(None if unlabelled else (rng.random(J) < 0.7).astype(np.int64))
        if joints[0] is None:
            raise AssertionError
        videos.append((frames, segs, joints, vis))
    preds = []
    for frames, segs, joints, vis in videos:
        T = len(frames)
        p_ = np.zeros((2, J, T))
        for t in range(T):
            j = joints[t] if joints[t] is not None else np.zeros((J, 2))
            p_[0, :, t] = j[:, 1] + rng.normal(0, 4, J)
            p_[1, :, t] = j[:, 0] + rng.normal(0, 4, J)
        preds.append(p_)
    with tempfile.TemporaryDirectory() as tmp, ref_import.cuda_as_cpu():
        ds = Cls.__new__(Cls)
        ds.size, ds.length, ds.vis_traj = (H, W), -1, False
        ds.get_video = lambda i: ([f.copy() for f in videos[i][0]], [s_.copy() for s_ in videos[i][1]],
                                  [None if j is None else j.copy() for j in videos[i][2]],
                                  [None if q is None else q.copy() for q in videos[i][3]], f"v{i}")
        Cls.__len__ = lambda self: 2
        res = ds.pck_evaluate([p_.copy() for p_ in preds], tmp)
        txt = open(os.path.join(tmp, "result.txt")).read()
    avg = float(txt.split("PCK@0.1 AVG:")[1].strip())                # (the per-video mean of PCK@0.2, written under this label: :552-557, :578)
    arrs = {}
    for v, (frames, segs, joints, vis) in enumerate(videos):
        T = len(frames)
        arrs[f"pred{v}"] = preds[v]
        arrs[f"segs{v}"] = np.stack(segs, 0)
        arrs[f"labelled{v}"] = np.array([j is not None for j in joints])
        arrs[f"joints{v}"] = np.stack([j if j is not None else np.zeros((J, 2)) for j in joints], 0)
        arrs[f"visible{v}"] = np.stack([q if q is not None else np.zeros(J, np.int64) for q in vis], 0)
    save("badja_pck", n_videos=2, per_video_mean_pck02=np.float64(avg), **arrs,
         **{k.replace("@", "_at_"): np.float64(v_) for k, v_ in res.items()})


def gen_tracker_all():
    gen_tracker_all_queries(False)


def gen_tracker_trained():
    gen_tracker_all_queries(True)


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate single fixtures: python gen_golden.py gen_hr_tracker ...
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        main()
        gen_tapvid_metrics()
        gen_hr_tracker()
        gen_extra_modes()
        gen_tv_keymap()
        gen_tracker_cfg0()
        gen_dense_api()
        gen_tracker_8f()
        gen_jhmdb_pck()
        gen_badja_pck()
        gen_tracker_all()
        gen_tracker_trained()
