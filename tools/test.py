#!/usr/bin/env python3
"""Evaluation entry point with the steps of the reference's tools/test.py:71-198, on fgvc_amd.

    python tools/test.py CONFIG --task davis [--checkpoint CKPT] [--videos 4 --frames 8 --size 256 256]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/test.py CONFIG --launcher pytorch

CONFIG may be the reference's own configs/eval/res18_d1_eval.py.  The TAP-Vid / JHMDB files are not available
offline, so the default dataset is `SyntheticTapVid` (same sample format); `--data-root DIR_OR_PKL` reads TAP-Vid
pickles (`fgvc_amd.datasets.TapVidPickles`: a directory of per-video pickles as the reference globs them, or the
published tapvid_davis.pkl).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fgvc_amd.mmpt_api as api  # noqa: E402
from fgvc_amd import apis, metrics  # noqa: E402
from fgvc_amd.datasets import BadjaPoses, JhmdbPoses, StridedLoader, SyntheticTapVid, TapVidPickles, badja_evaluate, jhmdb_evaluate  # noqa: E402

DEFAULT_CFG = dict(
    model=dict(type="VanillaTracker",
               backbone=dict(type="ResNet", depth=18, strides=(1, 1, 1, 4), out_indices=(2,), pool_type="none")),
    test_cfg_davis=dict(precede_frames=5, topk=10, temperature=0.07, strides=(1, 1, 1, 4), out_indices=(2,),
                        neighbor_range=30, step=512, with_first=True, with_first_neighbor=True),
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default=None)
    ap.add_argument("--task", default="davis")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    ap.add_argument("--videos", type=int, default=4)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, nargs=2, default=(256, 256))
    ap.add_argument("--points", type=int, default=8)
    ap.add_argument("--query-mode", default="first")
    ap.add_argument("--data-root", default=None, help="TAP-Vid pickles (directory of *.pkl or one .pkl); default: synthetic clips")
    ap.add_argument("--out", default=None)
    ap.add_argument("--out-dir", default=None, help="write summaries<task>.json / results_df<task>.csv / results_list<task>.pkl there "
                                                    "(the files of the reference's save_results, tapvid.py:316-350)")
    a = ap.parse_args()

    cfg = api.Config.fromfile(a.config) if a.config else api.Config(DEFAULT_CFG)           # tools/test.py:75
    distributed = a.launcher != "none"
    rank, world = 0, 1
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:                                                                          # :106-110
        dist.init_process_group("nccl", device_id=dev)
        rank, world = dist.get_rank(), dist.get_world_size()

    if a.task in ("jhmdb", "badja"):
        if not a.data_root:
            raise SystemExit("--task jhmdb needs --data-root (JHMDB frames + joint_positions + val_list.txt); --task badja the BADJA root "
                             "(joint_annotations/*.json, JPEGImages/, Annotations/)")
        dataset = None
    elif a.data_root:
        dataset = TapVidPickles(a.data_root, a.query_mode, tuple(a.size), device=dev)                   # :121-122
    else:
        dataset = SyntheticTapVid(a.videos, a.frames, tuple(a.size), a.points, a.query_mode, device=dev)
    loader = StridedLoader(dataset, rank, world) if dataset is not None else None          # :124-134
    key = "test_cfg_" + a.task                                                               # :135
    if key not in cfg and a.task in ("jhmdb", "badja") and "test_cfg_davis" in cfg:
        key = "test_cfg_davis"         # the pose task of DEFAULT_CFG (and of configs without a test_cfg_jhmdb) tracks with the TAP-Vid settings
    if key not in cfg:
        raise SystemExit(f"the config has no '{key}' (tasks it defines: {sorted(k[9:] for k in cfg if k.startswith('test_cfg_'))})")
    test_cfg = cfg[key]
    model_cfg = dict(type=cfg.get("eval_arc", "VanillaTracker"), backbone=dict(cfg.model.backbone))   # :139
    for k in ("out_indices", "strides", "dilations"):                                        # :141-145
        if k in test_cfg:
            model_cfg["backbone"][k] = test_cfg[k]
    model = api.build_model(model_cfg, train_cfg=None, test_cfg=test_cfg)                    # :152
    model.init_weights()                                                                     # :153
    if a.checkpoint:
        api.load_checkpoint(model, a.checkpoint)                                             # :158-159
    model = model.to(dev).eval()

    if a.task == "badja":      # animal pose tracking: the 20 annotated SMAL joints of frame 0 are the query points (datasets.BadjaPoses)
        if rank == 0:          # (one process scores the set, as for JHMDB below; badja_dataset.py:451-571)
            pck = badja_evaluate(model, BadjaPoses(a.data_root, size=(320, 512), device=dev))
            print(json.dumps({k: round(v, 2) for k, v in pck.items()}))
        outputs = None
    elif a.task == "jhmdb":    # pose tracking: the 15 joints of frame 0 are the query points (fgvc_amd.datasets.JhmdbPoses)
        # PCK is a mean over ALL videos' joints (jhmdb_dataset.py:174-256), so the set is scored by one process: rank 0 runs it, the
        # other ranks of a `--launcher pytorch` job wait at the common teardown below
        if rank == 0:
            pck = jhmdb_evaluate(model, JhmdbPoses(a.data_root, split="val", input_size=(320, 320), device=dev))
            print(json.dumps({k: round(v, 2) for k, v in pck.items()}))
        outputs = None
    else:
        outputs = apis.multi_gpu_test(model, loader) if distributed else apis.single_gpu_test(model, loader)   # :160-190
    if rank == 0 and outputs is not None:
        summary = metrics.tapvid_evaluate(outputs, a.query_mode)                             # :192-198
        keep = ("average_pts_within_thresh", "average_jaccard", "occlusion_accuracy", "ade_visible")
        print(json.dumps({k: round(summary[k], 3) for k in keep}))
        if a.out:
            with open(a.out, "w") as f:
                json.dump(summary, f, indent=1)
        if a.out_dir:
            summaries, results_list = metrics.tapvid_summaries(outputs, a.query_mode, tuple(a.size), tuple(a.size))
            print(json.dumps(metrics.save_results(summaries, results_list, a.out_dir, {"dataset": a.task, "query_mode": a.query_mode})))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
