"""The clip-sharded path with the PRODUCT backend (HipBackend) at world size 2 on ONE GPU.

RCCL refuses two ranks of a communicator on one device ("Duplicate GPU detected"), and the GPU boxes of this project have one
GPU: so the two ranks talk over gloo, with fgvc_amd.dist staging device tensors through the host.  Everything else is the code
that runs over RCCL on a node: bank assembly with messages landing in its slices, pairs launched in two phases around the halo
wait, the all_gather + sweep on the side stream, the bank travelling in the pair kernel's split format.

Each rank checks its trajectories against the un-sharded tracker on the same GPU; the parent (which never touches the GPU: it
must be able to start children) prints one JSON line and exits non-zero on a mismatch.

  python tools/two_ranks_one_gpu.py [--frames 12] [--tail-stream]
  python tools/two_ranks_one_gpu.py --frames 64 --size 256 256 --strides 1 1 1 4 --precede 5 --neighbor-range 30 --points 32 \
         --halos exchange --tail-stream        # BASELINE configs[3]: a TAP-Vid-DAVIS-shape video (128 x 128 x 256 features) over two ranks
"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, path, a, q):
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
        import fgvc_amd.mmpt_api as api
        from fgvc_amd import dist as fdist, engine
        torch.manual_seed(5)                                     # the same random-init encoder on every rank
        model = api.build_model(dict(type="VanillaTracker",
                                     backbone=dict(type="ResNet", depth=18, strides=tuple(a.strides), out_indices=(2,), pool_type="none")),
                                train_cfg=None,
                                test_cfg=api.ConfigDict(precede_frames=a.precede, topk=10, temperature=0.07, neighbor_range=a.neighbor_range,
                                                        with_first=True, with_first_neighbor=True)).to(dev).eval()
        g = torch.Generator().manual_seed(21)
        h, w = a.size
        rgbs = torch.randn(a.frames, 3, h, w, generator=g)
        qp = torch.tensor([[0, 20.0, 12.0], [0, 70.0, 40.0], [2, 33.0, 50.0], [0, 5.5, 60.25]])
        if a.points > 4:                                         # more points, spread over the frame, a quarter of them queried at frame 2
            extra = torch.rand(a.points - 4, 3, generator=g) * torch.tensor([0.0, w - 1.0, h - 1.0])
            extra[::4, 0] = 2.0
            qp = torch.cat([qp, extra], 0)
        cfg = model.engine_config()
        out = {}
        for halo in a.halos:
            for tail in ((False, True, "pairs") if a.tail_stream else (False,)):
                ts = torch.cuda.Stream(dev) if tail else None
                be = fdist.HipBackend(model, tail_stream=ts, tail_from="pairs" if tail == "pairs" else "sweep")
                timing = fdist.Timing(dev)
                cache = {}
                for _ in range(2):                               # second call: cached schedule, buffers of the first still in use
                    traj_s, order_s = fdist.track_points_sharded(be, rgbs, qp, cfg, device=dev, halo=halo, timing=timing, cache=cache)
                torch.cuda.synchronize()
                feats, Hf, Wf = model.get_feats_hwc(rgbs.to(dev))
                traj, order = engine.track_points(feats, Hf, Wf, h, w, qp, cfg)
                rep = timing.report()
                out[halo + {False: "", True: "+tail", "pairs": "+tail_from_pairs"}[tail]] = dict(
                    max_abs_diff_px=float((traj_s.cpu() - traj.cpu()).abs().max()), order_equal=bool(torch.equal(order_s, order)),
                    finite=bool(torch.isfinite(traj_s).all()), phases=sorted(rep),
                    bank_in_place=int(cache["schedule"].get("bank_in_place", 0)), halo_early=int(cache["schedule"].get("halo_early", 0)))     # second call: the encoder wrote into the local bank
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                       # surface the failure instead of letting the parent time out
        import traceback
        q.put((rank, {"error": repr(e), "trace": traceback.format_exc()[-1500:]}))
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--precede", type=int, default=3)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--tail-stream", action="store_true")
    ap.add_argument("--size", type=int, nargs=2, default=(64, 96))
    ap.add_argument("--strides", type=int, nargs=4, default=(1, 2, 1, 1))
    ap.add_argument("--neighbor-range", type=int, default=12)
    ap.add_argument("--points", type=int, default=4)
    ap.add_argument("--halos", nargs="+", default=["exchange", "recompute"], choices=["exchange", "recompute", "auto"])
    a = ap.parse_args()
    fd, path = tempfile.mkstemp(prefix="fgvc_rdzv_")
    os.close(fd)
    os.unlink(path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, a.world, path, a, q)) for r in range(a.world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(a.world))
    for p in procs:
        p.join(60)
    ok = all("error" not in r and all(v["order_equal"] and v["finite"] and v["max_abs_diff_px"] < 1e-3 for v in r.values())
             for r in res.values()) and all(p.exitcode == 0 for p in procs)
    print(json.dumps({"ok": ok, "world": a.world, "frames": a.frames, "size": list(a.size), "backend": "gloo (host-staged) on one GPU",
                      "ranks": {str(k): v for k, v in sorted(res.items())}}))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
