"""Host time per bench step (how far the CPU runs ahead of the GPU): enqueue K steps without synchronising, then synchronise."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fgvc_amd import _lib, engine
dev = torch.device("cuda", 0)
_lib.load()
wl = bench.WORKLOADS["cfg2_480p_8f"]
model = bench.build_tracker(wl, dev)
cfg = model.engine_config()
T, h, w, P = wl["frames"], wl["h"], wl["w"], wl["points"]
rgbs = torch.randn(1, T, 3, h, w, device=dev)
pts = torch.rand(P, 2, device=dev) * torch.tensor([w - 1.0, h - 1.0], device=dev)
plan = engine.plan_clip(T, [0], cfg)
tail = torch.cuda.Stream(dev)


def step():
    feats, Hf, Wf = model.get_feats_hwc(rgbs[0], split=True)
    tk = engine.run_affinity(feats, Hf, Wf, plan, cfg)
    return engine.run_propagation_async(tk, 0, pts, Hf, Wf, h, w, cfg, tail)


for _ in range(5):
    step()
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, total {1e3 * (t2 - t0) / K:.2f} ms/step")
