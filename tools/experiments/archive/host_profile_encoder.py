"""Host-side cost of one encoder call (python + launches, no synchronisation) at a small clip, with cProfile's top entries: the
2-frame 256 x 256 case is host-bound."""
import os, sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg1_256_2f"]
model = bench.build_tracker(wl, dev)
rgbs = torch.randn(wl["frames"], 3, wl["h"], wl["w"], device=dev)
for _ in range(5):
    model.get_feats_hwc(rgbs, split=True)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K):
    model.get_feats_hwc(rgbs, split=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"encoder: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/call, with the GPU {1e3 * (t2 - t0) / K:.3f} ms/call")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    model.get_feats_hwc(rgbs, split=True)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
