"""ResNet.use_graph: the encoder call replayed from a HIP graph against the eager path -- same bits, host / total time per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fgvc_amd.mmpt_api.backbones import ResNet
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["cfg1_256_2f", "cfg2_480p_8f"]):
    wl = bench.WORKLOADS[name]
    model = bench.build_tracker(wl, dev)
    rgbs = torch.randn(wl["frames"], 3, wl["h"], wl["w"], device=dev)
    model.test_cfg["batch_step"] = wl["frames"]
    res = {}
    for use in (False, True):
        ResNet.use_graph = use
        for _ in range(4):
            f, Hf, Wf = model.get_feats_hwc(rgbs, split=True)
        torch.cuda.synchronize()
        res[use] = f.clone()
        K = 50
        t0 = time.perf_counter()
        for _ in range(K):
            model.get_feats_hwc(rgbs, split=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name} use_graph={use}: host {1e3 * (t1 - t0) / K:.3f} ms/call, with the GPU {1e3 * (t2 - t0) / K:.3f} ms/call", flush=True)
    ResNet.use_graph = False
    print(f"{name}: identical features: {torch.equal(res[False], res[True])}")
