"""How much of the step do the side-stream kernels cost?  bench.py with the read-out and / or the propagation sweep replaced by no-ops
(results WRONG): the gain an infinitely fast tail could bring.   python3 tools/experiments/ablate_tail.py [readout|propagate|both|none] <bench args>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
what = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import torch
from fgvc_amd import ops
if what in ("readout", "both"):
    _cache = {}
    def fake_readout(labels, Hf, Wf, h, w, gauss_points=None, sigma=6.0):
        n, P = labels.shape[0], labels.shape[2]
        k = (n, P, labels.device)
        if k not in _cache:
            _cache[k] = torch.zeros((n, P, 2), device=labels.device, dtype=torch.float64)
        return _cache[k]
    ops.softargmax_top5 = fake_readout
if what in ("propagate", "both"):
    real = ops.propagate_topk
    def fake_prop(*a, **k):
        out = k.get("out")
        return out if out is not None else real(*a, **k)
    ops.propagate_topk = fake_prop
import importlib.util
spec = importlib.util.spec_from_file_location("bench_main", os.path.join(ROOT, "bench.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
m.main()
