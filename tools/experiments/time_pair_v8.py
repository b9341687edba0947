"""s_memtime probe of one workgroup of pair_topk_kernel_v8 (pair_f16_debug = 256; workgroup (40, 0): a tile of the longest run): per wave
(waves 0-3 = query blocks (br, 0)) the cycles of its loop, of issuing + posting key blocks, of waiting for a block to be complete, of
its chains and of keys + selection."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgvc_amd import engine, ops, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W, C, T = 120, 214, 256, 8
feats = ops.normalize_to_hwc(torch.randn(T, C, H, W, device=dev))
sp6 = ops.split_f16f6x(feats)
cfg = engine.TrackerConfig()
plan = engine.plan_clip(T, [0], cfg)
pairs = ops.make_pairs(plan.pairs, dev)
buf = (ctypes.c_int64 * 32)()
V8 = 4194304
variants = tuple((V8 + d, n) for d, n in ((256, "default"), (256 + 1, "no bytes moved"), (256 + 2, "no chain"), (256 + 1048576, "no selection"), (256 + 1 + 2 + 1048576, "protocol only"), (256 + 1024, "row-major list")))
for dbg, name in variants:
    ops.set_option("pair_f16_debug", dbg)
    for _ in range(3):
        ops.pair_topk_split(sp6, sp6, pairs, H, W, H, W, cfg.mask, 10, validate=False, all_masked=True, fmt="f16f6x")
    torch.cuda.synchronize()
    _lib.call("fgvc_pair_topk_f16x3_probe", ctypes.cast(buf, ctypes.c_void_p))
    v = list(buf)
    print(name)
    for q in range(4):
        tot, fill, chain, sel, iss, nt, ns, gc = v[8 * q:8 * q + 8]
        nit = max(ns * gc, 1)
        print(f"  wave {q}: loop {tot:8d} cycles, {gc} pairs x {ns} blocks = {tot / nit:6.0f} per step; issue + post {iss:8d} ({iss / nit:5.0f} per step); waiting for blocks {fill:8d} "
              f"({fill / max(nt, 1):5.0f} per tile); chains {chain:8d} = {chain / max(nt, 1):5.0f} per tile x {nt}; keys + selection {sel:8d} = {sel / max(nt, 1):5.0f} per tile")
ops.set_option("pair_f16_debug", 0)
print("timed out:", ops.pair_f16x3_timed_out())
