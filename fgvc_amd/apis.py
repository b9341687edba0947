"""Test-loop counterparts of mmpt/apis/test.py:13-59 (single_gpu_test) and :62-128 (multi_gpu_test).

The reference collects per-rank results through pickle files or an all_gather of uint8 blobs
(:131-236); here results are lists of tensors gathered with torch.distributed.all_gather_object
(gloo on CPU tests, RCCL on the GPUs).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _check_kernels():
    """The pair kernel's LDS protocol waits with bounded spins; a wait that ever gave up leaves wrong lists behind.  Never seen -- but a
    test loop must not return such results silently: checked once per loop (the check synchronises the device)."""
    if torch.cuda.is_available():
        from . import ops
        if ops.pair_f16x3_timed_out():
            raise RuntimeError("fgvc_pair_topk_f16x3: a bounded wait of the kernel's LDS protocol timed out; the results of this run are invalid")


@torch.no_grad()
def single_gpu_test(model, data_loader, **kw):
    model.eval()
    results = []
    for data in data_loader:
        results.append(model(test_mode=True, **data))
    _check_kernels()
    return results


@torch.no_grad()
def multi_gpu_test(model, data_loader, tmpdir=None, gpu_collect=True, **kw):
    """Each rank runs its share of the videos (the loader is expected to stride the list
    indices[rank::world], mmpt/datasets/samplers/distributed_sampler.py:53); rank 0 gets everything back
    in dataset order."""
    model.eval()
    results = [model(test_mode=True, **data) for data in data_loader]
    _check_kernels()
    return collect_results(results, getattr(data_loader, "total", None))


def collect_results(part, size=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return part
    rank, world = dist.get_rank(), dist.get_world_size()
    cpu_part = [tuple(t.cpu() if torch.is_tensor(t) else t for t in r) for r in part]
    gathered = [None] * world
    dist.all_gather_object(gathered, cpu_part)
    if rank != 0:
        return None
    ordered = []
    for i in range(max(len(g) for g in gathered)):        # interleave back: rank r holds items r, r+world, ...
        for g in gathered:
            if i < len(g):
                ordered.append(g[i])
    return ordered[:size] if size is not None else ordered
