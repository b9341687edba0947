"""Test-loop counterparts of mmpt/apis/test.py:13-59 (single_gpu_test) and :62-128 (multi_gpu_test).

The reference collects per-rank results through pickle files (its default, :131-189) or an all_gather of uint8 blobs
(:192-236); here: collect_results_cpu (the same files) and collect_results (torch.distributed.all_gather_object;
gloo on CPU tests, RCCL on the GPUs).
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
def multi_gpu_test(model, data_loader, tmpdir=None, gpu_collect=False, **kw):
    """Each rank runs its share of the videos (the loader is expected to stride the list
    indices[rank::world], mmpt/datasets/samplers/distributed_sampler.py:53); rank 0 gets everything back
    in dataset order.  `gpu_collect` / `tmpdir` as in the reference (mmpt/apis/test.py:62-128): False (its default) collects
    through `part_{rank}.pkl` files in `tmpdir` (collect_results_cpu), True through the process group (collect_results)."""
    model.eval()
    results = [model(test_mode=True, **data) for data in data_loader]
    _check_kernels()
    size = getattr(data_loader, "total", None)
    if gpu_collect:
        return collect_results(results, size)
    return collect_results_cpu(results, size, tmpdir)


def collect_results_cpu(part, size=None, tmpdir=None):
    """The reference's default collection (mmpt/apis/test.py:131-189): every rank pickles its results to `tmpdir/part_{rank}.pkl`
    (a directory rank 0 creates and names to the others when none is given), rank 0 reads the parts back, interleaves them into
    dataset order (rank r ran videos r, r + world, ...), cuts the sampler's padding at `size` and removes the directory."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return part
    import os
    import pickle
    import shutil
    import tempfile
    rank, world = dist.get_rank(), dist.get_world_size()
    if tmpdir is None:
        name = [None]
        if rank == 0:
            os.makedirs("/tmp/dist_test", exist_ok=True)
            name[0] = tempfile.mkdtemp(dir="/tmp/dist_test")
        dist.broadcast_object_list(name, src=0)
        tmpdir = name[0]
    else:
        os.makedirs(tmpdir, exist_ok=True)
    dist.barrier()                                       # the directory exists on every rank's view of the file system
    cpu_part = [tuple(t.cpu() if torch.is_tensor(t) else t for t in r) for r in part]
    with open(os.path.join(tmpdir, f"part_{rank}.pkl"), "wb") as f:
        pickle.dump(cpu_part, f)
    dist.barrier()                                       # every part is written
    if rank != 0:
        return None
    parts = []
    for r in range(world):
        with open(os.path.join(tmpdir, f"part_{r}.pkl"), "rb") as f:
            parts.append(pickle.load(f))
    ordered = []
    for i in range(max(len(g) for g in parts)):
        for g in parts:
            if i < len(g):
                ordered.append(g[i])
    shutil.rmtree(tmpdir)
    return ordered[:size] if size is not None else ordered


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
