"""HIP streams PyTorch does not hand out: a stream whose kernels may only run on a subset of the chip's compute units.

Why (round 6): the launches behind the pair top-k -- the slot merge, the re-scoring kernels, seven sweeps, the read-out -- are short
chains of small workgroups.  On the same footing as everything else they settle on whatever SIMD is free, all over the chip, and a
compute unit that holds one of them cannot take a workgroup of the encoder's one-wave-per-SIMD kernels (those need the whole register
file and 150 KB of LDS of a CU).  Confined to a few CUs they fragment nothing; they are latency-bound and not on the step's critical path.
`hipExtStreamCreateWithCUMask` is a user-level HIP call (no privileges); the handle is wrapped as a `torch.cuda.ExternalStream`, so
`torch.cuda.stream(...)`, `wait_stream`, events and `fgvc_amd.ops` (which launch on torch's current stream) work unchanged."""
import ctypes
import os
from typing import Optional

import torch

_hip = None
_keep = []          # the streams live as long as the process (ExternalStream does not own its handle)


def _runtime():
    """the HIP runtime this process already has (the one torch loaded): never a second copy"""
    global _hip
    if _hip is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("fgvc_amd.streams: the HIP runtime is not loaded (import torch with ROCm first)")
        _hip = ctypes.CDLL(path)
        _hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    return _hip


def cu_mask(n_cus: int, total: int, spread: bool = True):
    """`n_cus` of `total` compute units as the words hipExtStreamCreateWithCUMask takes.  `spread`: every 32-bit word (32 consecutive CU
    ids) gives the same share, so that whatever the id order is -- XCD-major or interleaved -- every part of the chip gives a few"""
    if not 0 < n_cus <= total:
        raise ValueError(f"cu_mask: {n_cus} of {total}")
    words = (total + 31) // 32
    out = [0] * words
    if not spread:
        for i in range(n_cus):
            out[i // 32] |= 1 << (i % 32)
        return out
    per, extra = divmod(n_cus, words)
    for w in range(words):
        k = per + (1 if w < extra else 0)
        lim = min(32, total - 32 * w)
        step = max(1, lim // max(k, 1))
        for j in range(k):
            out[w] |= 1 << min(j * step, lim - 1)
    return out


def cu_masked_stream(device, n_cus: int, spread: bool = True) -> "torch.cuda.Stream":
    """a stream of `device` whose kernels run on `n_cus` compute units only"""
    dev = torch.device(device)
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    mask = cu_mask(int(n_cus), total, spread)
    arr = (ctypes.c_uint32 * len(mask))(*mask)
    handle = ctypes.c_void_p()
    with torch.cuda.device(dev):
        torch.cuda.current_stream(dev)                      # (the context exists)
        rc = _runtime().hipExtStreamCreateWithCUMask(ctypes.byref(handle), len(mask), arr)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with {rc}")
    s = torch.cuda.ExternalStream(handle.value, device=dev)
    _keep.append((s, handle))
    return s


def small_stream_from_env(device) -> Optional["torch.cuda.Stream"]:
    """FGVC_SMALL_STREAM_CUS=<n>: the stream the backend's small launches go to (unset or 0: none)"""
    n = int(os.environ.get("FGVC_SMALL_STREAM_CUS", "0") or 0)
    return cu_masked_stream(device, n) if n > 0 else None
