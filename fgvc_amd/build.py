"""Build libfgvc_hip.so (the product) in-tree with hipcc for gfx950.

    python -m fgvc_amd.build            # incremental
    python -m fgvc_amd.build --force

The shared library is a plain C-ABI object (no torch, no pybind): see include/fgvc_hip.h.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "lib", "libfgvc_hip.so")
LIB_ABLATIONS = os.path.join(HERE, "lib", "libfgvc_hip_ablations.so")      # the same objects, fgvc_set_option accepts the results-wrong profiling switches
SOURCES = ["capi.hip", "pair_topk.hip", "pair_topk_v5.hip", "conv_split.hip", "conv_s2.hip", "conv64.hip", "stem7.hip", "post.hip", "corr_volume.hip", "corr_volume_f8.hip", "corr_volume_f6.hip", "local.hip", "dense_attend.hip", "refine.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "pair_common.hpp"), os.path.join(CSRC, "sortnet.hpp"), os.path.join(ROOT, "include", "fgvc_hip.h"),
           os.path.join(CSRC, "pair_topk_v7.hpp"), os.path.join(CSRC, "pair_v7.inc"), os.path.join(CSRC, "pair_topk_v8.hpp"), os.path.join(CSRC, "pair_v8.inc"), os.path.join(CSRC, "pair_v5_chain.inc")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# pair_topk_v5.hip unrolls a 48-MFMA chain with one slice of a sorting network behind every MFMA: beyond clang's default budget for
# `#pragma unroll` (the loop would stay rolled and the register-resident operands would go to scratch)
EXTRA_FLAGS = {"pair_topk_v5.hip": ["-mllvm", "-pragma-unroll-threshold=1048576", "-Werror=pass-failed"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[fgvc build]", " ".join(cmd[-3:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    # the experiment library: capi.hip once more with -DFGVC_ABLATIONS, every other object shared (identical kernels: A/B timings carry over)
    capi_ab = os.path.join(OBJ, "capi_ablations.o")
    if force or _stale(capi_ab, [os.path.join(CSRC, "capi.hip")] + HEADERS):
        run([hipcc, *FLAGS, "-DFGVC_ABLATIONS", "-c", os.path.join(CSRC, "capi.hip"), "-o", capi_ab])
    objs_ab = [capi_ab if o.endswith(os.sep + "capi.o") else o for o in objs]
    if force or jobs or _stale(LIB_ABLATIONS, objs_ab):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_ABLATIONS, *objs_ab])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
