import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's CPU work runs on torch's intra-op pool: a GPU box shows every host core but grants a share of 16 -- a pool of
    # 100+ threads on that share made one oracle comparison take minutes instead of seconds
    try:
        import torch
        torch.set_num_threads(max(1, min(torch.get_num_threads(), 16)))
    except Exception:
        pass


def pytest_sessionstart(session):
    """The two-rank run of the product backend on one GPU is a tree of child processes.  It is started HERE, before any test has
    touched the GPU from this process (a process that has initialised the GPU must not start other programs on the GPU boxes;
    counting devices does not initialise it), and its verdict is read by tests/test_gpu_api.py."""
    import subprocess
    TWO_RANKS = session.config._fgvc_two_ranks = {}
    expr = session.config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    tool = os.path.join(ROOT, "tools", "two_ranks_one_gpu.py")
    try:
        r = subprocess.run([sys.executable, tool, "--tail-stream"], capture_output=True, text=True, timeout=600)
        TWO_RANKS.update(rc=r.returncode, out=r.stdout[-6000:], err=r.stderr[-3000:])
    except Exception as e:   # reported by the test
        TWO_RANKS.update(rc=-1, out="", err=repr(e))
    try:   # BASELINE configs[3]: a 64-frame TAP-Vid-DAVIS-shape video (256 x 256 -> 128 x 128 x 256 features, P = 32) over the two ranks
        r = subprocess.run([sys.executable, tool, "--frames", "64", "--size", "256", "256", "--strides", "1", "1", "1", "4", "--precede", "5",
                            "--neighbor-range", "30", "--points", "32", "--halos", "exchange", "--tail-stream"],
                           capture_output=True, text=True, timeout=900)
        TWO_RANKS["cfg4"] = dict(rc=r.returncode, out=r.stdout[-6000:], err=r.stderr[-3000:])
    except Exception as e:
        TWO_RANKS["cfg4"] = dict(rc=-1, out="", err=repr(e))
    try:   # `python bench.py --gpus 2` on its own, with a clean environment: the bench starts its two ranks itself
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                                 "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "5", "--warmup", "2",
                            "--repeats", "0", "--no-corr-volume", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
        TWO_RANKS["bench2"] = dict(rc=r.returncode, out=r.stdout[-12000:], err=r.stderr[-3000:])
    except Exception as e:
        TWO_RANKS["bench2"] = dict(rc=-1, out="", err=repr(e))
    try:   # four ranks (the two middle ones both send and receive a halo in one message batch), 26 frames, the full five-frame halo
        r = subprocess.run([sys.executable, tool, "--world", "4", "--frames", "26", "--precede", "5", "--halos", "exchange", "--tail-stream"],
                           capture_output=True, text=True, timeout=900)
        TWO_RANKS["w4"] = dict(rc=r.returncode, out=r.stdout[-8000:], err=r.stderr[-3000:])
    except Exception as e:
        TWO_RANKS["w4"] = dict(rc=-1, out="", err=repr(e))


@pytest.fixture(scope="session")
def two_ranks(request):
    """Verdict of tools/two_ranks_one_gpu.py (dict with rc / out / err; empty when the session is not a GPU run)."""
    return getattr(request.config, "_fgvc_two_ranks", {})


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))

    return load
