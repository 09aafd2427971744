"""Host-side configuration of the GPU paths: how many intra-op CPU threads torch may use while a step loop is being driven.

The GPU paths do a little CPU tensor work per step (collation, noise drawing, table look-ups): small ops, for which torch's intra-op pool
is useless -- and harmful in a container: torch sizes the pool to the machine's hardware threads (256 on the MI355X boxes) while the
cgroup grants far fewer CPUs (16 here, `cpu.max`); after every tiny parallel region the OpenMP workers spin for their block time, the
process burns its CPU quota and the kernel throttles ALL its threads until the next 100 ms period -- including the thread that feeds
the GPU.  Measured on the fine-tuning step (batch 8, round 4): 32-36 ms per step with 12-50 ms GPU-idle holes every 100 ms, 21.4 ms with one
intra-op thread; the stalls were misread as "host-bound by launch count" in rounds 2-3.

`glue_threads()` is a context manager / decorator used by `sampling()`, `train_epoch()`, `train_step()`, `run_complex_set()` ...: inside it
torch's intra-op thread count is min(current, CBD_HOST_THREADS) (default 1) and restored on exit.  The limit is process-global in torch, so
nested / concurrent users are counted under a lock: the first one in saves the old value, the last one out restores it.  The decorator
applies the limit only when the wrapped call's `device` argument is a GPU (a CPU run with a stand-in forward keeps torch's threads).

`canonical_device()` / `dev_key()`: one spelling per device for every per-device cache key ('cuda' and 'cuda:0' name the same device; a
cache written under one and read under the other silently misses)."""
from __future__ import annotations

import contextlib
import functools
import inspect
import os
import threading

import torch


def canonical_device(device) -> torch.device:
    """torch.device(device) with an explicit index for GPUs ('cuda' -> 'cuda:<current device>')."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return dev


def dev_key(device) -> str:
    return str(canonical_device(device))


def _limit() -> int:
    try:
        return max(1, int(os.environ.get("CBD_HOST_THREADS", "1")))
    except ValueError:
        return 1


_LOCK = threading.Lock()
_STATE = {"users": 0, "saved": None}


@contextlib.contextmanager
def glue_threads(n: int = None):
    n = _limit() if n is None else int(n)
    with _LOCK:
        if _STATE["users"] == 0:
            _STATE["saved"] = torch.get_num_threads()
        _STATE["users"] += 1
        if torch.get_num_threads() > n:
            torch.set_num_threads(n)
    try:
        yield
    finally:
        with _LOCK:
            _STATE["users"] -= 1
            if _STATE["users"] == 0 and _STATE["saved"] is not None:
                if torch.get_num_threads() != _STATE["saved"]:
                    torch.set_num_threads(_STATE["saved"])
                _STATE["saved"] = None


def with_glue_threads(fn):
    """Runs `fn` under glue_threads() when its `device` argument (if it has one) names a GPU."""
    try:
        sig = inspect.signature(fn)
        has_device = "device" in sig.parameters
    except (TypeError, ValueError):
        sig, has_device = None, False

    @functools.wraps(fn)
    def wrapped(*a, **k):
        if has_device:
            try:
                dev = sig.bind_partial(*a, **k).arguments.get("device")
                if dev is not None and torch.device(dev).type != "cuda":
                    return fn(*a, **k)
            except (TypeError, RuntimeError):
                pass
        with glue_threads():
            return fn(*a, **k)
    return wrapped
