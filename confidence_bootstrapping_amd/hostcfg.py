"""Host-side configuration of the GPU paths: how many intra-op CPU threads torch may use while a step loop is being driven.

The GPU paths do a little CPU tensor work per step (collation, noise drawing, table look-ups): small ops, for which torch's intra-op pool
is useless -- and harmful in a container: torch sizes the pool to the machine's hardware threads (256 on the MI355X boxes) while the
cgroup grants far fewer CPUs (16 here, `cpu.max`); after every tiny parallel region the OpenMP workers spin for their block time, the
process burns its CPU quota and the kernel throttles ALL its threads until the next 100 ms period -- including the thread that feeds
the GPU.  Measured on the fine-tuning step (batch 8, round 4): 32-36 ms per step with 12-50 ms GPU-idle holes every 100 ms, 21.4 ms with one
intra-op thread; the stalls were misread as "host-bound by launch count" in rounds 2-3.

`glue_threads()` is a context manager / decorator used by `sampling()`, `train_epoch()`, `train_step()`, `run_complex_set()` ...: inside it
torch's intra-op thread count is min(current, CBD_HOST_THREADS) (default 1) and restored on exit.  The CPU ORACLE is never run under it."""
from __future__ import annotations

import contextlib
import functools
import os

import torch


def _limit() -> int:
    try:
        return max(1, int(os.environ.get("CBD_HOST_THREADS", "1")))
    except ValueError:
        return 1


@contextlib.contextmanager
def glue_threads(n: int = None):
    n = _limit() if n is None else int(n)
    old = torch.get_num_threads()
    changed = old > n
    if changed:
        torch.set_num_threads(n)
    try:
        yield
    finally:
        if changed:
            torch.set_num_threads(old)


def with_glue_threads(fn):
    @functools.wraps(fn)
    def wrapped(*a, **k):
        with glue_threads():
            return fn(*a, **k)
    return wrapped
