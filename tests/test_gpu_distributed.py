"""The N > 1 path on the REAL engine (reference inference.py:537-547 ranks the samples of a complex by confidence):
`distributed.sampling_distributed` run by two FRESH rank processes (started with torch.distributed.run from here; a process that
has touched the GPU is never re-executed) must give the ranked poses, confidences and sample order of the one-rank run.
Agreement is to fp32 rounding, not bitwise: a rank's batch holds a different subset of the samples, so a pose's edges fall at other
positions of the 32-edge reduction tiles and its per-node sums are associated differently (results ARE bitwise repeatable for a fixed
split, tests/test_gpu_parity.py).  Stated tolerance: poses 1e-4 A, confidences 1e-5, the same ranking wherever two confidences are
further apart than that.
  * gloo, both ranks on cuda:0 -- runs on the 1-GPU box: real engines in two processes, CPU gather;
  * nccl (= RCCL), one GPU per rank -- runs when >= 2 GPUs are visible, skipped otherwise."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "dist_sampling_check.py")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, TOOL, "--out", out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), TOOL, "--backend", backend, "--out", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


@pytest.fixture(scope="module")
def reference_run(tmp_path_factory):
    return _run(1, None, str(tmp_path_factory.mktemp("dist") / "w1.npz"))


def _same(a, b):
    assert int(b["world"]) == 2
    n = len(a["index"])
    assert sorted(a["index"].tolist()) == sorted(b["index"].tolist()) == list(range(n))
    ia, ib = np.argsort(a["index"]), np.argsort(b["index"])              # back to sample order
    ca, cb = a["confidence"][ia], b["confidence"][ib]
    assert np.abs(ca - cb).max() <= 1e-5
    assert np.sqrt(((a["pos"][ia] - b["pos"][ib]) ** 2).sum(-1).mean(-1)).max() <= 1e-4
    for r in range(n - 1):                                               # ranked by descending confidence, ties by sample index
        assert b["confidence"][r] >= b["confidence"][r + 1]
        if a["index"][r] != b["index"][r]:
            assert abs(ca[a["index"][r]] - ca[b["index"][r]]) <= 2e-5      # only near-ties may swap places


def test_two_ranks_one_gpu_gloo_equals_one_rank(reference_run, tmp_path):
    _same(reference_run, _run(2, "gloo", str(tmp_path / "w2_gloo.npz")))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_two_gpus_rccl_equals_one_rank(reference_run, tmp_path):
    _same(reference_run, _run(2, "nccl", str(tmp_path / "w2_nccl.npz")))


def test_rccl_initialises_and_runs_the_gathers_on_this_gpu(tmp_path):
    """No multi-GPU box was ever available to this build, so RCCL itself had never executed under this code.  What CAN
    run on one GPU: a world of ONE rank on the "nccl" backend (= RCCL) in a fresh process -- communicator creation with the pool's
    HSA_ENABLE_IPC_MODE_LEGACY=0 setting, an all-reduce and the two gathers of distributed.py on DEVICE tensors through RCCL kernels
    (gather_ranked's early return for world 1 is bypassed by calling the collective the way it does)."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%r, RANK='0', WORLD_SIZE='1')\n"
        "os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "dev = torch.device('cuda', 0)\n"
        "t = torch.arange(8, dtype=torch.float64, device=dev)\n"
        "dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "pos = torch.randn(5, 7, 3, device=dev); conf = torch.tensor([0.1, 0.9, 0.5, 0.9, -1.0], device=dev)\n"
        "payload = torch.cat([conf[:, None].double(), pos.reshape(5, -1).double()], 1)\n"
        "out = [torch.empty_like(payload)]\n"
        "dist.gather(payload, out, dst=0)\n"
        "dist.barrier()\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(out[0], payload) and t.tolist() == list(range(8))\n"
        "from confidence_bootstrapping_amd.distributed import gather_ranked\n"
        "p, c, i = gather_ranked(pos, conf, 1, 0, 0, ids=torch.arange(5))\n"
        "assert i.tolist() == [1, 3, 2, 0, 4]\n"
        "print('rccl ok', dist.get_backend())\n"
        "dist.destroy_process_group()\n") % (ROOT, str(_port()))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "rccl ok nccl" in r.stdout
