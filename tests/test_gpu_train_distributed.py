"""Two ranks of the REAL fine-tuning step (SURVEY.md section 5 last row / 8e; reference utils/utils.py:285-286 wraps the model in
DataParallel, utils/training.py:184-233 is the loop): `training.train_epoch` on the HIP training path in two FRESH processes
(tools/dist_train_check.py, started with torch.distributed.run), gradients averaged by the flat all-reduce of
`training.allreduce_gradients`.
  * gradients: with per-sample-independent layers (dropout 0, BatchNorm on running statistics) the mean of the two ranks' gradients
    must equal the gradient of the concatenated batch on one rank -- to fp32 rounding: the two runs associate the sums over edges and
    samples differently.  Stated tolerance: 2e-5 of the largest gradient component per step (observed ~1e-6).
  * a NaN score on ONE rank: every rank skips that step (parameters untouched) and takes the next one.
  * shipped configuration (train-mode BatchNorm, dropout, Adam, EMA): the ranks end bitwise identical, BatchNorm statistics included.
gloo with both ranks on cuda:0 runs on the 1-GPU box; the nccl (= RCCL) variants need two GPUs and skip otherwise."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "dist_train_check.py")
TWO_GPUS = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, mode, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, TOOL, "--mode", mode, "--out", out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), TOOL, "--backend", backend, "--mode", mode, "--out", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return np.load(out)


@pytest.fixture(scope="module")
def one_rank(tmp_path_factory):
    d = tmp_path_factory.mktemp("train_dist")
    return {m: _run(1, None, m, str(d / f"w1_{m}.npz")) for m in ("grads", "nan")}


def _check_grads(w1, w2):
    assert int(w2["world"]) == 2 and w1["snaps"].shape == w2["snaps"].shape and w1["snaps"].shape[0] == 3
    lr = float(w1["lr"])
    for s in (1, 2):
        g1 = (w1["snaps"][s - 1] - w1["snaps"][s]) / lr
        g2 = (w2["snaps"][s - 1] - w2["snaps"][s]) / lr
        assert np.isfinite(g1).all() and np.abs(g1).max() > 1e-3                       # a real step
        # parameters are O(0.1): (p - lr g) rounds at ~1e-8, i.e. 2e-7 in g; the rest is the association of the sums
        assert np.abs(g1 - g2).max() <= 2e-5 * np.abs(g1).max() + 4e-7, (s, np.abs(g1 - g2).max(), np.abs(g1).max())
    assert np.array_equal(w2["params_by_rank"][0], w2["params_by_rank"][1])            # both ranks applied the same update
    assert np.allclose(w1["losses"], w2["losses"], rtol=0, atol=0.5)                   # rank 0's half-batch loss: same ballpark only


def _check_nan(w1, w2):
    # world 2: step 1 skipped on BOTH ranks although only rank 1 saw the NaN, step 2 taken; world 1 reference = step 2 alone
    assert np.array_equal(w2["snaps"][0], w2["snaps"][1])
    assert not np.array_equal(w2["snaps"][1], w2["snaps"][2])
    lr = float(w1["lr"])
    g1, g2 = (w1["snaps"][0] - w1["snaps"][1]) / lr, (w2["snaps"][1] - w2["snaps"][2]) / lr
    assert np.abs(g1 - g2).max() <= 2e-5 * np.abs(g1).max() + 4e-7
    assert np.array_equal(w2["params_by_rank"][0], w2["params_by_rank"][1])


def _check_full(w2):
    assert np.isfinite(w2["snaps"]).all() and not np.array_equal(w2["snaps"][0], w2["snaps"][2])
    assert np.array_equal(w2["params_by_rank"][0], w2["params_by_rank"][1])
    assert np.array_equal(w2["bn_by_rank"][0], w2["bn_by_rank"][1])


def test_two_ranks_gloo_gradients_equal_the_concatenated_batch(one_rank, tmp_path):
    _check_grads(one_rank["grads"], _run(2, "gloo", "grads", str(tmp_path / "w2.npz")))


def test_two_ranks_gloo_skip_a_nan_step_together(one_rank, tmp_path):
    _check_nan(one_rank["nan"], _run(2, "gloo", "nan", str(tmp_path / "w2.npz")))


def test_two_ranks_gloo_shipped_configuration_stays_in_sync(tmp_path):
    _check_full(_run(2, "gloo", "full", str(tmp_path / "w2.npz")))


@TWO_GPUS
def test_two_ranks_rccl_gradients_equal_the_concatenated_batch(one_rank, tmp_path):
    _check_grads(one_rank["grads"], _run(2, "nccl", "grads", str(tmp_path / "w2.npz")))


@TWO_GPUS
def test_two_ranks_rccl_nan_and_shipped_configuration(one_rank, tmp_path):
    _check_nan(one_rank["nan"], _run(2, "nccl", "nan", str(tmp_path / "w2n.npz")))
    _check_full(_run(2, "nccl", "full", str(tmp_path / "w2f.npz")))
