"""bench.py's multi-rank branch on the 1-GPU box: `python bench.py --gpus 2` starts two fresh rank processes
(torch.distributed.run, before any GPU call); with the TEST-ONLY switch CBD_BENCH_ALLOW_SHARED_GPU=1 both ranks use cuda:0 and talk over
gloo, so the barrier, the MAX all-reduce of the elapsed time, the final gathers (`gather_poses` / one `gather_ranked` per complex) and the
rank-0-only printing of measure() all execute -- everything the driver's 8-GPU SCALE run does except RCCL itself.
Checked: exit status 0, exactly ONE contract line (rank 0), n_gpus 2, ranks_seen 2 (all-reduce of ones), `shared_gpu` flagged, the
whole-job value against the N = 1 value of the same build (two processes time-slicing one GPU: the SUM of their work per second must
stay near the one-process figure), for both splits."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "6", "--warmup", "2", "--headline-only", "--no-cpu-baseline"]


def _bench(extra, shared):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.pop("CBD_BENCH_ALLOW_SHARED_GPU", None)
    if shared:
        env["CBD_BENCH_ALLOW_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + extra, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = []
    for ln in r.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            lines.append(json.loads(ln))
    return r, lines


@pytest.fixture(scope="module")
def one_rank():
    r, lines = _bench([], shared=False)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["ranks_seen"] == 1 and "shared_gpu" not in lines[0]
    return lines[0]


def test_two_ranks_refused_without_the_test_switch():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the run is legitimate")
    r, lines = _bench(["--gpus", "2"], shared=False)
    assert r.returncode != 0 and not lines and "refusing" in r.stderr


@pytest.mark.parametrize("split,floor", [("complexes", 0.75), ("samples", 0.5)])
def test_two_ranks_shared_gpu_gloo(one_rank, split, floor):
    r, lines = _bench(["--gpus", "2", "--split", split], shared=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, f"{len(lines)} contract lines (rank 0 only must print)"
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["ranks_seen"] == 2 and ln.get("shared_gpu") is True
    assert ln["steps"] == 6 and ln["warmup"] == 2 and ln["config"]["split"] == split
    assert ln["scaling"] == ("weak" if split == "complexes" else "strong")
    # complexes: 2 x 6 complexes x 40 poses in the max-over-ranks time; samples: 6 x 40 poses whatever N
    poses = 40 * 6 * (2 if split == "complexes" else 1)
    assert abs(ln["value"] - poses / (ln["ms_per_step"] * 1e-3 * 6)) <= 1e-2 * ln["value"]
    ratio = ln["value"] / one_rank["value"]
    print(f"two ranks on one GPU, --split {split}: {ln['value']:.1f} poses/s = {ratio:.3f} of the one-rank {one_rank['value']:.1f}")
    assert floor <= ratio <= 1.15, ratio
    assert 0.0 < ln["roofline"]["frac"] <= 1.0
