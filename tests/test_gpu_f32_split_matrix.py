"""`f32_split` as a first-class operand mode: the ORACLE / reference-golden parity matrix of the exact-fp32 kernel re-run with
CBD_PRECISION=2 (every engine created in the child process then uses fp32 operands split into three exact bf16 planes on the bf16
matrix cores) -- against the oracle and the reference's own goldens, with the UNCHANGED fp32 tolerances of those tests:
  reference golden forward g6 (rel 2e-5), per-layer intermediates vs oracle, the reference's 20-step trajectory (1e-3 A), the
  reference-shaped Python API under the reference's seed, the C2-sized complex + invariances, configs[0] (1a0q trajectory), configs[3]'s
  complex in fp32 arithmetic, the three-schedule golden g14, the 14 randomised complexes of the fuzz file, and configs[1]'s own headline
  workload (tests/test_gpu_configs.py).
The matrix runs in a fresh child process because the operand policy is a creation-time default of the engines (the hook is an
environment variable so that the parity tests themselves stay untouched)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATRIX = ["tests/test_gpu_parity.py::test_forward_matches_reference_golden",
          "tests/test_gpu_parity.py::test_intermediates_match_oracle",
          "tests/test_gpu_parity.py::test_sampling_matches_reference_trajectory",
          "tests/test_gpu_parity.py::test_python_api_sampling_matches_reference",
          "tests/test_gpu_parity.py::test_median_workload_and_invariants",
          "tests/test_gpu_parity.py::test_config_c1_1a0q_single_sample_trajectory",
          "tests/test_gpu_parity.py::test_config_c4_large_pocket_forward",
          "tests/test_gpu_parity.py::test_sampling_with_different_schedules_matches_reference",
          "tests/test_gpu_fuzz.py::test_random_complex_forward",
          "tests/test_gpu_configs.py::test_config_c2_headline_workload_vs_oracle"]


def test_oracle_parity_matrix_in_f32_split_mode():
    env = dict(os.environ, CBD_PRECISION="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + MATRIX, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1800)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail and "skipped" not in tail, tail


def test_the_hook_really_selects_the_split_kernels():
    """the child-process matrix is only meaningful if CBD_PRECISION=2 changes the arithmetic: same poses, two fresh processes, the
    scores must differ in the last bits (fp32-grade, not bitwise fp32) and agree to 2e-5"""
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from confidence_bootstrapping_amd.synthetic import make_workload\n"
            "from confidence_bootstrapping_amd.utils import make_score_model\n"
            "from confidence_bootstrapping_amd.engine import make_steps\n"
            "m, a = make_score_model(device='cuda:0', seed=0); c = make_workload('tiny'); e = m.engine(); e.set_complex(c)\n"
            "p = c['ligand'].pos[None].repeat(3, 1, 1).cuda() + torch.arange(3, device='cuda')[:, None, None] * 0.7\n"
            "tr, rot, tor = e.score(p, make_steps(np.array([0.6]), a, m.timestep_emb_func)[0])\n"
            "print(' '.join(repr(float(x)) for x in torch.cat([tr.reshape(-1), rot.reshape(-1), tor.reshape(-1)]).cpu()))\n") % ROOT
    outs = []
    for prec in ("0", "2"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CBD_PRECISION=prec), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([float(x) for x in r.stdout.strip().splitlines()[-1].split()])
    a, b = outs
    assert a != b
    scale = max(abs(x) for x in a)
    assert max(abs(x - y) for x, y in zip(a, b)) <= 2e-5 * scale
