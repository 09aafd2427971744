"""The cost-weighted work split of the register-stationary bf16 kernel (csrc/tp_conv_bf16s.hip, round 6) only decides WHICH workgroup runs a
32-edge unit: every setting of the unit costs must give bitwise the same scores.  Runs on the diagnostic library (the overrides
CBD_S_WEIGHTS / CBD_S_EQUAL_UNITS exist there only), one fresh process per setting:
    python -m pytest experiments/test_split_weights.py -q          (on a GPU box)"""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")

CODE = (
    "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
    "from tools.diag_lib import use_diag_library; use_diag_library()\n"
    "from confidence_bootstrapping_amd.synthetic import make_workload\n"
    "from confidence_bootstrapping_amd.utils import make_score_model\n"
    "from confidence_bootstrapping_amd.engine import make_steps\n"
    "m, a = make_score_model(device='cuda:0', seed=0); c = make_workload('c2_dockgen_median'); e = m.engine(); e.set_complex(c)\n"
    "e.set_option('bf16', 1); e.set_option('bf16_stationary', 1)\n"
    "g = torch.Generator().manual_seed(0)\n"
    "p = (c['ligand'].pos[None].repeat(6, 1, 1) + 2 * torch.randn(6, 1, 3, generator=g)).cuda()\n"
    "out = e.score(p, make_steps(np.array([0.5]), a, m.timestep_emb_func)[0])\n"
    "print(' '.join(float(x).hex() for t in out for x in t.reshape(-1).cpu()))\n") % ROOT


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_scores_do_not_depend_on_the_unit_costs():
    base = _run({})
    assert len(base.split()) >= 18
    for extra in ({"CBD_S_EQUAL_UNITS": "1"}, {"CBD_S_WEIGHTS": "90,64,80,70"}, {"CBD_S_WEIGHTS": "64,90,64,64"}, {"CBD_S_WEIGHTS": "255,1,3,200"}):
        assert _run(extra) == base, extra
