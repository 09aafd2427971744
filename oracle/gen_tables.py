"""Generate the SO(3) / torus score-normaliser tables by IMPORTING the reference's own
modules (this container only; /root/reference never travels to the GPU box).

TEST/BUILD INFRASTRUCTURE -- not part of the product path.

  reference utils/so3.py:46-65   -> _exp_score_norms[2000]  (deterministic series)
  reference utils/torus.py:65-75 -> score_norm_[5001]       (Monte-Carlo, UNSEEDED in the
                                     reference; we seed numpy with TORUS_SEED and record it)

Both modules build their tables at import time and cache ~460 MB of .npy files in the
*current directory*, so this script chdir()s to a scratch dir first.

Outputs (data, not source):
  confidence_bootstrapping_amd/data/so3_exp_score_norms.npy   float64 [2000]
  confidence_bootstrapping_amd/data/torus_score_norm.npy      float64 [5001]
  confidence_bootstrapping_amd/data/tables_meta.json
Run:  python oracle/gen_tables.py [scratch_dir]
"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

REF = "/root/reference"
TORUS_SEED = 0
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "confidence_bootstrapping_amd", "data")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    scratch = sys.argv[1] if len(sys.argv) > 1 else "/tmp/cb_tables"
    os.makedirs(scratch, exist_ok=True)
    os.makedirs(OUT, exist_ok=True)
    os.chdir(scratch)
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    meta = {}
    if which in ("both", "torus"):
        t0 = time.time()
        np.random.seed(TORUS_SEED)
        torus = _load("ref_torus", os.path.join(REF, "utils", "torus.py"))
        tab = np.asarray(torus.score_norm_, dtype=np.float64)
        assert tab.shape == (5001,)
        np.save(os.path.join(OUT, "torus_score_norm.npy"), tab)
        meta["torus"] = {"seed": TORUS_SEED, "secs": time.time() - t0,
                         "score_norm(0.0314)": float(torus.score_norm(np.array([0.0314]))[0])}
        print("torus done", meta["torus"], flush=True)
    if which in ("both", "so3"):
        t0 = time.time()
        so3 = _load("ref_so3", os.path.join(REF, "utils", "so3.py"))
        tab = np.asarray(so3._exp_score_norms, dtype=np.float64)
        assert tab.shape == (2000,)
        np.save(os.path.join(OUT, "so3_exp_score_norms.npy"), tab)
        import torch
        meta["so3"] = {"secs": time.time() - t0,
                       "score_norm": {str(e): float(so3.score_norm(torch.tensor([e]))[0]) for e in (0.06, 1.0, 3.1)}}
        print("so3 done", meta["so3"], flush=True)
    mp = os.path.join(OUT, "tables_meta.json")
    old = json.load(open(mp)) if os.path.exists(mp) else {}
    old.update(meta)
    json.dump(old, open(mp, "w"), indent=1)


if __name__ == "__main__":
    main()
