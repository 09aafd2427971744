"""Golden vectors for the aggregate evaluation metrics: EXECUTES the reference's own block
inference.py:593-885 (it is inline in the script's `__main__`, so it is compiled from the file at generation time with the
surrounding names bound to synthetic inputs) and stores inputs + the resulting `performance_metrics` dictionary.
Run from the repo root:  python -m oracle.make_golden_eval      (needs /root/reference)"""
import os
import tempfile
import textwrap
from argparse import Namespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/inference.py"


def reference_block():
    lines = open(REF).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.strip() == "performance_metrics = {}")
    end = next(i for i, l in enumerate(lines) if i > start and l.strip() == "for k in performance_metrics:")
    return compile(textwrap.dedent("\n".join(lines[start:end])), REF, "exec")


def main():
    code = reference_block()
    out = {}
    rng = np.random.default_rng(3)
    for tag, (C, N, with_conf) in {"n12": (9, 12, True), "n6": (7, 6, True), "n3_noconf": (5, 3, False)}.items():
        rmsds = np.abs(rng.normal(3.0, 2.5, size=(C, N)))
        cent = np.abs(rng.normal(1.5, 1.5, size=(C, N)))
        selfd = np.abs(rng.normal(1.0, 0.5, size=(C, N)))
        conf = rng.normal(size=(C, N))
        overlap = rng.random(C) < 0.6
        overlap[0] = True
        times = np.abs(rng.normal(10, 2, size=C))
        with tempfile.TemporaryDirectory() as tmp:
            ns = dict(np=np, args=Namespace(out_dir=tmp, filtering_model_dir="x" if with_conf else None), N=N,
                      filtering_model=object() if with_conf else None,
                      rmsds_list=list(rmsds), centroid_distances_list=list(cent), min_self_distances_list=list(selfd),
                      confidences_list=list(conf) if with_conf else [], names_list=[f"c{i}" for i in range(C)],
                      without_rec_overlap_list=list(overlap.astype(int)), run_times=list(times), print=lambda *a, **k: None)
            exec(code, ns)
        pm = ns["performance_metrics"]
        out.update({f"{tag}_rmsds": rmsds, f"{tag}_centroid": cent, f"{tag}_self": selfd, f"{tag}_conf": conf, f"{tag}_overlap": overlap,
                    f"{tag}_times": times, f"{tag}_keys": np.array(list(pm.keys())), f"{tag}_vals": np.array([float(pm[k]) for k in pm])})
        print(tag, len(pm), "metrics")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g13_eval.npz"), **out)


if __name__ == "__main__":
    main()
