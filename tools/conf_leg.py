"""The confidence leg of bench.py on its own (random poses around the ligand):   python tools/conf_leg.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from confidence_bootstrapping_amd.synthetic import make_workload, BENCH_GEOMETRY
dev = torch.device("cuda:0")
c = make_workload("c2_dockgen_median", seed=1234, all_atoms=True, **BENCH_GEOMETRY)
g = torch.Generator().manual_seed(0)
pos = (c["ligand"].pos[None] + 1.0 * torch.randn(40, 1, 3, generator=g) + 0.3 * torch.randn(40, 28, 3, generator=g)).to(dev)
print(json.dumps(bench.confidence_leg("c2_dockgen_median", 40, 1234, pos, dev, dict(BENCH_GEOMETRY))))
