"""Where does a run-to-run difference first appear?  One score-model forward on the C4 workload is repeated from identical inputs with the
per-layer debug snapshots on; every snapshot is compared bitwise with run 0.  GPU box only.
usage: python tools/race_hunt.py [runs] [f32|bf16|f32_split] [batch]"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from confidence_bootstrapping_amd import Batch
from confidence_bootstrapping_amd.synthetic import make_workload
from confidence_bootstrapping_amd.utils import make_score_model
from confidence_bootstrapping_amd.engine import DockEngine, make_steps
from confidence_bootstrapping_amd.sampling import randomize_position

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
model, args = make_score_model(device=dev, seed=0)
cplx = make_workload("c4_large_pocket")
eng = DockEngine.from_model(model, dev, max_batch=B)
eng.set_complex(cplx)
torch.manual_seed(12); np.random.seed(12)
dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
randomize_position(dl, False, False, args.tr_sigma_max)
pos0 = torch.stack([d["ligand"].pos for d in dl]).to(dev)
pos0 = pos0 * 0.25 + torch.as_tensor(np.asarray(cplx["ligand"].pos.mean(0)), device=dev) * 0.75   # near the pocket: many cross edges
step = make_steps(np.array([0.4]), args, model.timestep_emb_func)[0]
if mode != "f32":
    eng.set_option(mode, 1)
eng.debug(True)
NAMES = ["lig_node_emb0", "lig_emb_0", "lig_emb_1", "lig_emb_2", "conv_0", "conv_0_rec", "conv_1", "conv_1_rec", "conv_2", "conv_2_rec",
         "conv_3", "conv_3_rec", "conv_4"]
ref, bad = None, 0
for k in range(n):
    tr, rot, tor = eng.score(pos0.contiguous(), step)
    torch.cuda.synchronize()
    cur = {nm: eng.fetch(nm).copy() for nm in NAMES}
    cur["tr"], cur["rot"], cur["tor"] = tr.cpu().numpy(), rot.cpu().numpy(), tor.cpu().numpy()
    if ref is None:
        ref = cur
        print("edge counts", eng.edge_counts(), flush=True)
        continue
    diffs = [(nm, int((cur[nm] != ref[nm]).sum()), float(np.abs(cur[nm] - ref[nm]).max())) for nm in cur if not np.array_equal(cur[nm], ref[nm])]
    if diffs:
        bad += 1
        first = diffs[0]
        rows = np.nonzero((cur[first[0]] != ref[first[0]]).reshape(-1, 80).any(1))[0] if cur[first[0]].size % 80 == 0 else []
        print(f"run {k}: first difference in {first[0]}: {first[1]} values, max |d| {first[2]:.3e}; rows {list(rows[:12])} "
              f"({len(rows)} rows); all differing: {[d[0] for d in diffs]}", flush=True)
print(f"{mode} B={B}: {n} runs, {bad} differing (CBD_NO_SIDE={'1' if os.environ.get('CBD_NO_SIDE') else '0'})", flush=True)
