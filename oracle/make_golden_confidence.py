"""Generate tests/golden/g8_confidence.npz by RUNNING THE REFERENCE's all-atom confidence model (this container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python oracle/make_golden_confidence.py   (needs /root/reference).
Executes, from the reference: utils.utils.get_model (all-atom class, confidence_mode), utils.utils.crop_beyond per pose,
utils.diffusion_utils.set_time(.., 0, 0, 0, 0, ..) and the model's forward -- the sequence utils/sampling.py:240-256
performs -- on the synthetic 'tiny' complex with the all-atom stores (synthetic.add_atoms).  Third-party pieces are the
shims listed in oracle/ref_import.py (e3nn, torch_cluster, torch_scatter, torch_geometric containers + subgraph).
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def main():
    from oracle import ref_import
    from oracle.make_golden import npz
    hetero = ref_import.install(load_tables=False)
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_confidence_model
    import utils.utils as ref_utils
    from utils.diffusion_utils import set_time

    torch.set_num_threads(8)
    mine, cargs = make_confidence_model(seed=5)
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    ref_model, _ = ref_import.reference_confidence_model(sd)
    arrs = {}
    for wl, B in (("tiny", 4),):
        cplx = make_workload(wl, all_atoms=True)
        g = torch.Generator().manual_seed(99)
        Nl = cplx["ligand"].pos.shape[0]
        base = cplx["ligand"].pos
        shifts = torch.tensor([[0.0, 0, 0], [4.0, -3.0, 2.0], [-9.0, 6.0, 5.0], [14.0, 10.0, -12.0]])[:B]
        pos = torch.stack([base + shifts[b] + 0.4 * torch.randn(Nl, 3, generator=g) for b in range(B)])
        graphs = []
        for b in range(B):
            gph = copy.deepcopy(cplx)
            gph["ligand"].pos = pos[b].clone()
            ref_utils.crop_beyond(gph, cargs.crop_beyond, cargs.all_atoms)
            graphs.append(gph)
        batch = hetero.Batch.from_data_list(graphs)
        set_time(batch, 0, 0, 0, 0, B, cargs.all_atoms, False, torch.device("cpu"))
        layer_out = []
        hooks = [m.register_forward_hook(lambda mod, inp, out: layer_out.append(out.detach().clone())) for m in ref_model.conv_layers]
        with torch.no_grad():
            conf, atom_conf = ref_model(batch)
        for h in hooks:
            h.remove()
        for l, o in enumerate(layer_out):   # ligand rows come first in the joint node tensor (all_atom_score_model.py:398)
            arrs[f"{wl}_lig_layer{l + 1}"] = o[:B * Nl]
        arrs.update({f"{wl}_pos": pos, f"{wl}_confidence": conf, f"{wl}_atom_confidence": atom_conf,
                     f"{wl}_n_res": np.array([gph["receptor"].pos.shape[0] for gph in graphs]),
                     f"{wl}_n_atom": np.array([gph["atom"].pos.shape[0] for gph in graphs])})
        print(wl, "kept residues", arrs[f"{wl}_n_res"], "atoms", arrs[f"{wl}_n_atom"], "confidence", conf)
    npz("g8_confidence.npz", **arrs)


if __name__ == "__main__":
    main()
