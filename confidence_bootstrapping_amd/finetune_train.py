"""The confidence-bootstrapping loop (reference finetune_train.py:133-349): alternate
  (1) `inference_epoch`: for every target complex sample `inference_samples` poses by reverse diffusion (the fused MI355X
      engine), score them with the confidence model (all-atom engine), compute symmetry-corrected RMSDs when the crystal pose is
      known, and keep the poses whose confidence exceeds `confidence_cutoff`;
  (2) push them into the `CBBuffer`, and
  (3) `train_epoch` on the buffer through `NoiseTransform` (differentiable HIP path), with EMA weights used for inference.
Same function names, argument meaning and metric keys as the reference; checkpoints / wandb / dataset construction (which need
rdkit, PyG datasets and the MOAD files) are the caller's business here: `inference_finetune` takes the complexes directly.
"""
from __future__ import annotations

import copy
import traceback
from functools import partial

import numpy as np
import torch

from .diffusion_utils import get_inverse_schedule, get_t_schedule
from .hetero import Batch
from .molecules_utils import get_symmetry_rmsd, remove_all_hs
from .sampling import randomize_position, sampling
from .training import loss_function, train_epoch
from .hostcfg import with_glue_threads


def _copy(g):
    """the reference deep-copies the complex once per pose (finetune_train.py:169); poses only re-bind `pos`, so sharing the
    tensors is equivalent and 100x cheaper"""
    return g.shallow_copy() if hasattr(g, "shallow_copy") else copy.deepcopy(g)


def _as_batch1(g):
    return g if isinstance(g, Batch) else Batch.from_data_list([g])


@with_glue_threads
def inference_epoch(model, filtering_model, complex_graphs, filtering_complex_dict, device, t_to_sigma, args, filtering_args,
                    confidence_cutoff):
    """Sample + score every complex; returns (metrics, [(graph, confidence), ...] above the cutoff, top-confidence RMSDs)
    (reference finetune_train.py:133-245)."""
    t_schedule = get_t_schedule(sigma_schedule="expbeta", inference_steps=args.inference_steps, inf_sched_alpha=1, inf_sched_beta=1)
    asyn = bool(getattr(args, "asyncronous_noise_schedule", False))
    if asyn:      # finetune_train.py:137-140: every component runs on its own Beta quantile of the common time grid
        tr_schedule = get_inverse_schedule(t_schedule, args.sampling_alpha, args.sampling_beta)
        rot_schedule = get_inverse_schedule(t_schedule, args.rot_alpha, args.rot_beta)
        tor_schedule = get_inverse_schedule(t_schedule, args.tor_alpha, args.tor_beta)
    else:
        tr_schedule = rot_schedule = tor_schedule = t_schedule
    rmsds, min_rmsds, top_rmsds, confidences_list, complexes_to_keep = [], [], [], [], []
    model.eval()
    n = args.inference_samples

    def run(items, bs):
        """one sampling() call over the poses of all `items` (complexes): consecutive complexes are co-scheduled on the GPU"""
        flat = [g for it in items for g in it[1]]
        filt = [g for it in items for g in it[2]] if items[0][2] is not None else None
        preds, conf = sampling(data_list=flat, model=model, inference_steps=args.inference_steps, tr_schedule=tr_schedule,
                               rot_schedule=rot_schedule, tor_schedule=tor_schedule, device=device, t_to_sigma=t_to_sigma, model_args=args,
                               confidence_model=filtering_model, filtering_data_list=filt, filtering_model_args=filtering_args,
                               asyncronous_noise_schedule=asyn, t_schedule=t_schedule, batch_size=bs)
        return [(preds[k * n:(k + 1) * n], None if conf is None else conf[k * n:(k + 1) * n]) for k in range(len(items))]

    prepared = []
    for orig in complex_graphs:
        orig = _as_batch1(orig)
        name = orig.name[0] if isinstance(orig.name, (list, tuple)) else orig.name
        filtering_data_list = None
        if filtering_model is not None and filtering_complex_dict is not None:
            if name not in filtering_complex_dict:
                print(f"HAPPENING | The filtering dataset did not contain {name}. We are skipping this complex.")
                continue
            filtering_data_list = [_copy(filtering_complex_dict[name]) for _ in range(n)]
        data_list = [_copy(orig) for _ in range(n)]
        randomize_position(data_list, args.no_torsion, False, args.tr_sigma_max,
                           pocket_knowledge=getattr(args, "inf_pocket_knowledge", False), pocket_cutoff=getattr(args, "inf_pocket_cutoff", 7))
        prepared.append((orig, data_list, filtering_data_list))

    results = []
    group = 8 if n % max(args.inference_batch_size, 1) == 0 else 1      # loader batches must not straddle complexes
    for k in range(0, len(prepared), group):
        items = prepared[k:k + group]
        try:
            results.extend(zip(items, run(items, args.inference_batch_size)))
            continue
        except Exception as e:
            print("Exception while running inference on a group of complexes, retrying one by one:", e)
        for it in items:           # the reference's per-complex halve-and-retry protocol (finetune_train.py:176-196)
            out, failed, bs = None, 0, args.inference_batch_size
            while out is None and failed <= 5:
                try:
                    out = run([it], bs)[0]
                except Exception as e:
                    failed += 1
                    bs = max(bs // 2, 1)
                    print("Exception while running inference on complex:", e)
                    traceback.print_exc()
            if out is None:
                print("failed 5 times - skipping the complex")
                continue
            results.append((it, out))

    for (orig, _, _), (predictions_list, confidences) in results:
        ligand_pos = np.asarray([g["ligand"].pos.cpu().numpy() for g in predictions_list])
        if confidences is not None and isinstance(getattr(filtering_args, "rmsd_classification_cutoff", None), list):
            confidences = confidences[:, 0]
        orig_pos = getattr(orig["ligand"], "orig_pos", None)
        if orig_pos is not None:   # crystal pose known: RMSD metrics
            if isinstance(orig_pos, list):
                orig_pos = orig_pos[0]
            orig_pos = np.asarray(orig_pos, dtype=np.float32)
            orig_pos = orig_pos[None] if orig_pos.ndim == 2 else orig_pos
            filterHs = torch.not_equal(predictions_list[0]["ligand"].x[:, 0], 0).cpu().numpy()
            lp = ligand_pos[:, filterHs]
            ref = orig_pos[:, filterHs] - orig.original_center.cpu().numpy()
            mol = getattr(orig, "mol", None)
            mol = mol[0] if isinstance(mol, (list, tuple)) else mol
            mol = remove_all_hs(mol)          # RemoveAllHs(orig_complex_graph.mol[0]) in the reference; the coordinates are filtered with filterHs
            per_ref = []
            for r in ref:
                try:
                    per_ref.append(np.asarray(get_symmetry_rmsd(mol, r, [l for l in lp], device=device)))
                except Exception as e:
                    print("Using non corrected RMSD because of the error:", e)
                    per_ref.append(np.sqrt(((lp - r) ** 2).sum(axis=2).mean(axis=1)))
            rmsd = np.min(np.asarray(per_ref), axis=0)
            rmsds.extend(rmsd.tolist())
            min_rmsds.append(rmsd.min())
            if confidences is not None:
                top_rmsds.append(rmsd[int(np.argmax(np.asarray([float(c) for c in confidences])))])
            if getattr(args, "oracle_confidence", False):
                confidences = -4 * np.tanh(2 * rmsd / 3 - 2)
        if confidences is None:
            continue
        confidences_list.extend(float(c) for c in confidences)
        complexes_to_keep.extend((predictions_list[i], float(confidences[i])) for i in range(len(predictions_list))
                                 if float(confidences[i]) > confidence_cutoff)
    rmsds, min_rmsds, top_rmsds, conf = (np.asarray(x, dtype=np.float64) for x in (rmsds, min_rmsds, top_rmsds, confidences_list))
    pct = lambda a, thr, n: float(100 * (a < thr).sum() / n) if n else None
    losses = {"rmsds_lt2": pct(rmsds, 2, len(rmsds)), "rmsds_lt5": pct(rmsds, 5, len(rmsds)),
              "filtered_rmsds_lt2": pct(top_rmsds, 2, len(min_rmsds)), "filtered_rmsds_lt5": pct(top_rmsds, 5, len(min_rmsds)),
              "min_rmsds_lt2": pct(min_rmsds, 2, len(min_rmsds)), "min_rmsds_lt5": pct(min_rmsds, 5, len(min_rmsds)),
              "avg_confidence": float(conf.mean()) if len(conf) else None,
              "median_confidence": float(np.median(conf)) if len(conf) else None}
    return losses, complexes_to_keep, top_rmsds


class _Loader:
    """DataListLoader stand-in: shuffled lists of `batch_size` transformed buffer items per epoch."""

    def __init__(self, dataset, batch_size, shuffle=True, drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last

    def __iter__(self):
        n = len(self.dataset)
        order = np.random.permutation(n) if self.shuffle else np.arange(n)
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size]
            if self.drop_last and len(idx) < self.batch_size:
                break
            yield [self.dataset[int(k)] for k in idx]

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size


def inference_finetune(args, model, filtering_model, filtering_args, filtering_complex_dict, confidence_cutoff, optimizer, ema_weights,
                       finetune_dataset, target_complexes, t_to_sigma, device, log=print):
    """The outer loop (reference finetune_train.py:248-349).  `finetune_dataset` is a CBBuffer whose transform is a NoiseTransform,
    `target_complexes` the graphs of the target cluster.  Returns the per-epoch logs."""
    loss_fn = partial(loss_function, tr_weight=args.tr_weight, rot_weight=args.rot_weight, tor_weight=args.tor_weight,
                      no_torsion=args.no_torsion)
    history, loader = [], None
    for epoch in range(args.n_epochs):
        logs = {}
        ema_weights.store(model.parameters())
        if args.use_ema:
            ema_weights.copy_to(model.parameters())    # inference with the EMA weights
        if epoch % args.cb_inference_freq == 0:
            inf_dataset = list(target_complexes)[:args.num_inference_complexes]
            iterations = args.initial_iterations if epoch == 0 else args.inference_iterations
            complexes, metrics = [], None
            for _ in range(iterations):
                m, kept, _ = inference_epoch(model, filtering_model, inf_dataset, filtering_complex_dict, device, t_to_sigma, args,
                                             filtering_args, confidence_cutoff)
                metrics = metrics or {k: [] for k in m}
                for k in m:
                    if m[k] is not None:
                        metrics[k].append(m[k])
                complexes.extend(kept)
            finetune_dataset.add_complexes(complexes)
            loader = _Loader(finetune_dataset, args.batch_size, shuffle=True, drop_last=getattr(args, "dataloader_drop_last", False))
            logs.update({"targetinf_" + k: (float(np.mean(v)) if v else None) for k, v in metrics.items()})
            logs["kept"] = len(complexes)
            logs["buffer"] = len(finetune_dataset.complexes)
        ema_weights.restore(model.parameters())
        if loader is not None and len(finetune_dataset.complexes) > 1:
            train_losses = train_epoch(model, loader, optimizer, device, t_to_sigma, loss_fn, ema_weights)
            logs.update({"train_" + k: v for k, v in train_losses.items()})
        log(f"Epoch {epoch}: " + ", ".join(f"{k} {v:.4f}" if isinstance(v, float) else f"{k} {v}" for k, v in logs.items()
                                         if k in ("kept", "buffer", "train_loss", "targetinf_avg_confidence")))
        history.append(logs)
    return history
