"""A SET of complexes on one rank (BASELINE.json configs[2]; reference inference.py:409-580 walks its test loader one complex at a
time): per-complex set-up, reverse diffusion of `samples` poses x `steps` denoising steps in co-scheduled groups of up to four
complexes (ONE cbd_sample_multi call per group), confidence model on the final poses, ranking by confidence
(inference.py:537-547).  `distributed.run_complex_set` partitions the set over the ranks (LPT) and calls `sample_group` here.

Used by tools/run_set.py, bench.py (`complex_set` leg) and tests/test_gpu_configs.py, so that the number the bench prints and the
parity test are about the same code.
"""
from __future__ import annotations

import copy
import time
from typing import Dict, List, Sequence

import numpy as np
import torch

from .engine import DockEngine, make_steps
from .hetero import Batch
from .sampling import randomize_position


class ComplexSetRunner:
    """Engines (one score engine + `group - 1` partners sharing its weights, one confidence engine) and the per-group work."""

    def __init__(self, score_model, score_args, conf_model, conf_args, device, samples=40, denoise_steps=20, group=4, keep_poses=False):
        from .diffusion_utils import get_t_schedule
        self.dev = torch.device(device)
        self.samples, self.S, self.group = int(samples), int(denoise_steps), max(1, min(int(group), 8))
        self.score_args, self.conf_args = score_args, conf_args
        self.sched = get_t_schedule("expbeta", self.S)
        self.steps = make_steps(self.sched, score_args, score_model.timestep_emb_func)
        self.engines = [DockEngine.from_model(score_model, self.dev, max_batch=self.samples)]
        for _ in range(self.group - 1):
            p = DockEngine(self.dev, max_batch=self.samples, lm_embedding_dim=self.engines[0].cfg.lm_embedding_dim,
                           no_torsion=bool(self.engines[0].cfg.no_torsion))
            p.share_weights_from(self.engines[0])
            self.engines.append(p)
        self.ceng = conf_model.engine(max_batch=self.samples) if conf_model is not None else None
        self.keep_poses = keep_poses
        self.times = {"setup": 0.0, "sample": 0.0, "conf": 0.0}
        self.prepared: Dict[int, tuple] = {}

    def set_option(self, name, value):
        for e in self.engines:
            e.set_option(name, value)

    def prepare(self, i, cplx):
        """Host-side inputs of complex i (data loading, outside any timed region): initial poses by the reference's
        randomize_position and pre-drawn N(0,1) noise, both seeded by the complex index -> independent of the partitioning."""
        state = (np.random.get_state(), torch.random.get_rng_state())
        try:
            torch.manual_seed(i)
            np.random.seed(i)
            dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(self.samples)]
            randomize_position(dl, False, False, self.score_args.tr_sigma_max)
            pos0 = torch.stack([d["ligand"].pos for d in dl]).contiguous()
            R = int(cplx["ligand"].edge_mask.sum())
            noise = (torch.randn(self.S, self.samples, 3), torch.randn(self.S, self.samples, 3), torch.randn(self.S, self.samples * R))
        finally:
            np.random.set_state(state[0])
            torch.random.set_rng_state(state[1])
        self.prepared[i] = (cplx, pos0, noise)
        return self.prepared[i]

    def sample_group(self, items: Sequence) -> List[dict]:
        """`items` = [(index, complex), ...] (<= group): set-up of every complex (graph upload, receptor embedding, all-atom tables),
        ONE co-scheduled sampling call, confidence ranking.  One picklable dict per complex."""
        ta = time.perf_counter()
        engines = self.engines[:len(items)]
        staged = []
        for e, (i, c) in zip(engines, items):
            cplx, pos0, noise = self.prepared[i] if i in self.prepared else self.prepare(i, c)
            e.set_complex(cplx)
            staged.append((pos0.to(self.dev), [z.to(self.dev) for z in noise]))
        torch.cuda.synchronize(self.dev)
        tb = time.perf_counter()
        if len(items) == 1:
            engines[0].sample(staged[0][0], self.steps, *staged[0][1])
        else:
            DockEngine.sample_multi(engines, [s[0] for s in staged], self.steps, [s[1] for s in staged])
        torch.cuda.synchronize(self.dev)
        tc = time.perf_counter()
        out = []
        for (i, _), (pos, _) in zip(items, staged):
            res = {"complex": i}
            if self.ceng is not None:
                self.ceng.set_complex(self.prepared[i][0])
                conf, _ = self.ceng.score(pos, self.conf_args.crop_beyond)
                order = torch.argsort(conf, descending=True, stable=True)
                best = int(order[0])
                res.update(confidence=float(conf[best]), best=best, pos=pos[best].cpu().numpy(), order=order.cpu().numpy())
            else:
                res.update(confidence=None, best=0, pos=pos[0].cpu().numpy())
            if self.keep_poses:
                res["all_pos"] = pos.cpu()
            out.append(res)
        td = time.perf_counter()
        self.times["setup"] += tb - ta
        self.times["sample"] += tc - tb
        self.times["conf"] += td - tc
        return out
