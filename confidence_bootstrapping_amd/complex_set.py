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
        # two alternating sets of `group` engines (the partners share the first engine's weights): the complexes of the NEXT group are
        # set up on the idle set while the current group runs (cbd_set_complex with "async_setup", uploads on a side stream)
        self.engines = [DockEngine.from_model(score_model, self.dev, max_batch=self.samples)]
        for _ in range(2 * self.group - 1):
            p = DockEngine(self.dev, max_batch=self.samples, lm_embedding_dim=self.engines[0].cfg.lm_embedding_dim,
                           no_torsion=bool(self.engines[0].cfg.no_torsion))
            p.share_weights_from(self.engines[0])
            self.engines.append(p)
        for e in self.engines:
            e.set_option("async_setup", 1)
        self.sets = [self.engines[:self.group], self.engines[self.group:]]
        self.side = torch.cuda.Stream(self.dev)
        # event on the current stream recorded just before a group is launched: the side-stream fills of the NEXT group follow everything
        # queued up to it (their buffers come from the current stream's allocator pool and may have been freed by the host while work
        # queued there still reads them -- ADVICE round 5), but not the running group itself
        self._fill_after = None
        self._staged = None          # (indices, set, staged inputs) of the group prepared ahead
        self._turn = 0
        self.ceng = conf_model.engine(max_batch=self.samples) if conf_model is not None else None
        # one confidence engine per complex of a group: up to four complexes are scored in ONE set of fused-conv launches
        self.cengs = ([self.ceng] + conf_model.co_engines(min(self.group, 4) - 1, self.ceng)) if conf_model is not None else []
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

    def _stage(self, items, which, ahead=False):
        """set-up of every complex of a group on engine set `which` + its pose / noise uploads (device buffers from the current stream's
        pool, filled on the side stream: an allocation ON the side stream would wait for the running group)"""
        def up(t):
            t = t.to(torch.float32).contiguous()
            d = torch.empty(t.shape, dtype=torch.float32, device=self.dev)
            with torch.cuda.stream(self.side):
                d.copy_(t, non_blocking=True)
            return d
        if ahead and self._fill_after is not None:
            self.side.wait_event(self._fill_after)
        else:       # nothing of this runner is on the GPU: simply behind whatever the caller has queued
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
        staged = []
        for e, (i, c) in zip(self.sets[which], items):
            cplx, pos0, noise = self.prepared[i] if i in self.prepared else self.prepare(i, c)
            e.set_complex(cplx)
            staged.append((up(pos0), [up(z) for z in noise]))
        return staged

    def sample_group(self, items: Sequence, next_items: Sequence = None) -> List[dict]:
        """`items` = [(index, complex), ...] (<= group): set-up of every complex (graph upload, receptor embedding, all-atom tables),
        ONE co-scheduled sampling call, confidence ranking.  One picklable dict per complex.  `next_items`: the group that will be
        asked for next -- its set-up runs on the other engine set while this group's step loop is on the GPU."""
        ta = time.perf_counter()
        key = tuple(i for i, _ in items)
        if self._staged is not None and self._staged[0] == key:
            _, which, staged = self._staged
        else:
            which, staged = self._turn, self._stage(items, self._turn)
        self._staged = None
        self._turn = 1 - which
        engines = self.sets[which][:len(items)]
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_stream(self.side)
        self._fill_after = torch.cuda.Event()
        self._fill_after.record(cur)
        tb = time.perf_counter()
        if len(items) == 1:
            engines[0].sample(staged[0][0], self.steps, *staged[0][1])
        else:
            DockEngine.sample_multi(engines, [s[0] for s in staged], self.steps, [s[1] for s in staged])
        t_ahead = 0.0
        if next_items:
            t0 = time.perf_counter()
            self._staged = (tuple(i for i, _ in next_items), 1 - which, self._stage(next_items, 1 - which, ahead=True))
            t_ahead = time.perf_counter() - t0
        torch.cuda.synchronize(self.dev)
        tc = time.perf_counter()
        confs = [None] * len(items)
        if self.ceng is not None:
            from .engine import ConfidenceEngine
            for k0 in range(0, len(items), len(self.cengs)):      # cbd_conf_score_multi: bitwise the results of separate score() calls
                part = range(k0, min(k0 + len(self.cengs), len(items)))
                for ce, k in zip(self.cengs, part):
                    ce.set_complex(self.prepared[items[k][0]][0])
                got = ConfidenceEngine.score_multi(self.cengs[:len(part)], [staged[k][0] for k in part], self.conf_args.crop_beyond)
                for k, (c, _) in zip(part, got):
                    confs[k] = c
        out = []
        for (i, _), (pos, _), conf in zip(items, staged, confs):
            res = {"complex": i}
            if conf is not None:
                order = torch.argsort(conf, descending=True, stable=True)
                best = int(order[0])
                res.update(confidence=float(conf[best]), best=best, pos=pos[best].cpu().numpy(), order=order.cpu().numpy())
            else:
                res.update(confidence=None, best=0, pos=pos[0].cpu().numpy())
            if self.keep_poses:
                res["all_pos"] = pos.cpu()
            out.append(res)
        td = time.perf_counter()
        self.times["setup"] += tb - ta + t_ahead       # the part spent ahead ran under the step loop (inside "sample" as well)
        self.times["sample"] += tc - tb
        self.times["conf"] += td - tc
        return out
