"""Noise schedules, time embedding and `set_time` with the reference's names and argument meaning
(reference utils/diffusion_utils.py:21-32, 99-143, 150-179).  Host-side scalar logic; the per-step
tensors the reference allocates in `set_time` are replaced by scalars handed to the engine."""
from __future__ import annotations

import math

import numpy as np
import torch
from scipy.stats import beta


def t_to_sigma_individual(t, schedule_type, sigma_min, sigma_max):
    if schedule_type != "exponential":
        raise NotImplementedError(schedule_type)
    return sigma_min ** (1 - t) * sigma_max ** t


def t_to_sigma(t_tr, t_rot, t_tor, args):
    return (t_to_sigma_individual(t_tr, "exponential", args.tr_sigma_min, args.tr_sigma_max),
            t_to_sigma_individual(t_rot, "exponential", args.rot_sigma_min, args.rot_sigma_max),
            t_to_sigma_individual(t_tor, "exponential", args.tor_sigma_min, args.tor_sigma_max))


def sinusoidal_embedding(timesteps, embedding_dim, max_positions=10000):
    assert len(timesteps.shape) == 1
    half = embedding_dim // 2
    k = math.log(max_positions) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32, device=timesteps.device) * -k)
    arg = timesteps.float()[:, None] * freqs[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


def get_timestep_embedding(embedding_type, embedding_dim, embedding_scale=10000):
    if embedding_type != "sinusoidal":
        raise NotImplementedError(embedding_type)
    f = lambda x: sinusoidal_embedding(embedding_scale * x, embedding_dim)
    f.embedding_scale, f.embedding_dim = embedding_scale, embedding_dim
    return f


def get_t_schedule(sigma_schedule, inference_steps, inf_sched_alpha=1, inf_sched_beta=1, t_max=1):
    if sigma_schedule != "expbeta":
        raise Exception()
    lin_max = beta.cdf(t_max, a=inf_sched_alpha, b=inf_sched_beta)
    c = np.linspace(lin_max, 0, inference_steps + 1)[:-1]
    return beta.ppf(c, a=inf_sched_alpha, b=inf_sched_beta)


def get_inverse_schedule(t, sched_alpha=1, sched_beta=1):
    """Component time of the asynchronous noise schedule: the Beta(alpha, beta) quantile of the common time
    (utils/diffusion_utils.py:146-147; callers inference.py:387-389, finetune_train.py:138-140)."""
    return beta.ppf(t, a=sched_alpha, b=sched_beta)


def set_time(complex_graphs, t, t_tr, t_rot, t_tor, batchsize, all_atoms, asyncronous_noise_schedule, device,
             include_miscellaneous_atoms=False):
    """Attach the diffusion time to a batch.  Same fields as the reference (node_t / complex_t dicts of fp32
    tensors) so user code that reads them keeps working; the engine only consumes complex_t."""
    if all_atoms or include_miscellaneous_atoms:
        raise NotImplementedError("all_atoms / misc-atom time fields are outside the MI355X hot path (the confidence engine evaluates at t = 0)")
    kinds = (("tr", t_tr), ("rot", t_rot), ("tor", t_tor)) + ((("t", t),) if asyncronous_noise_schedule else ())
    for nt in ("ligand", "receptor"):
        n = complex_graphs[nt].num_nodes
        complex_graphs[nt].node_t = {k: v * torch.ones(n, device=device) for k, v in kinds}
    complex_graphs.complex_t = {k: v * torch.ones(batchsize, device=device) for k, v in kinds}
