"""ctypes binding of libcbdock.so (include/cbdock.h) and the glue between the reference-shaped Python API
(`TensorProductScoreModel.forward(batch)`, `sampling()`) and the C ABI.

PyTorch is used here only for device memory and streams.  There is no CPU or eager fallback: if the HIP library
cannot be loaded every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import so3, torus

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcbdock.so")
_lib = None


class cbd_config(C.Structure):
    _fields_ = [("ns", C.c_int32), ("nv", C.c_int32), ("num_conv_layers", C.c_int32), ("num_prot_emb_layers", C.c_int32),
                ("lm_embedding_dim", C.c_int32), ("no_torsion", C.c_int32), ("lig_max_radius", C.c_float),
                ("rec_max_radius", C.c_float), ("cross_max_distance", C.c_float), ("center_max_distance", C.c_float),
                ("lig_radius_cap", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32)]


class cbd_step(C.Structure):
    _fields_ = [("t", C.c_float), ("tr_sigma", C.c_float), ("cross_cutoff", C.c_float), ("rot_score_norm", C.c_float),
                ("tor_score_norm_sqrt", C.c_float), ("tr_score_coef", C.c_float), ("tr_noise_coef", C.c_float),
                ("rot_score_coef", C.c_float), ("rot_noise_coef", C.c_float), ("tor_score_coef", C.c_float),
                ("tor_noise_coef", C.c_float), ("sigma_emb", C.c_float * 32), ("sigma_emb_t", C.c_float * 32)]


class cbd_conf_config(C.Structure):
    _fields_ = [("ns", C.c_int32), ("nv", C.c_int32), ("num_conv_layers", C.c_int32), ("lm_embedding_dim", C.c_int32),
                ("lig_max_radius", C.c_float), ("cross_cutoff", C.c_float), ("lig_radius_cap", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32)]


# every symbol include/cbdock.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "cbd_last_error": (C.c_char_p, []),
    "cbd_version": (C.c_char_p, []),
    "cbd_create": (C.c_int, [C.POINTER(cbd_config), C.POINTER(_P)]),
    "cbd_destroy": (C.c_int, [_P]),
    "cbd_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32]),
    "cbd_finalize_weights": (C.c_int, [_P]),
    "cbd_set_complex": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_score": (C.c_int, [_P, C.c_int32, _P, C.POINTER(cbd_step), _P, _P, _P, _P]),
    "cbd_modify_conformer": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_sample": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(cbd_step), _P, _P, _P, _P, _P, _P]),
    "cbd_sample_pair": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(cbd_step)] + [_P] * 9),
    "cbd_sample_multi": (C.c_int, [C.c_int32, _P, C.POINTER(C.c_int32), C.c_int32, C.POINTER(cbd_step), _P, _P, _P, _P, _P]),
    "cbd_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "cbd_share_weights": (C.c_int, [_P, _P]),
    "cbd_recompute_receptor": (C.c_int, [_P, _P]),
    "cbd_stats": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_uint64)]),
    "cbd_debug_fetch": (C.c_int64, [_P, C.c_char_p, _P, C.c_int64]),
    "cbd_last_edge_counts": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "cbd_kernel_timing": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "cbd_conv_stream_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "cbd_pack_conv_stream": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_conv_stream_floats_infer": (C.c_int64, [C.c_int32, C.c_int32]),
    "cbd_pack_conv_stream_infer": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_symm_rmsd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_knn_graph": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P]),
    "cbd_radius_neighbors": (C.c_int, [C.c_int32, C.c_float, C.c_int32, _P, _P, _P, _P]),
    "cbd_conf_create": (C.c_int, [C.POINTER(cbd_conf_config), C.POINTER(_P)]),
    "cbd_conf_destroy": (C.c_int, [_P]),
    "cbd_conf_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32]),
    "cbd_conf_finalize_weights": (C.c_int, [_P]),
    "cbd_conf_set_complex": (C.c_int, [_P] + [C.c_int32] * 6 + [_P] * 10),
    "cbd_conf_score": (C.c_int, [_P, C.c_int32, _P, C.c_float, _P, _P, _P]),
    "cbd_conf_score_multi": (C.c_int, [C.c_int32, _P, C.POINTER(C.c_int32), _P, C.c_float, _P, _P, _P]),
    "cbd_conf_check": (C.c_int, [_P]),
    "cbd_conf_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "cbd_conf_debug_fetch": (C.c_int64, [_P, C.c_char_p, _P, C.c_int64]),
    "cbd_conf_last_edge_counts": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "cbd_conf_kernel_timing": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "cbd_conf_stream_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "cbd_conf_pack_stream": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_tp_packed_width": (C.c_int64, [C.c_int32, C.c_int32]),
    "cbd_tp_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_tp_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_tp_backward_dw": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P, _P, _P, _P, C.c_int32, _P, _P]),
    "cbd_tp_backward_gh": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_outer_accum_part_floats": (C.c_int64, []),
    "cbd_outer_accum": (C.c_int, [C.c_int64, _P, _P, C.c_int32, _P, _P]),
    "cbd_segment_sum": (C.c_int, [C.c_int64, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_segment_mean": (C.c_int, [C.c_int64, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_tp_backward_dw_groups": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_fc1_forward": (C.c_int, [C.c_int32, _P, _P, _P, _P, C.c_float, _P, C.c_int64, _P, _P]),
    "cbd_fc1_backward": (C.c_int, [C.c_int32, _P, _P, _P, _P, C.c_float, _P, _P, _P]),
    "cbd_outer_accum_groups": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P]),
    "cbd_partial_reduce": (C.c_int, [C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "cbd_linear_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_float, _P, C.c_int64, _P, _P]),
    "cbd_linear_backward_chunks": (C.c_int64, [C.c_int64]),
    "cbd_linear_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P, C.c_int32, C.c_float, _P, _P, _P, _P]),
    "cbd_irreps_bn_forward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P]),
    "cbd_irreps_bn_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_segment_mean_backward": (C.c_int, [C.c_int64, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_segment_sum_ld": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_edge_cat": (C.c_int, [C.c_int64, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "cbd_edge_cat_backward": (C.c_int, [C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_gather_pad": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "cbd_center_tp_forward": (C.c_int, [C.c_int64, _P, C.c_int32, _P, _P, _P, _P]),
    "cbd_center_tp_backward": (C.c_int, [C.c_int64, _P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "cbd_bond_tp_forward": (C.c_int, [C.c_int64, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "cbd_bond_tp_backward": (C.c_int, [C.c_int64, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "cbd_score_loss": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [_P] * 9 + [C.c_float] * 3 + [_P] * 5),
    "cbd_edge_geometry": (C.c_int, [C.c_int64, _P, _P, _P, _P, C.c_int32, _P, C.c_float, _P, _P, _P, _P]),
    "cbd_radius_count": (C.c_int, [C.c_int64, _P, _P, _P, C.c_float, _P, _P, C.c_int64, C.c_int32, _P, _P]),
    "cbd_radius_fill": (C.c_int, [C.c_int64, _P, _P, _P, C.c_float, _P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "cbd_csr_build_batched": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t), _P]),
    "cbd_csr_build": (C.c_int, [C.c_int64, C.c_int64, _P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t), _P]),
}


def load_library(path: Optional[str] = None):
    """dlopen the in-tree HIP library and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build the HIP engine first (python __graft_entry__.py); "
                           "there is no CPU fallback for the score model / sampler")
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"cbdock error {rc}: {load_library().cbd_last_error().decode()}")


def _dptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _hptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def make_steps(t_schedule, model_args, timestep_emb_func, ode=False, no_random=False, no_final_step_noise=False,
               temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5, rot_schedule=None, tor_schedule=None, common_t_schedule=None):
    """Per-step host scalars, computed with the reference's own scalar arithmetic and dtypes
    (utils/sampling.py:94-167; models/score_model.py:338,347,419-420,447).  Returns a ctypes array of cbd_step.
    `t_schedule` is the translation schedule; `rot_schedule` / `tor_schedule` default to it (inference.py:393-396) and differ
    under --different_schedules (inference.py:375-383): the model embeds the TRANSLATION time only (score_model.py:323,499),
    sigma_rot(t_rot) / sigma_tor(t_tor) enter through the score normalisers and the SDE coefficients of their component.
    `common_t_schedule`: for a model built with asyncronous_noise_schedule (inference.py:384-388: the three component schedules are
    beta-quantile images of ONE common time grid t) -- the ligand side and the magnitude heads then embed t instead of t_tr
    (score_model.py:408,460,497), the receptor side keeps t_tr."""
    S = len(t_schedule)
    scheds = [np.asarray(t_schedule, dtype=np.float64),
              np.asarray(t_schedule if rot_schedule is None else rot_schedule, dtype=np.float64),
              np.asarray(t_schedule if tor_schedule is None else tor_schedule, dtype=np.float64)]
    if any(len(x) != S for x in scheds):
        raise ValueError("tr / rot / tor schedules must have the same length")
    steps = (cbd_step * S)()
    a = model_args
    temp_sampling = list(temp_sampling) if np.iterable(temp_sampling) else [temp_sampling] * 3
    temp_psi = list(temp_psi) if np.iterable(temp_psi) else [temp_psi] * 3
    lims = [(a.tr_sigma_min, a.tr_sigma_max), (a.rot_sigma_min, a.rot_sigma_max), (a.tor_sigma_min, a.tor_sigma_max)]
    for i in range(S):
        t = scheds[0][i]
        st = steps[i]
        st.t = float(t)
        # model side: complex_t[k] is an fp32 tensor, t_to_sigma evaluated on it
        cts = [float(sc[i]) * torch.ones(1) for sc in scheds]
        sig_t = [lo ** (1 - ct) * hi ** ct for (lo, hi), ct in zip(lims, cts)]
        st.tr_sigma = float(sig_t[0][0])
        st.cross_cutoff = float((sig_t[0] * 3 + 20)[0])
        st.rot_score_norm = float(so3.score_norm(sig_t[1])[0])
        st.tor_score_norm_sqrt = float(torch.sqrt(torch.tensor(torus.score_norm(sig_t[2].numpy())).float())[0])
        emb = timestep_emb_func(cts[0])[0]
        emb_t = emb if common_t_schedule is None else timestep_emb_func(float(np.asarray(common_t_schedule, dtype=np.float64)[i]) * torch.ones(1))[0]
        for k in range(32):
            st.sigma_emb[k] = float(emb[k])
            st.sigma_emb_t[k] = float(emb_t[k])
        # sampler side: float64 sigma, fp32 g (0-dim tensor), python/numpy scalars for dt -- per component on its own schedule
        noise_on = not (no_random or ode or (no_final_step_noise and i == S - 1))
        coefs = []
        for k, (lo, hi) in enumerate(lims):
            tk = scheds[k][i]
            dt = scheds[k][i] - scheds[k][i + 1] if i < S - 1 else scheds[k][i]
            sigma = lo ** (1 - tk) * hi ** tk
            g = sigma * torch.sqrt(torch.tensor(2 * np.log(hi / lo)))
            if ode:
                sc, nc = 0.5 * g ** 2 * dt, 0.0
            elif temp_sampling[k] != 1.0:
                sigma_data = np.exp(temp_sigma_data * np.log(hi) + (1 - temp_sigma_data) * np.log(lo))
                lam = (sigma_data + sigma) / (sigma_data + sigma / temp_sampling[k])
                sc = g ** 2 * dt * (lam + temp_sampling[k] * temp_psi[k] / 2)
                nc = g * np.sqrt(dt * (1 + temp_psi[k]))
            else:
                sc, nc = g ** 2 * dt, g * np.sqrt(dt)
            coefs.append((float(sc), float(nc) if noise_on else 0.0))
        (st.tr_score_coef, st.tr_noise_coef), (st.rot_score_coef, st.rot_noise_coef), (st.tor_score_coef, st.tor_noise_coef) = coefs
    return steps


class DockEngine:
    """One engine per (model weights, device).  Holds the C handle; methods mirror the C ABI."""

    def __init__(self, device: torch.device, max_batch: int = 64, lm_embedding_dim: int = 1280, no_torsion: bool = False,
                 lig_max_radius=5.0, rec_max_radius=30.0, cross_max_distance=80.0, center_max_distance=30.0):
        self.lib = load_library()
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the docking engine runs on an MI355X (device type 'cuda' under ROCm); got " + str(device))
        self.device = device
        cfg = cbd_config(32, 6, 5, 3, lm_embedding_dim, int(no_torsion), lig_max_radius, rec_max_radius, cross_max_distance,
                         center_max_distance, 32, max_batch, device.index or 0)
        self.cfg = cfg
        h = C.c_void_p()
        _check(self.lib.cbd_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.max_batch = max_batch
        self.complex_key = None
        self.Nl = self.Nr = self.R = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.cbd_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- weights
    def load_state_dict(self, sd):
        for k, v in sd.items():
            a = np.ascontiguousarray(v.detach().cpu().float().numpy())
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape) if a.ndim else (C.c_int64 * 1)(1)
            _check(self.lib.cbd_load_weight(self.h, k.encode(), _hptr(a), shape, a.ndim))
        _check(self.lib.cbd_finalize_weights(self.h))
        self.complex_key = None

    @classmethod
    def from_model(cls, model, device, max_batch: int = 64):
        eng = cls(device, max_batch=max_batch, lm_embedding_dim=1280 if model.lm_embedding_type == "precomputed" else 0,
                  no_torsion=model.no_torsion, lig_max_radius=model.lig_max_radius, rec_max_radius=model.rec_max_radius,
                  cross_max_distance=model.cross_max_distance, center_max_distance=model.center_max_distance)
        eng.load_state_dict(model.state_dict())
        return eng

    # ---- complex
    def set_complex(self, graph, key=None):
        """graph: one HeteroData-like complex (un-batched; a 1-graph Batch is fine)."""
        lig, rec = graph["ligand"], graph["receptor"]
        mr = lig.mask_rotate
        while isinstance(mr, (list, tuple)):
            mr = mr[0]
        mr = np.ascontiguousarray(np.asarray(mr), dtype=np.uint8)
        lig_x = np.ascontiguousarray(lig.x.cpu().numpy().astype(np.int64))
        bidx = np.ascontiguousarray(graph["ligand", "ligand"].edge_index.cpu().numpy().astype(np.int64))
        battr = np.ascontiguousarray(graph["ligand", "ligand"].edge_attr.cpu().float().numpy())
        emask = np.ascontiguousarray(lig.edge_mask.cpu().numpy().astype(np.uint8))
        rec_x = np.ascontiguousarray(rec.x.cpu().float().numpy())
        rec_pos = np.ascontiguousarray(rec.pos.cpu().float().numpy())
        ridx = np.ascontiguousarray(graph["receptor", "receptor"].edge_index.cpu().numpy().astype(np.int64))
        Nl, Nr, nbd, R, Err = lig_x.shape[0], rec_x.shape[0], bidx.shape[1], int(emask.sum()), ridx.shape[1]
        if rec_x.shape[1] != 1 + self.cfg.lm_embedding_dim:
            raise RuntimeError(f"receptor features have {rec_x.shape[1]} columns, expected {1 + self.cfg.lm_embedding_dim}")
        if mr.size and mr.shape != (R, Nl):
            raise RuntimeError(f"mask_rotate shape {mr.shape} != ({R}, {Nl})")
        _check(self.lib.cbd_set_complex(self.h, Nl, Nr, nbd, R, Err, _hptr(lig_x), _hptr(bidx), _hptr(battr), _hptr(emask),
                                        _hptr(mr) if mr.size else None, _hptr(rec_x), _hptr(rec_pos), _hptr(ridx)))
        self.Nl, self.Nr, self.R = Nl, Nr, R
        self.complex_key = key

    # ---- compute
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def score(self, pos: torch.Tensor, step: cbd_step):
        B = pos.shape[0]
        pos = pos.to(self.device, torch.float32).contiguous()
        tr = torch.empty(B, 3, device=self.device)
        rot = torch.empty(B, 3, device=self.device)
        tor = torch.empty(B * self.R, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib.cbd_score(self.h, B, _dptr(pos), C.byref(step), _dptr(tr), _dptr(rot), _dptr(tor), self._stream()))
        return tr, rot, tor

    def modify_conformer(self, pos, tr, rot, tor):
        B = pos.shape[0]
        pos = pos.to(self.device, torch.float32).contiguous().clone()
        f = lambda x: None if x is None else x.to(self.device, torch.float32).contiguous()
        tr, rot, tor = f(tr), f(rot), f(tor)
        with torch.cuda.device(self.device):
            _check(self.lib.cbd_modify_conformer(self.h, B, _dptr(pos), _dptr(tr), _dptr(rot), _dptr(tor), self._stream()))
        return pos

    def sample(self, pos, steps, noise_tr=None, noise_rot=None, noise_tor=None, return_scores=False):
        """In-place reverse diffusion of pos [B,Nl,3] (device tensor) over len(steps) steps."""
        B, S = pos.shape[0], len(steps)
        assert pos.is_cuda and pos.dtype == torch.float32 and pos.is_contiguous()
        f = lambda x: None if x is None else x.to(self.device, torch.float32).contiguous()
        noise_tr, noise_rot, noise_tor = f(noise_tr), f(noise_rot), f(noise_tor)
        scores = torch.empty(S, B * (6 + self.R), device=self.device) if return_scores else None
        with torch.cuda.device(self.device):
            _check(self.lib.cbd_sample(self.h, B, S, steps, _dptr(pos), _dptr(noise_tr), _dptr(noise_rot), _dptr(noise_tor),
                                       _dptr(scores), self._stream()))
        return scores

    def share_weights_from(self, other: "DockEngine"):
        """Use `other`'s device-resident weights (one copy in HBM/L2) instead of uploading this engine's own."""
        _check(self.lib.cbd_share_weights(self.h, other.h))

    def sample_pair(self, other: "DockEngine", pos, steps, noise, other_pos, other_noise):
        """In-place reverse diffusion of two batches (two complexes) in lockstep with merged tensor-product launches
        (cbd_sample_pair).  noise / other_noise: (tr, rot, tor) device tensors or None."""
        f = lambda x: None if x is None else x.to(self.device, torch.float32).contiguous()
        n0 = [f(x) for x in (noise or (None, None, None))]
        n1 = [f(x) for x in (other_noise or (None, None, None))]
        for p in (pos, other_pos):
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        with torch.cuda.device(self.device):
            _check(self.lib.cbd_sample_pair(self.h, other.h, pos.shape[0], other_pos.shape[0], len(steps), steps, _dptr(pos),
                                            _dptr(n0[0]), _dptr(n0[1]), _dptr(n0[2]), _dptr(other_pos), _dptr(n1[0]), _dptr(n1[1]),
                                            _dptr(n1[2]), self._stream()))

    @staticmethod
    def sample_multi(engines, poses, steps, noises):
        """cbd_sample_multi: up to 8 engines (sharing weights, each with its own complex) advanced in lockstep with merged
        tensor-product launches.  poses: list of [b,Nl,3] device tensors (updated in place); noises: list of (tr, rot, tor) or None."""
        n = len(engines)
        e0 = engines[0]
        f = lambda x: None if x is None else x.to(e0.device, torch.float32).contiguous()
        nz = [[f(x) for x in (nn or (None, None, None))] for nn in noises]
        for p in poses:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        arr = lambda vals: (C.c_void_p * n)(*[None if v is None else v.data_ptr() for v in vals])
        hs = (C.c_void_p * n)(*[e.h.value for e in engines])
        Bs = (C.c_int32 * n)(*[p.shape[0] for p in poses])
        with torch.cuda.device(e0.device):
            _check(e0.lib.cbd_sample_multi(n, hs, Bs, len(steps), steps, arr(poses), arr([z[0] for z in nz]), arr([z[1] for z in nz]),
                                           arr([z[2] for z in nz]), e0._stream()))

    def recompute_receptor(self):
        with torch.cuda.device(self.device):
            _check(self.lib.cbd_recompute_receptor(self.h, self._stream()))

    def set_option(self, name: str, value: int):
        _check(self.lib.cbd_set_option(self.h, name.encode(), int(value)))
        self.__dict__.setdefault("_options", {})[name] = int(value)

    def get_option(self, name: str, default=None):
        """the value last set through set_option (the library has no getter); `default` for an option never set"""
        return self.__dict__.get("_options", {}).get(name, default)

    def stats(self, reset=False):
        out = (C.c_uint64 * 4)()
        _check(self.lib.cbd_stats(self.h, int(reset), out))
        return {"ll_edges": out[0], "conv_edge_visits": out[1], "forwards": out[2], "shared_rr_visits": out[3]}

    # ---- introspection
    def debug(self, enable=True):
        self.lib.cbd_debug_fetch(self.h, b"enable" if enable else b"disable", None, 0)

    def fetch(self, name: str, capacity: int = 1 << 24):
        buf = np.empty(capacity, dtype=np.float32)
        n = self.lib.cbd_debug_fetch(self.h, name.encode(), _hptr(buf), capacity)
        if n < 0:
            raise RuntimeError(self.lib.cbd_last_error().decode())
        return buf[:n].copy()

    def edge_counts(self):
        c = (C.c_int64 * 5)()
        _check(self.lib.cbd_last_edge_counts(self.h, c))
        return dict(zip(("ll", "lr", "rr", "rl", "tor"), list(c)))

    def kernel_timing(self, enable=True, reset=False):
        avg, n, tot = C.c_double(), C.c_int64(), C.c_double()
        _check(self.lib.cbd_kernel_timing(self.h, int(enable), int(reset), C.byref(avg), C.byref(n), C.byref(tot)))
        return avg.value, n.value, tot.value


class DockEnginePool:
    """n engines with the same weights on one GPU, each on its own HIP stream.  A batch of independent pose samples is
    split into n contiguous chunks whose step loops run concurrently: the latency-bound small kernels and the tail of
    one chunk's tensor-product launches overlap with the other chunk's matrix-core work.  Results are identical to a
    single engine (samples never interact)."""

    def __init__(self, state_dict, device, n: int = 1, max_batch: int = 64, share_from=None, **engine_kw):
        self.device = torch.device(device)
        self.n = max(1, int(n))
        per = (max_batch + self.n - 1) // self.n
        self.engines = [DockEngine(self.device, max_batch=per, **engine_kw) for _ in range(self.n)]
        if share_from is not None:     # reuse the weights another pool already holds on this device
            self.engines[0].share_weights_from(share_from.engines[0])
        else:
            self.engines[0].load_state_dict(state_dict)
        for e in self.engines[1:]:     # one copy of the weights in HBM/L2, shared by all streams
            _check(e.lib.cbd_share_weights(e.h, self.engines[0].h))
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.n)]
        self.max_batch = per * self.n
        self.complex_key = None

    @classmethod
    def from_model(cls, model, device, n=1, max_batch=64, share_from=None):
        return cls(model.state_dict(), device, n=n, max_batch=max_batch, share_from=share_from,
                   lm_embedding_dim=1280 if model.lm_embedding_type == "precomputed" else 0, no_torsion=model.no_torsion,
                   lig_max_radius=model.lig_max_radius, rec_max_radius=model.rec_max_radius,
                   cross_max_distance=model.cross_max_distance, center_max_distance=model.center_max_distance)

    @property
    def R(self):
        return self.engines[0].R

    @property
    def Nl(self):
        return self.engines[0].Nl

    def set_complex(self, graph, key=None):
        for e in self.engines:
            e.set_complex(graph, key)
        self.complex_key = key

    def set_option(self, name, value):
        for e in self.engines:
            e.set_option(name, value)

    def get_option(self, name, default=None):
        return self.engines[0].get_option(name, default)

    def recompute_receptor(self):
        cur = torch.cuda.current_stream(self.device)
        for e, st in zip(self.engines, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                e.recompute_receptor()

    def _chunks(self, B):
        per = (B + self.n - 1) // self.n
        return [(lo, min(B, lo + per)) for lo in range(0, B, per)]

    def sample(self, pos, steps, noise_tr=None, noise_rot=None, noise_tor=None):
        B = pos.shape[0]
        if B > self.max_batch:
            raise RuntimeError(f"cbdock error -4: batch {B} exceeds max_batch {self.max_batch}")
        cur = torch.cuda.current_stream(self.device)
        R = self.R
        f = lambda x: None if x is None else x.to(self.device, torch.float32)
        noise_tr, noise_rot, noise_tor = f(noise_tr), f(noise_rot), f(noise_tor)
        for (lo, hi), e, st in zip(self._chunks(B), self.engines, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                e.sample(pos[lo:hi], steps,
                         None if noise_tr is None else noise_tr[:, lo:hi].contiguous(),
                         None if noise_rot is None else noise_rot[:, lo:hi].contiguous(),
                         None if noise_tor is None or R == 0 else noise_tor[:, lo * R:hi * R].contiguous())
        for st in self.streams:
            cur.wait_stream(st)

    def kernel_timing(self, enable=True, reset=False):
        tot_ms, n = 0.0, 0
        for e in self.engines:
            _, k, t = e.kernel_timing(enable, reset)
            tot_ms, n = tot_ms + t, n + k
        return (tot_ms / n if n else 0.0), n, tot_ms

    def stats(self, reset=False):
        out = {"ll_edges": 0, "conv_edge_visits": 0, "forwards": 0, "shared_rr_visits": 0}
        for e in self.engines:
            for k, v in e.stats(reset).items():
                out[k] += v
        return out


def pack_conv_stream(in_level, out_level, w1, b1, w2, b2, merged=False):
    """Host-only: the MFMA weight-tile stream of one FCBlock (for the CPU emulation tests).  `merged`: the layout the inference kernel
    reads (cbd_pack_conv_stream_infer); plain: the training kernels' (cbd_pack_conv_stream)."""
    lib = load_library()
    n = (lib.cbd_conv_stream_floats_infer if merged else lib.cbd_conv_stream_floats)(in_level, out_level)
    out = np.empty(n, dtype=np.float32)
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (w1, b1, w2, b2)]
    _check((lib.cbd_pack_conv_stream_infer if merged else lib.cbd_pack_conv_stream)(in_level, out_level, *[_hptr(a) for a in arrs], _hptr(out)))
    return out


def pack_fctp_stream(in_level, out_level, w1, b1, w2, b2):
    """Host-only: the MFMA weight-tile stream of one FCBlock of a confidence-model layer (CPU emulation tests)."""
    lib = load_library()
    n = lib.cbd_conf_stream_floats(in_level, out_level)
    out = np.empty(n, dtype=np.float32)
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (w1, b1, w2, b2)]
    _check(lib.cbd_conf_pack_stream(in_level, out_level, *[_hptr(a) for a in arrs], _hptr(out)))
    return out


CONF_GROUPS = ("ll", "lr", "la", "rr", "rl", "ra", "aa", "al", "ar")


class ConfidenceEngine:
    """All-atom confidence model on one GPU (cbd_conf_* of include/cbdock.h).  One engine per (weights, device)."""

    def __init__(self, device, max_batch: int = 64, lm_embedding_dim: int = 1280, lig_max_radius=5.0, cross_cutoff=20.0):
        self.lib = load_library()
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the confidence engine runs on an MI355X (device type 'cuda' under ROCm); got " + str(device))
        self.device = device
        self.cfg = cbd_conf_config(24, 6, 5, lm_embedding_dim, lig_max_radius, cross_cutoff, 32, max_batch, device.index or 0)
        h = C.c_void_p()
        _check(self.lib.cbd_conf_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.max_batch = max_batch
        self.complex_key = None
        self.Nl = self.Nr = self.Na = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.cbd_conf_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def load_state_dict(self, sd):
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(v.detach().cpu().float().numpy())
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape) if a.ndim else (C.c_int64 * 1)(1)
            _check(self.lib.cbd_conf_load_weight(self.h, k.encode(), _hptr(a), shape, a.ndim))
        _check(self.lib.cbd_conf_finalize_weights(self.h))
        self.complex_key = None

    @classmethod
    def from_model(cls, model, device, max_batch: int = 64):
        # dynamic_max_cross at t = 0: cutoff = 3 * 0 + 20 (models/all_atom_score_model.py:388)
        eng = cls(device, max_batch=max_batch, lm_embedding_dim=1280 if model.lm_embedding_type == "precomputed" else 0,
                  lig_max_radius=model.lig_max_radius, cross_cutoff=20.0)
        eng.load_state_dict(model.state_dict())
        return eng

    def set_complex(self, graph, key=None):
        """graph: one un-cropped complex carrying the all-atom stores (un-batched or a 1-graph Batch)."""
        f32 = lambda t: np.ascontiguousarray(t.cpu().float().numpy())
        i64 = lambda t: np.ascontiguousarray(t.cpu().numpy().astype(np.int64))
        lig_x, battr = f32(graph["ligand"].x), f32(graph["ligand", "ligand"].edge_attr)
        bidx = i64(graph["ligand", "ligand"].edge_index)
        rec_x, rec_pos, ridx = f32(graph["receptor"].x), f32(graph["receptor"].pos), i64(graph["receptor", "receptor"].edge_index)
        atom_x, atom_pos, aidx = f32(graph["atom"].x), f32(graph["atom"].pos), i64(graph["atom", "atom"].edge_index)
        ar = i64(graph["atom", "receptor"].edge_index)
        Nl, Nr, Na = lig_x.shape[0], rec_x.shape[0], atom_x.shape[0]
        if not np.array_equal(ar[0], np.arange(Na)):
            raise RuntimeError("('atom','atom_rec_contact','receptor').edge_index[0] must be arange(num_atoms)")
        if rec_x.shape[1] != 1 + self.cfg.lm_embedding_dim:
            raise RuntimeError(f"receptor features have {rec_x.shape[1]} columns, expected {1 + self.cfg.lm_embedding_dim}")
        if lig_x.shape[1] != 16 or atom_x.shape[1] != 4:
            raise RuntimeError("ligand / atom feature widths must be 16 / 4")
        ares = np.ascontiguousarray(ar[1])
        _check(self.lib.cbd_conf_set_complex(self.h, Nl, Nr, Na, bidx.shape[1], ridx.shape[1], aidx.shape[1], _hptr(lig_x), _hptr(bidx),
                                             _hptr(battr), _hptr(rec_x), _hptr(rec_pos), _hptr(ridx), _hptr(atom_x), _hptr(atom_pos),
                                             _hptr(aidx), _hptr(ares)))
        self.Nl, self.Nr, self.Na = Nl, Nr, Na
        self.complex_key = key

    def score(self, pos: torch.Tensor, crop_beyond=None, check=True):
        """pos [B,Nl,3] -> (confidence [B], atom_confidence [B*Nl, 1]) device tensors."""
        B = pos.shape[0]
        pos = pos.to(self.device, torch.float32).contiguous()
        conf = torch.empty(B, device=self.device)
        atom = torch.empty(B * self.Nl, device=self.device)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _check(self.lib.cbd_conf_score(self.h, B, _dptr(pos), float(crop_beyond or 0.0), _dptr(conf), _dptr(atom), stream))
            if check:
                _check(self.lib.cbd_conf_check(self.h))
        return conf, atom.unsqueeze(1)

    @staticmethod
    def score_multi(engines, poses, crop_beyond=None, check=True):
        """The pose batches of up to four complexes (one engine each, `poses[k]` [B_k, Nl_k, 3]) in ONE set of fused-conv launches
        (cbd_conf_score_multi; bitwise the results of separate score() calls) -> list of (confidence, atom_confidence)."""
        e0, n = engines[0], len(engines)
        if n == 1:
            return [e0.score(poses[0], crop_beyond, check=check)]
        poses = [p.to(e0.device, torch.float32).contiguous() for p in poses]
        conf = [torch.empty(p.shape[0], device=e0.device) for p in poses]
        atom = [torch.empty(p.shape[0] * e.Nl, device=e0.device) for p, e in zip(poses, engines)]
        hs = (C.c_void_p * n)(*[e.h for e in engines])
        Bs = (C.c_int32 * n)(*[int(p.shape[0]) for p in poses])
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        with torch.cuda.device(e0.device):
            stream = C.c_void_p(torch.cuda.current_stream(e0.device).cuda_stream)
            _check(e0.lib.cbd_conf_score_multi(n, hs, Bs, arr(poses), float(crop_beyond or 0.0), arr(conf), arr(atom), stream))
            if check:
                for e in engines:
                    _check(e.lib.cbd_conf_check(e.h))
        return [(c, a.unsqueeze(1)) for c, a in zip(conf, atom)]

    def check(self):
        """cbd_conf_check: synchronises and raises if any score() since the last check exceeded a per-atom edge capacity."""
        with torch.cuda.device(self.device):
            _check(self.lib.cbd_conf_check(self.h))

    def set_option(self, name: str, value: int):
        _check(self.lib.cbd_conf_set_option(self.h, name.encode(), int(value)))

    def fetch(self, name: str, capacity: int = 1 << 24):
        buf = np.empty(capacity, dtype=np.float32)
        n = self.lib.cbd_conf_debug_fetch(self.h, name.encode(), _hptr(buf), capacity)
        if n < 0:
            raise RuntimeError(self.lib.cbd_last_error().decode())
        return buf[:n].copy()

    def edge_counts(self):
        c = (C.c_int64 * 9)()
        _check(self.lib.cbd_conf_last_edge_counts(self.h, c))
        return dict(zip(CONF_GROUPS, list(c)))

    def kernel_timing(self, enable=True, reset=False):
        avg, n, tot = C.c_double(), C.c_int64(), C.c_double()
        _check(self.lib.cbd_conf_kernel_timing(self.h, int(enable), int(reset), C.byref(avg), C.byref(n), C.byref(tot)))
        return avg.value, n.value, tot.value


def _single_all_atom_complex(data):
    """Un-batched view of the (identical, un-cropped) all-atom complexes in a batch."""
    g, B, Nl = _single_complex(data) if "mask_rotate" in data["ligand"] else (None, data.num_graphs, data["ligand"].num_nodes // data.num_graphs)
    from .hetero import HeteroData
    if g is None:
        g = HeteroData()
        lig, rec = data["ligand"], data["receptor"]
        Nr = rec.num_nodes // B
        M, Err = data["ligand", "ligand"].num_edges // B, data["receptor", "receptor"].num_edges // B
        g["ligand"].x, g["ligand"].pos = lig.x[:Nl], lig.pos[:Nl]
        g["ligand", "ligand"].edge_index = data["ligand", "ligand"].edge_index[:, :M]
        g["ligand", "ligand"].edge_attr = data["ligand", "ligand"].edge_attr[:M]
        g["receptor"].x, g["receptor"].pos = rec.x[:Nr], rec.pos[:Nr]
        g["receptor", "receptor"].edge_index = data["receptor", "receptor"].edge_index[:, :Err]
    atom = data["atom"]
    Na = atom.num_nodes // B
    g["atom"].x, g["atom"].pos = atom.x[:Na], atom.pos[:Na]
    g["atom", "atom"].edge_index = data["atom", "atom"].edge_index[:, :data["atom", "atom"].num_edges // B]
    g["atom", "receptor"].edge_index = data["atom", "receptor"].edge_index[:, :Na]
    return g, B, Nl


def confidence_batch(model, data, crop_beyond=None):
    """All-atom TensorProductScoreModel.forward(batch) in confidence mode: (confidence [B], atom_confidence [B*Nl,1]).
    `data` holds B poses of ONE un-cropped complex; the crop (model.crop_beyond or the argument) runs on the GPU."""
    ct = getattr(data, "complex_t", None)
    if ct is not None and float(torch.as_tensor(ct["tr"]).abs().max()) != 0.0:
        raise NotImplementedError("the confidence engine evaluates at t = 0 (utils/sampling.py:253)")
    eng = model.engine(max_batch=max(64, data.num_graphs))
    g, B, Nl = _single_all_atom_complex(data)
    key = complex_fingerprint(data) + (g["atom"].pos.shape[0],)
    if eng.complex_key != key:
        eng.set_complex(g, key)
    crop = crop_beyond if crop_beyond is not None else getattr(model, "crop_beyond", None)
    return eng.score(data["ligand"].pos.reshape(B, Nl, 3), crop)


def _single_complex(data):
    """Un-batched view of the (identical) complexes in a batch: node/edge slices of graph 0."""
    B = data.num_graphs
    lig, rec = data["ligand"], data["receptor"]
    Nl, Nr = lig.num_nodes // B, rec.num_nodes // B
    M, Err = data["ligand", "ligand"].num_edges // B, data["receptor", "receptor"].num_edges // B
    from .hetero import HeteroData
    g = HeteroData()
    g["ligand"].x = lig.x[:Nl]
    g["ligand"].pos = lig.pos[:Nl]
    g["ligand"].edge_mask = lig.edge_mask[:M]
    mr = lig.mask_rotate
    while isinstance(mr, (list, tuple)):
        mr = mr[0]
    g["ligand"].mask_rotate = mr
    g["ligand", "ligand"].edge_index = data["ligand", "ligand"].edge_index[:, :M]
    g["ligand", "ligand"].edge_attr = data["ligand", "ligand"].edge_attr[:M]
    g["receptor"].x = rec.x[:Nr]
    g["receptor"].pos = rec.pos[:Nr]
    g["receptor", "receptor"].edge_index = data["receptor", "receptor"].edge_index[:, :Err]
    return g, B, Nl


def _copies_of_one_complex(data):
    """True when every graph of the batch is a pose of the same complex (what sampling() builds): equal names and equal node counts."""
    B = data.num_graphs
    if B <= 1:
        return True
    names = getattr(data, "name", None)
    if isinstance(names, (list, tuple)) and len(names) == B:
        flat = [n[0] if isinstance(n, (list, tuple)) and len(n) else n for n in names]
        if any(n != flat[0] for n in flat[1:]):
            return False
    for kind in ("ligand", "receptor"):
        b = getattr(data[kind], "batch", None)
        if b is None:
            continue
        cnt = torch.bincount(b.detach().cpu(), minlength=B)
        if not bool(torch.all(cnt == cnt[0])):
            return False
    return True


def complex_fingerprint(data):
    """Cheap identity of the complex a batch is made of (names + sizes), to skip re-uploading it every step."""
    name = getattr(data, "name", None)
    if isinstance(name, (list, tuple)):
        name = name[0]
        while isinstance(name, (list, tuple)):
            name = name[0]
    B = data.num_graphs
    Nl, Nr = data["ligand"].num_nodes // B, data["receptor"].num_nodes // B
    key = (name, Nl, Nr, data["ligand", "ligand"].num_edges // B, int(data["receptor"].pos[0].sum().item() * 1e3))
    if name is None:
        # unnamed graphs of equal shapes must not alias a cached receptor: hash the first graph's receptor trace and atom types
        import hashlib
        h = hashlib.blake2b(digest_size=8)
        h.update(np.ascontiguousarray(data["receptor"].pos[:Nr].detach().cpu().numpy()).tobytes())
        h.update(np.ascontiguousarray(data["ligand"].x[:Nl].detach().cpu().numpy()).tobytes())
        key += (h.hexdigest(),)
    return key


def score_batch(model, data):
    """TensorProductScoreModel.forward(batch): returns (tr_pred, rot_pred, tor_pred, None)."""
    eng = model.engine()
    ct = data.complex_t
    t_tr, t_rot, t_tor = (ct[k].detach().cpu() for k in ("tr", "rot", "tor"))
    if not all(bool(torch.all(x == x[0])) for x in (t_tr, t_rot, t_tor)) or not _copies_of_one_complex(data):
        # a diffusion time per complex or different complexes in one batch (validation batches, utils/training.py:236-252): the fused
        # engine advances copies of ONE complex at ONE time, so these go through the batched HIP forward of the fine-tuning path in eval mode (same kernels' arithmetic, no autograd)
        with torch.no_grad():
            return model.forward_train(data)
    g, B, Nl = _single_complex(data)
    key = complex_fingerprint(data)
    if eng.complex_key != key:
        eng.set_complex(g, key)
    common = None
    if getattr(model, "asyncronous_noise_schedule", False):
        t_c = ct["t"].detach().cpu()
        if not bool(torch.all(t_c == t_c[0])):
            raise NotImplementedError("per-sample diffusion times within one batch are outside the MI355X hot path")
        common = np.array([float(t_c[0])])
    steps = make_steps(np.array([float(t_tr[0])]), _ArgsFromModel(model), model.timestep_emb_func,
                       rot_schedule=np.array([float(t_rot[0])]), tor_schedule=np.array([float(t_tor[0])]), common_t_schedule=common)
    pos = data["ligand"].pos.reshape(B, Nl, 3)
    tr, rot, tor = eng.score(pos, steps[0])
    if model.no_torsion or eng.R == 0:
        tor = torch.empty(0, device=eng.device)
    return tr, rot, tor, None


class _ArgsFromModel:
    """sigma limits for a bare forward call: recovered from model.t_to_sigma (a partial over args)."""

    def __init__(self, model):
        a = getattr(model.t_to_sigma, "keywords", {}).get("args")
        if a is None:
            raise RuntimeError("model.t_to_sigma must be functools.partial(t_to_sigma, args=model_args)")
        for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max", "tor_sigma_min", "tor_sigma_max"):
            setattr(self, k, getattr(a, k))
