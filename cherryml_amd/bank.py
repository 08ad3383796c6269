"""CherryBank: a count-matrix bank resident on one MI355X, evaluated by
libcherrybank (HIP).  Thin, typed wrapper over the C ABI."""
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import CB_NO_SYNC, CB_NORMALIZE, CB_PTR_DEVICE, CB_TRAIN_RESUME


def _as_f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


class CherryBank:
    """Count matrices C[L,B,S,S] (or [B,S,S]) at branch lengths t[L,B] (or [B]),
    uploaded once to `device` and kept there (the reference re-uploads them
    every epoch: cherryml/estimation/_ratelearn/trainer.py:164-167)."""

    def __init__(self, t, C, device: int = 0, dtype: str = "f64", per_bucket_products: bool = False):
        """per_bucket_products (S > 32, cb_create's CB_PER_BUCKET_PRODUCTS): always form U^T G_b U bucket by bucket; by default a
        float64 bank with symmetric counts runs in a TIME BASIS from 28 live buckets on (products on ~30 skeleton buckets,
        csrc/tbasis.hip.h) and sums the buckets before the last product from 24 on (see include/cherrybank.h).
        dtype: element type of the bank products (cb_create's `dtype`): "f64"; "f32" -- the
        reference's own arithmetic, ratelearner.py:98,107 -- for S > 32 (float32 MFMA; the
        eigendecomposition, loss accumulation and everything crossing the ABI stay float64); or "mixed":
        P_b, loss and G_b in float64, the two contractions of the gradient on the float32 MFMA."""
        self._h = None
        codes = {"f64": _lib.CB_F64, "f32": _lib.CB_F32, "mixed": _lib.CB_MIXED}
        if dtype not in codes:
            raise ValueError(f'dtype must be "f64", "f32" or "mixed", got {dtype!r}')
        self.dtype = dtype
        code = codes[dtype]
        cflags = _lib.CB_PER_BUCKET_PRODUCTS if per_bucket_products else 0
        lib = _lib.load()
        if lib.cb_device_count() <= 0:
            raise _lib.CherryBankError("no HIP device visible; cherryml_amd has no CPU fallback")
        is_torch = hasattr(C, "data_ptr")
        if is_torch:
            import torch
            if not C.is_cuda:
                C = C.detach().cpu().numpy()
                t = t.detach().cpu().numpy() if hasattr(t, "detach") else t
                is_torch = False
        shape = tuple(C.shape)
        if len(shape) == 3:
            shape = (1,) + shape
        if len(shape) != 4 or shape[2] != shape[3]:
            raise ValueError(f"C must be [L,B,S,S] or [B,S,S], got {tuple(C.shape)}")
        self.L, self.B, self.S = shape[0], shape[1], shape[2]
        self.device = int(device)
        import ctypes as Ct
        h = Ct.c_void_p()
        if is_torch:
            import torch
            Cd = C.to(dtype=torch.float64).contiguous()
            td = torch.as_tensor(t, dtype=torch.float64, device=Cd.device).reshape(-1)
            td = td.expand(self.L, self.B).reshape(-1).contiguous() if td.numel() == self.B and self.L > 1 \
                else td.contiguous()
            if td.numel() != self.L * self.B:
                raise ValueError("t must have L*B (or B) entries")
            torch.cuda.synchronize(Cd.device)
            rc = lib.cb_create(Cd.device.index or 0, self.S, self.L, self.B, code, td.data_ptr(),
                               Cd.data_ptr(), CB_PTR_DEVICE | cflags, Ct.byref(h))
            self.device = Cd.device.index or 0
        else:
            Cn = _as_f64(C).reshape(self.L, self.B, self.S, self.S)
            tn = _as_f64(t).reshape(-1)
            if tn.size == self.B and self.L > 1:
                tn = np.tile(tn, self.L)
            if tn.size != self.L * self.B:
                raise ValueError("t must have L*B (or B) entries")
            if not (np.all(np.isfinite(Cn)) and np.all(np.isfinite(tn))):
                raise ValueError("non-finite counts or branch lengths")
            rc = lib.cb_create(self.device, self.S, self.L, self.B, code, tn.ctypes.data, Cn.ctypes.data,
                               cflags, Ct.byref(h))
        _lib.check(rc, "cb_create")
        self._h = h
        n = np.zeros(self.L)
        _lib.check(lib.cb_total_counts(self._h, n.ctypes.data), "cb_total_counts")
        self.total_counts = n
        nl = np.zeros(self.L, dtype=np.int32)
        _lib.check(lib.cb_live_buckets(self._h, nl.ctypes.data), "cb_live_buckets")
        self.live_buckets = nl      # non-empty buckets per site: the ones the loss visits

    def allreduce_setup(self, rccl_comm, nccl_allreduce_fn, n_total):
        """In-library multi-GPU reduction (cb_allreduce_setup): `rccl_comm` an ncclComm_t (int / c_void_p),
        `nccl_allreduce_fn` the address of ncclAllReduce of the RCCL that owns it, `n_total[L]` the global
        total counts.  Afterwards loss_grad / loss_grad_general return the sums over the ranks.
        rccl_comm=None switches it off."""
        import ctypes as Ct
        if rccl_comm is None:
            _lib.check(_lib.load().cb_allreduce_setup(self._h, None, None, None), "cb_allreduce_setup")
            return
        n = _as_f64(n_total).reshape(-1)
        if n.size != self.L:
            raise ValueError("n_total must have L entries")
        _lib.check(_lib.load().cb_allreduce_setup(self._h, Ct.c_void_p(int(rccl_comm)), Ct.c_void_p(int(nccl_allreduce_fn)),
                                                  n.ctypes.data), "cb_allreduce_setup")

    # -- lifetime ---------------------------------------------------------
    @classmethod
    def expm_only(cls, t, num_states: int, device: int = 0, num_sites: int = 1) -> "CherryBank":
        """A counts-free bank (cb_create with CB_EXPM_ONLY): branch lengths only, serves `expm_bank` / `eigh`; the loss and
        training entry points refuse it.  num_sites > 1: t[num_sites, B], one rate matrix per site."""
        import ctypes as Ct
        lib = _lib.load()
        if lib.cb_device_count() <= 0:
            raise _lib.CherryBankError("no HIP device visible; cherryml_amd has no CPU fallback")
        self = cls.__new__(cls)
        self._h = None
        self.dtype = "f64"
        tn = _as_f64(t).reshape(-1)
        L = int(num_sites)
        if L < 1 or tn.size % L:
            raise ValueError("expm_only: t must hold num_sites x B branch lengths")
        self.L, self.B, self.S, self.device = L, int(tn.size) // L, int(num_states), int(device)
        h = Ct.c_void_p()
        _lib.check(lib.cb_create(self.device, self.S, L, self.B, _lib.CB_F64, tn.ctypes.data, None, _lib.CB_EXPM_ONLY,
                                 Ct.byref(h)), "cb_create")
        self._h = h
        return self

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.load().cb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream: Optional[int], own: bool = False):
        """Run on `hip_stream` (0 / None = HIP's default stream, torch's default) or, with
        own=True, on the handle's private stream."""
        _lib.check(_lib.load().cb_set_stream(self._h, hip_stream or None, int(own)), "cb_set_stream")

    # -- profiling ----------------------------------------------------------
    TIMING_NAMES = ("total", "eigh", "k1", "k2", "k3", "k4", "small", "allreduce")

    def profile(self, enable: bool = True, every: int = 1):
        """phase events on (every: the C-driven 400-state trainer records them in every `every`-th epoch only) / off"""
        _lib.check(_lib.load().cb_profile(self._h, int(every) if enable else 0), "cb_profile")

    def last_timings(self) -> dict:
        """milliseconds per phase of the last profiled cb_loss_grad (HIP events)."""
        ms = np.zeros(len(self.TIMING_NAMES))
        _lib.check(_lib.load().cb_last_timings(self._h, ms.ctypes.data, ms.size), "cb_last_timings")
        return dict(zip(self.TIMING_NAMES, ms.tolist()))

    def timing_means(self) -> dict:
        """mean milliseconds per phase over the profiled calls since profile(True)."""
        import ctypes as Ct
        ms = np.zeros(len(self.TIMING_NAMES))
        calls = Ct.c_int(0)
        _lib.check(_lib.load().cb_timing_sums(self._h, ms.ctypes.data, ms.size, Ct.byref(calls)),
                   "cb_timing_sums")
        k = max(calls.value, 1)
        out = dict(zip(self.TIMING_NAMES, (ms / k).tolist()))
        out["calls"] = calls.value
        return out

    def last_sweeps(self) -> int:
        return int(_lib.load().cb_last_sweeps(self._h))

    def eigh_counters(self) -> dict:
        """planned (device-controlled) warm eigensolves of the S > 32 trainer: how many, how many had to be continued,
        sweeps of the last one (cb_eigh_counters)."""
        v = np.zeros(4, dtype=np.int32)
        _lib.check(_lib.load().cb_eigh_counters(self._h, v.ctypes.data, 4), "cb_eigh_counters")
        return {"planned_solves": int(v[0]), "stalls": int(v[1]), "last_sweeps": int(v[2]), "record_spins": int(v[3])}

    def epoch_times(self, num_epochs: int) -> np.ndarray:
        """seconds from the entry of the last train_* call to the end of each of its epochs on the device
        (cb_train_epoch_times; the reference's df_res `time` column, trainer.py:207-217)."""
        s = np.zeros(int(num_epochs))
        _lib.check(_lib.load().cb_train_epoch_times(self._h, s.ctypes.data, s.size), "cb_train_epoch_times")
        return s

    def last_kernel_form(self) -> int:
        """which trainer kernels the last train_* call launched (cb_last_kernel_form: 1000 + 100 TS + 10 sym + w3, ...)"""
        return int(_lib.load().cb_last_kernel_form(self._h))

    def last_bank_form(self) -> dict:
        """how the last S > 32 evaluation ran the bank products (cb_last_bank_form): one persistent launch or separate
        ones, four- or eight-wave tiles, buckets summed before the last product (symmetric counts) or a third product each"""
        v = int(_lib.load().cb_last_bank_form(self._h))
        return {"fused": bool(v & 1), "waves_per_tile": 8 if v & 2 else 4, "bucket_sum_first": bool(v & 4),
                "time_basis": bool(v & 8)}

    def time_basis_info(self) -> dict:
        """the time basis of the last evaluation that ran in one (cb_time_basis_info; csrc/tbasis.hip.h): skeleton buckets of
        the short-branch forward family, long-branch buckets that keep their own product, skeleton buckets of the gradient
        family, bases built so far, epochs repeated with per-bucket products, the spectral bound the basis serves"""
        import ctypes as Ct
        n = np.zeros(5, dtype=np.int32)
        rho = Ct.c_double(0.0)
        _lib.check(_lib.load().cb_time_basis_info(self._h, n.ctypes.data, Ct.byref(rho)), "cb_time_basis_info")
        return {"forward_skeleton": int(n[0]), "direct": int(n[1]), "gradient_skeleton": int(n[2]), "builds": int(n[3]),
                "repeated_epochs": int(n[4]), "rho_max": float(rho.value)}

    # -- host-pointer API (numpy) -----------------------------------------
    def _shape_Q(self, Q, pi):
        Q = _as_f64(Q).reshape(self.L, self.S, self.S)
        pi = _as_f64(pi).reshape(self.L, self.S)
        return Q, pi

    def loss_grad(self, Q, pi, normalize: bool = True, want_grad: bool = True
                  ) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """loss[L], dL/dQ[L,S,S] for reversible Q (w.r.t. pi).  numpy in/out."""
        Q, pi = self._shape_Q(Q, pi)
        loss = np.zeros(self.L)
        dQ = np.zeros_like(Q) if want_grad else None
        rc = _lib.load().cb_loss_grad(self._h, Q.ctypes.data, pi.ctypes.data,
                                      CB_NORMALIZE if normalize else 0, loss.ctypes.data,
                                      dQ.ctypes.data if want_grad else None)
        _lib.check(rc, "cb_loss_grad")
        return loss, dQ

    def expm_bank(self, Q, pi=None) -> np.ndarray:
        """P[l,b] = expm(t[l,b] Q[l]); pi=None uses the general (scaling-and-squaring) path."""
        Q = _as_f64(Q).reshape(self.L, self.S, self.S)
        pi_ptr = None
        if pi is not None:
            pi = _as_f64(pi).reshape(self.L, self.S)
            pi_ptr = pi.ctypes.data
        P = np.zeros((self.L, self.B, self.S, self.S))
        _lib.check(_lib.load().cb_expm_bank(self._h, Q.ctypes.data, pi_ptr, 0, P.ctypes.data),
                   "cb_expm_bank")
        return P

    def loss_grad_general(self, Q, normalize: bool = True, want_grad: bool = True):
        """loss[L], dL/dQ for ANY rate matrix (no reversibility assumed): scaling and squaring."""
        Q = _as_f64(Q).reshape(self.L, self.S, self.S)
        loss = np.zeros(self.L)
        dQ = np.zeros_like(Q) if want_grad else None
        rc = _lib.load().cb_loss_grad_general(self._h, Q.ctypes.data, CB_NORMALIZE if normalize else 0,
                                              loss.ctypes.data, dQ.ctypes.data if want_grad else None)
        _lib.check(rc, "cb_loss_grad_general")
        return loss, dQ

    def eigh(self, A) -> Tuple[np.ndarray, np.ndarray]:
        A = _as_f64(A).reshape(self.L, self.S, self.S)
        lam = np.zeros((self.L, self.S))
        U = np.zeros_like(A)
        _lib.check(_lib.load().cb_eigh(self._h, A.ctypes.data, 0, lam.ctypes.data, U.ctypes.data),
                   "cb_eigh")
        return lam, U

    # -- fused device-side optimisers (pande_reversible: any S; SiteRM: S <= 32) -----
    def train_pande_reversible(self, upper_diag, log_pi, mask=None, num_epochs=2000, lr=0.1,
                               do_adam=True, normalize=True, resume=False):
        """All epochs of the reference loop (trainer.py:156-218) on the device.
        L == 1: returns dict(loss[E], Q_best, Q_last, Q_pow2 {epoch: Q}, upper_diag, log_pi).
        L > 1 (S <= 32): L independent problems in one batched launch sequence -- upper_diag [L, S(S-1)/2],
        log_pi [L, S] -> loss [E, L], Q_best / Q_last [L, S, S] (no power-of-two snapshots).
        resume=True (S > 32): continue the optimisation the previous call on this bank ended (CB_TRAIN_RESUME:
        parameters, Adam state and best iterate stay on the device; `upper_diag` / `log_pi` are ignored on the way
        in; E epochs + E' resumed epochs = E + E' epochs of one call, bit for bit; no power-of-two snapshots)."""
        S, E, L = self.S, int(num_epochs), self.L
        nup = S * (S - 1) // 2
        up = _as_f64(upper_diag, (nup,) if L == 1 else (L, nup)).copy()
        lp = _as_f64(log_pi, (S,) if L == 1 else (L, S)).copy()
        mk = None if mask is None else _as_f64(mask, (S, S))
        n_pow2 = (max(E, 1).bit_length() if E > 0 else 0) if (L == 1 and not resume) else 0
        loss = np.zeros(E) if L == 1 else np.zeros((E, L))
        Qb, Ql = (np.zeros((S, S)), np.zeros((S, S))) if L == 1 else (np.zeros((L, S, S)), np.zeros((L, S, S)))
        Qp = np.zeros((max(n_pow2, 1), S, S))
        rc = _lib.load().cb_train_pande_reversible(
            self._h, up.ctypes.data, lp.ctypes.data, None if mk is None else mk.ctypes.data, E,
            float(lr), int(bool(do_adam)), (CB_NORMALIZE if normalize else 0) | (CB_TRAIN_RESUME if resume else 0),
            loss.ctypes.data, Qb.ctypes.data, Ql.ctypes.data, Qp.ctypes.data if n_pow2 else None, n_pow2)
        _lib.check(rc, "cb_train_pande_reversible")
        snaps = {1 << i: Qp[i] for i in range(n_pow2) if (1 << i) <= E}
        return dict(loss=loss, Q_best=Qb, Q_last=Ql, Q_pow2=snaps, upper_diag=up, log_pi=lp, time=self.epoch_times(E))

    def train_siterm(self, theta, Theta, num_epochs, lr=0.1):
        """SiteRM loop (_cherryml_vectorized.py:351-383) for all sites in one launch.
        Returns dict(res[L,N,N] best Q per site, loss_per_epoch_per_site[E,L], theta, Theta)."""
        L, N, E = self.L, self.S, int(num_epochs)
        th = _as_f64(theta, (L, N)).copy()
        Th = _as_f64(Theta, (L, N, N)).copy()
        res = np.zeros((L, N, N))
        lpeps = np.zeros((max(E, 1), L))
        rc = _lib.load().cb_train_siterm(self._h, th.ctypes.data, Th.ctypes.data, E, float(lr), 0,
                                         res.ctypes.data, lpeps.ctypes.data)
        _lib.check(rc, "cb_train_siterm")
        return dict(res=res, loss_per_epoch_per_site=lpeps[:E], theta=th, Theta=Th, time=self.epoch_times(E))

    # -- device-pointer API (torch ROCm tensors, zero copy) -----------------
    def loss_grad_torch(self, Q, pi, normalize: bool = True, want_grad: bool = True):
        """Q[L,S,S] / pi[L,S] float64 tensors on this bank's device; returns
        (loss[L], dQ[L,S,S]) tensors.  Runs on torch's current stream.
        pi=None: general (non-reversible) path."""
        import torch
        if not Q.is_cuda or (pi is not None and not pi.is_cuda):
            raise ValueError("loss_grad_torch needs tensors on the GPU")
        Qc = Q.detach().to(torch.float64).reshape(self.L, self.S, self.S).contiguous()
        loss = torch.empty(self.L, dtype=torch.float64, device=Q.device)
        dQ = torch.empty_like(Qc) if want_grad else None
        self.set_stream(torch.cuda.current_stream(Q.device).cuda_stream)
        flags = CB_PTR_DEVICE | CB_NO_SYNC | (CB_NORMALIZE if normalize else 0)
        if pi is None:
            rc = _lib.load().cb_loss_grad_general(self._h, Qc.data_ptr(), flags, loss.data_ptr(),
                                                  dQ.data_ptr() if want_grad else None)
            _lib.check(rc, "cb_loss_grad_general")
            return loss, dQ
        pic = pi.detach().to(torch.float64).reshape(self.L, self.S).contiguous()
        rc = _lib.load().cb_loss_grad(self._h, Qc.data_ptr(), pic.data_ptr(), flags, loss.data_ptr(),
                                      dQ.data_ptr() if want_grad else None)
        _lib.check(rc, "cb_loss_grad")
        return loss, dQ


def time_basis(t, rho_max: float) -> dict:
    """HOST-ONLY (no GPU): the interpolative decomposition over the branch-length grid `t` (ascending) that the S > 32 bank
    uses for spectra inside [-rho_max, 0] (cb_time_basis, include/cherrybank.h): kind[B] (-1: expanded in the short-branch
    family, k >= 0: long-branch bucket k), the skeleton buckets, the interpolation matrices and the builder's residuals."""
    t = _as_f64(t).reshape(-1)
    B = t.size
    n = np.zeros(3, dtype=np.int32)
    kind = np.zeros(B, dtype=np.int32)
    sks, skg = np.zeros(24, dtype=np.int32), np.zeros(40, dtype=np.int32)
    Ls, Lg, res = np.zeros((B, 24)), np.zeros((B, 40)), np.zeros(2)
    _lib.check(_lib.load().cb_time_basis(B, t.ctypes.data, float(rho_max), n.ctypes.data, kind.ctypes.data, sks.ctypes.data,
                                         skg.ctypes.data, Ls.ctypes.data, Lg.ctypes.data, res.ctypes.data), "cb_time_basis")
    ns, nd, ng = (int(x) for x in n)
    return {"ns": ns, "nd": nd, "ng": ng, "kind": kind, "skel_s": sks[:ns].copy(), "skel_g": skg[:ng].copy(),
            "Ls": Ls[:, :ns].copy(), "Lg": Lg[:, :ng].copy(), "residuals": res}
