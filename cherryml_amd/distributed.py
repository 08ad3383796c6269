"""One process per GPU.

Two ways to build the sharded bank:
* `ShardedBank(t, C)`: every rank sees the whole host array and keeps its own buckets.
* `ShardedBank.from_rank_counts(t, C_rank)`: every rank holds the sufficient statistics of ITS OWN
  families (the counting stage is family-sharded, SURVEY 8e option 2); one reduce-scatter over the
  buckets sums them and leaves each rank with exactly the buckets it owns -- no rank ever holds
  the summed [B,S,S] tensor.

Then, per epoch: the bank's NON-EMPTY buckets are dealt round-robin to the ranks
(k-th non-empty bucket -> rank k mod world), every rank evaluates the partial loss and
partial dL/dQ of its own buckets, and ONE all-reduce (RCCL over xGMI with the
"nccl" backend; gloo in the CPU tests) of S*S + 1 float64 values per epoch sums
them (option 1).  The parameters and the optimiser are
replicated: identical inputs -> identical Adam steps, no broadcast needed.
Counts (C) never cross GPUs during the epochs."""
import ctypes
from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist


def bucket_shard(num_buckets: int, rank: int, world: int) -> np.ndarray:
    """Indices of the buckets owned by `rank`."""
    return np.arange(rank, num_buckets, world)


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _uid_to_bytes(uid: "_NcclUniqueId") -> bytes:
    """All 128 bytes (attribute access on a c_char array stops at the first NUL)."""
    return ctypes.string_at(ctypes.byref(uid), ctypes.sizeof(uid))


def _uid_from_bytes(raw: bytes) -> "_NcclUniqueId":
    if len(raw) != ctypes.sizeof(_NcclUniqueId):
        raise ValueError("ncclUniqueId must be 128 bytes")
    uid = _NcclUniqueId()
    ctypes.memmove(ctypes.byref(uid), raw, len(raw))
    return uid


class RcclCommunicator:
    """A raw RCCL communicator for the in-library all-reduce (`cb_allreduce_setup`): created through
    ctypes on the librccl that torch itself loaded, one rank per process / GPU; the unique id travels
    through `torch.distributed` (any backend).  Without an initialised process group it is a
    single-rank communicator.  `comm` is the ncclComm_t, `allreduce_fn` the address of ncclAllReduce."""

    def __init__(self, group=None):
        import ctypes as C
        import glob
        import os
        # the RCCL torch itself uses (one library, one set of kernels); CHERRYML_AMD_RCCL_LIB overrides
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        world = dist.get_world_size(group) if on else 1
        UniqueId = _NcclUniqueId
        uid = UniqueId()
        error, rccl = None, None
        try:   # the rank-local part; its outcome is exchanged before the first blocking RCCL call
            libs = ([os.environ["CHERRYML_AMD_RCCL_LIB"]] if os.environ.get("CHERRYML_AMD_RCCL_LIB") else
                    sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))))
            if not libs:
                raise RuntimeError("RcclCommunicator: no librccl next to torch")
            rccl = C.CDLL(libs[0])
            if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
        except Exception as exc:
            if not (on and world > 1):
                raise
            error = f"{type(exc).__name__}: {exc}"
        if on and world > 1:
            # one exchange carries rank 0's id and every rank's status: ncclCommInitRank blocks until all
            # ranks have called it, so no rank may enter it while another one has already given up
            boxes = [None] * world
            dist.all_gather_object(boxes, (error, _uid_to_bytes(uid) if rank == 0 and error is None else None), group=group)
            bad = [(r, e) for r, (e, _) in enumerate(boxes) if e is not None]
            if bad:
                raise RuntimeError(f"RcclCommunicator: rank {bad[0][0]} failed: {bad[0][1]}")
            uid = _uid_from_bytes(boxes[0][1])
        self._rccl = rccl
        comm = C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
        if rc != 0 or not comm.value:
            raise RuntimeError(f"ncclCommInitRank failed with code {rc}")
        self.comm = comm.value
        self.allreduce_fn = C.cast(rccl.ncclAllReduce, C.c_void_p).value
        self.rank, self.world = rank, world
        # the communicator's own rank count (what a bench line may call n_gpus), checked against the group's
        n = C.c_int(-1)
        rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        rc = rccl.ncclCommCount(C.c_void_p(self.comm), C.byref(n))
        if rc != 0 or n.value != world:
            self.destroy()
            raise RuntimeError(f"ncclCommCount says {n.value} rank(s) (code {rc}), the process group has {world}")
        self.count = n.value

    def destroy(self):
        if getattr(self, "comm", None):
            import ctypes as C
            self._rccl.ncclCommDestroy.argtypes = [C.c_void_p]
            self._rccl.ncclCommDestroy(C.c_void_p(self.comm))
            self.comm = None


class _ShardedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, pi, evaluate, inv_n, group, already_reduced=False):
        loss, dQ = evaluate(Q.detach(), pi.detach())  # unnormalised partial sums
        packed = torch.cat([loss.reshape(-1), dQ.reshape(-1)]).to(torch.float64)
        # after enable_in_library_allreduce() the bank's own entry points return job-wide sums
        # (cb_loss_grad all-reduces on its stream): reducing again would multiply by the world size
        if not already_reduced and dist.is_available() and dist.is_initialized():
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        packed = packed * inv_n
        nl = loss.numel()
        ctx.save_for_backward(packed[nl:].reshape(Q.shape))
        return packed[:nl].reshape(loss.shape).to(Q.dtype)

    @staticmethod
    def backward(ctx, grad_loss):
        (dQ,) = ctx.saved_tensors
        return (dQ * grad_loss.reshape(-1, 1, 1)).reshape(dQ.shape), None, None, None, None, None


class ShardedBank:
    """A bank whose buckets are spread over the ranks of a process group.

    `make_bank(t_local, C_local)` builds the local evaluator (a CherryBank on
    this rank's GPU; the CPU tests inject an oracle-backed stand-in with the
    same `loss_grad_torch` method)."""

    def __init__(self, t, C, make_bank: Optional[Callable] = None, group=None, dtype: str = "f64", emulate=None):
        """`emulate = (rank, world)`: deal the buckets as that rank of a `world`-rank job WITHOUT those peers (the
        collectives run over the real group, alone): ONE rank's share of an N-rank job on the one GPU there is
        (`bench.py --shard-of N`).  The loss normaliser is then the share's own count -- a self-consistent smaller
        problem, not a partial sum of the whole bank."""
        t = np.asarray(t, dtype=np.float64).reshape(-1)
        C = np.asarray(C, dtype=np.float64)
        if C.ndim != 3:
            raise ValueError("ShardedBank shards the buckets of a single (L = 1) bank: C must be [B,S,S]")
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        # only non-empty buckets cost anything (cb_create drops the others), so deal those:
        # on real co-evolution banks 2/3 of the buckets are empty and dealing all of them
        # round-robin would leave the ranks unbalanced
        live = np.flatnonzero(np.any(C.reshape(C.shape[0], -1) != 0.0, axis=1))
        if live.size == 0:
            raise ValueError("the bank has no counts")
        if live.size < self.world:   # known to every rank alike: all raise, none is left waiting in a collective
            raise ValueError(f"{live.size} non-empty buckets < world={self.world}: some rank would own no bucket")
        deal_rank, deal_world = (self.rank, self.world) if emulate is None else (int(emulate[0]), int(emulate[1]))
        if not 0 <= deal_rank < deal_world or live.size < deal_world:
            raise ValueError(f"emulate={emulate}: needs 0 <= rank < world <= {live.size} non-empty buckets")
        mine = live[bucket_shard(live.size, deal_rank, deal_world)]
        self.emulate = None if emulate is None else (deal_rank, deal_world)
        # every rank sees the full host array here (an emulated share normalises by its own count)
        self.total_count = float(C.sum()) if emulate is None else float(C[mine].sum())
        self.local_buckets = mine
        self.bank = self._make(make_bank, dtype)(t[mine], C[mine])

    @staticmethod
    def _make(make_bank, dtype="f64"):
        if make_bank is not None:
            return make_bank
        from .bank import CherryBank
        dev = torch.cuda.current_device()
        return lambda tt, CC: CherryBank(tt, CC, device=dev, dtype=dtype)

    @classmethod
    def from_rank_counts(cls, t, C_rank, make_bank: Optional[Callable] = None, group=None, dtype: str = "f64"):
        """`C_rank` [B,S,S] (numpy, or a torch tensor already on this rank's GPU): the counts of
        the families THIS rank counted.  The k-th globally non-empty bucket belongs to rank
        k mod world; those are laid out owner-major (zero padded to equal chunks) and reduce-scattered, so rank r receives
        sum_over_ranks C[b] for its own buckets only.  The grand total (the loss normaliser) is
        one scalar all-reduce."""
        self = cls.__new__(cls)
        t = np.asarray(t, dtype=np.float64).reshape(-1)
        on = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        Ct = torch.as_tensor(C_rank, dtype=torch.float64)
        if Ct.ndim != 3 or Ct.shape[0] != t.size:
            raise ValueError("from_rank_counts: C_rank must be [B,S,S] with B = len(t)")
        B, S = Ct.shape[0], Ct.shape[1]
        # Only non-empty buckets cost anything (cb_create drops the others) and a rank left with empty
        # buckets only could not even build its bank: deal the GLOBALLY non-empty ones.  Their masses
        # (B doubles) and the grand total (the loss normaliser) travel in one small all-reduce.
        stats = torch.cat([Ct.abs().reshape(B, -1).sum(dim=1), Ct.sum().reshape(1)])
        if on:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        stats = stats.cpu().numpy()
        live = np.flatnonzero(stats[:B] != 0.0)
        if live.size == 0:
            raise ValueError("the bank has no counts")
        if live.size < self.world:   # the same numbers on every rank: all raise together
            raise ValueError(f"{live.size} non-empty buckets < world={self.world}: some rank would own no bucket")
        chunk = -(-live.size // self.world)
        owner_major = [live[bucket_shard(live.size, r, self.world)] for r in range(self.world)]
        mine = owner_major[self.rank]
        if on:
            packed = torch.zeros((self.world * chunk, S, S), dtype=torch.float64, device=Ct.device)
            for r, idx in enumerate(owner_major):
                packed[r * chunk:r * chunk + idx.size] = Ct[torch.as_tensor(idx, device=Ct.device)]
            if dist.get_backend(group) == "gloo":   # CPU tests: gloo has no reduce-scatter
                dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
                got = packed[self.rank * chunk:(self.rank + 1) * chunk].clone()
            else:
                got = torch.empty((chunk, S, S), dtype=torch.float64, device=Ct.device)
                dist.reduce_scatter_tensor(got, packed, op=dist.ReduceOp.SUM, group=group)
            del packed
            C_mine = got[:mine.size]
        else:
            C_mine = Ct[torch.as_tensor(mine, device=Ct.device)]
        self.total_count = float(stats[B])
        self.local_buckets = mine
        self.bank = cls._make(make_bank, dtype)(t[mine], C_mine if C_mine.is_cuda else C_mine.numpy())
        return self

    def loss(self, Q: torch.Tensor, pi: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """Differentiable (in Q) loss of the WHOLE bank; every rank gets the same value."""
        ev = lambda q, p: self.bank.loss_grad_torch(q, p, normalize=False, want_grad=True)  # noqa: E731
        inv_n = 1.0 / self.total_count if normalize else 1.0
        return _ShardedLoss.apply(Q, pi, ev, inv_n, self.group, getattr(self, "rccl", None) is not None)

    def enable_in_library_allreduce(self):
        """Hand a raw RCCL communicator to the bank (`cb_allreduce_setup`): from then on the bank's own
        entry points return job-wide sums, and `train_pande_reversible` runs the WHOLE sharded epoch
        loop from C -- theta -> A, replicated eigensolve, this rank's buckets, one ncclAllReduce of
        (loss, dL/dA) on the handle's stream, identical Adam steps -- with no torch in the loop.
        Collective: every rank of the group must call it."""
        error, rccl = None, None
        try:
            rccl = RcclCommunicator(self.group)
        except Exception as exc:   # exchanged below: a rank failing alone would mix torch and RCCL collectives
            error = f"{type(exc).__name__}: {exc}"
        if self.world > 1:
            errs = [None] * self.world
            dist.all_gather_object(errs, error, group=self.group)
            error = next((f"rank {r}: {e}" for r, e in enumerate(errs) if e is not None), None)
        if error is not None:
            if rccl is not None:
                rccl.destroy()
            raise RuntimeError(f"enable_in_library_allreduce: no raw RCCL communicator ({error})")
        self.rccl = rccl
        self.bank.allreduce_setup(self.rccl.comm, self.rccl.allreduce_fn, [self.total_count])
        return self

    def train_pande_reversible(self, upper_diag, log_pi, mask=None, num_epochs=2000, lr=0.1, do_adam=True,
                               normalize=True, resume=False):
        """The reference loop (trainer.py:156-218) over the sharded bank, driven from C on every rank
        (needs `enable_in_library_allreduce()`); every rank returns the same result."""
        if getattr(self, "rccl", None) is None:
            raise RuntimeError("ShardedBank.train_pande_reversible: call enable_in_library_allreduce() first")
        return self.bank.train_pande_reversible(upper_diag, log_pi, mask=mask, num_epochs=num_epochs, lr=lr,
                                                do_adam=do_adam, normalize=normalize, resume=resume)

    def close(self):
        if hasattr(self.bank, "close"):
            self.bank.close()
        if getattr(self, "rccl", None) is not None:
            self.rccl.destroy()
            self.rccl = None
