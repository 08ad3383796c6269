"""world_size-2 gloo test of the bucket-sharded bank (cherryml_amd/distributed.py).
The local evaluator is an oracle-backed stand-in with the CherryBank interface
(tests may use the oracle as the checker); what is under test is the sharding,
the all-reduce and the autograd plumbing."""
import os

import pytest
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, relerr


class OracleBank:
    """CherryBank look-alike on the CPU: unnormalised loss and dL/dQ of its buckets."""

    def __init__(self, t, C):
        self.t = torch.tensor(np.asarray(t))
        self.C = torch.tensor(np.asarray(C))

    def loss_grad_torch(self, Q, pi, normalize=False, want_grad=True):
        from oracle import ratelearn_oracle as orc
        with torch.enable_grad():  # called from inside an autograd.Function.forward
            q = Q.detach().reshape(Q.shape[-2:]).clone().requires_grad_(True)
            loss = orc.bank_loss(q, self.t, self.C, normalize=normalize)
            loss.backward()
        return loss.detach().reshape(1), q.grad.reshape(1, *q.shape)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cherryml_amd.distributed import ShardedBank, bucket_shard
    from cherryml_amd.estimation._ratelearn._rate_matrix import RateMatrix
    g = load_golden("traj_lgbank.npz")
    t, C = g["t"][::8], g["C"][::8]  # 17 buckets, some of them empty
    sb = ShardedBank(t, C, make_bank=lambda tt, CC: OracleBank(tt, CC))
    live = np.flatnonzero(C.reshape(len(t), -1).any(axis=1))   # only non-empty buckets are dealt
    assert list(sb.local_buckets) == list(live[bucket_shard(len(live), rank, world)])
    torch.manual_seed(0)
    mod = RateMatrix(num_states=20, mode="pande_reversible", mask=torch.ones(20, 20),
                     pi=torch.ones(20, dtype=torch.float64) / 20, pi_requires_grad=True,
                     initialization=g["init"])
    opt = torch.optim.Adam(mod.parameters(), lr=0.1)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = sb.loss(mod(), mod.stationary(), normalize=True)[0]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.save(dict(losses=losses, Q=mod().detach()), os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_bank_two_ranks_matches_single(tmp_path):
    from oracle import ratelearn_oracle as orc
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    g = load_golden("traj_lgbank.npz")
    ref = orc.train(g["t"][::8], g["C"][::8], None, initialization=g["init"], num_epochs=4)
    # both ranks hold the same replica and it equals the unsharded optimisation
    assert r0["losses"] == r1["losses"]
    assert torch.equal(r0["Q"], r1["Q"])
    assert np.allclose(r0["losses"], ref["loss"][:3], rtol=1e-12, atol=0)
    assert relerr(r0["Q"].numpy(), ref["Q_4"]) < 1e-11


def _worker_family(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cherryml_amd.distributed import ShardedBank
    g = load_golden("traj_lgbank.npz")
    t, C = g["t"][::8], g["C"][::8]
    # family sharding: each rank counted a different part of the data (integer counts, so the
    # split is exact): rank 0 holds floor(C/3), rank 1 the rest
    part = np.floor(C / 3.0) if rank == 0 else C - np.floor(C / 3.0)
    sb = ShardedBank.from_rank_counts(t, part, make_bank=lambda tt, CC: OracleBank(tt, CC))
    assert abs(sb.total_count - float(C.sum())) <= 1e-12 * float(C.sum())
    live = np.flatnonzero(C.reshape(len(t), -1).any(axis=1))
    assert live.size < len(t)                                       # the fixture has empty buckets
    mine = live[rank::world]                                        # only globally non-empty buckets are dealt
    assert list(sb.local_buckets) == list(mine)
    assert np.array_equal(sb.bank.C.numpy(), C[mine])               # summed counts of MY buckets only
    sizes = [live[r::world].size for r in range(world)]
    assert max(sizes) - min(sizes) <= 1                             # balanced over the LIVE buckets
    Q = torch.tensor(g["init"], dtype=torch.float64, requires_grad=True)
    pi = torch.full((20,), 0.05, dtype=torch.float64)
    loss = sb.loss(Q, pi, normalize=True)[0]
    loss.backward()
    torch.save(dict(loss=loss.item(), grad=Q.grad), os.path.join(out, f"f{rank}.pt"))
    dist.destroy_process_group()


def test_family_sharded_counts_reduce_scatter_two_ranks(tmp_path):
    from oracle import ratelearn_oracle as orc
    port = _free_port()
    mp.spawn(_worker_family, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "f0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "f1.pt"))
    g = load_golden("traj_lgbank.npz")
    Q = torch.tensor(g["init"], dtype=torch.float64, requires_grad=True)
    ref = orc.bank_loss(Q, torch.tensor(g["t"][::8]), torch.tensor(g["C"][::8]))
    ref.backward()
    assert r0["loss"] == r1["loss"] and torch.equal(r0["grad"], r1["grad"])
    assert abs(r0["loss"] - ref.item()) < 1e-12 * abs(ref.item())
    assert relerr(r0["grad"].numpy(), Q.grad.numpy()) < 1e-11


def test_emulated_share_deals_the_buckets_of_one_rank_without_its_peers():
    """`ShardedBank(emulate = (rank, world))` (bench.py --shard-of N): the non-empty buckets rank `rank` of a `world`-rank job
    would own, built without a process group; the share normalises by its own count (a self-consistent smaller problem)."""
    from cherryml_amd.distributed import ShardedBank
    rng = np.random.default_rng(2)
    B, S = 23, 6
    t = np.sort(rng.uniform(0.01, 2.0, size=B))
    C = rng.poisson(3.0, size=(B, S, S)).astype(np.float64)
    C[[1, 7, 8, 20]] = 0.0                                  # empty buckets are never dealt
    live = np.flatnonzero(C.reshape(B, -1).any(axis=1))
    seen = []
    for r in range(4):
        sb = ShardedBank(t, C, make_bank=lambda tt, CC: OracleBank(tt, CC), emulate=(r, 4))
        assert sb.world == 1 and sb.emulate == (r, 4)
        assert np.array_equal(sb.local_buckets, live[r::4])
        assert sb.total_count == C[sb.local_buckets].sum()
        seen.extend(sb.local_buckets.tolist())
    assert sorted(seen) == live.tolist()                    # the four shares partition the live buckets
    with pytest.raises(ValueError):
        ShardedBank(t, C, make_bank=lambda tt, CC: OracleBank(tt, CC), emulate=(4, 4))
    with pytest.raises(ValueError):
        ShardedBank(t, C, make_bank=lambda tt, CC: OracleBank(tt, CC), emulate=(0, 64))


def test_bucket_shard_partition():
    from cherryml_amd.distributed import bucket_shard
    for world in (1, 2, 3, 8):
        parts = [bucket_shard(129, r, world) for r in range(world)]
        assert sorted(np.concatenate(parts)) == list(range(129))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_nccl_unique_id_travels_whole():
    """The id handed to the other ranks must be all 128 bytes, NULs included."""
    from cherryml_amd.distributed import _NcclUniqueId, _uid_from_bytes, _uid_to_bytes
    raw = bytes([7, 0, 0, 9] + [0] * 60 + list(range(64)))
    uid = _uid_from_bytes(raw)
    assert _uid_to_bytes(uid) == raw and len(raw) == 128
    assert bytes(uid.internal) != raw          # the tempting spelling truncates at the first NUL
    with pytest.raises(ValueError):
        _uid_from_bytes(raw[:100])
    assert isinstance(uid, _NcclUniqueId)


def _rccl_worker(rank, world, port, lib, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CHERRYML_AMD_RCCL_LIB=lib, FAKE_RCCL_OUT=outdir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cherryml_amd.distributed import RcclCommunicator
        comm = RcclCommunicator()
        assert comm.rank == rank and comm.world == world and comm.comm and comm.allreduce_fn
        comm.destroy()
        assert comm.comm is None
    finally:
        dist.destroy_process_group()


def test_rccl_communicator_control_flow_two_ranks(tmp_path):
    """RcclCommunicator on two gloo ranks with a test double for librccl (tests/fixtures/fake_rccl.c, built
    here with gcc): rank 0 makes the unique id, every rank's ncclCommInitRank gets all 128 bytes of it
    (embedded NULs included), its own rank and the world size.  The real thing (RCCL over xGMI) needs GPUs."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from conftest import ROOT
    lib = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", lib, os.path.join(ROOT, "tests", "fixtures", "fake_rccl.c")])
    out = tmp_path / "out"
    out.mkdir()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(2, port, lib, str(out)), nprocs=2, join=True)
    want = bytes([42, 0, 0, 7] + [((i * 37 + 11) & 0xff) for i in range(4, 128)])
    want = want[:64] + b"\0" + want[65:]
    for rank in range(2):
        raw = (out / f"rank{rank}.bin").read_bytes()
        assert int.from_bytes(raw[0:4], "little") == 2 and int.from_bytes(raw[4:8], "little") == rank
        assert raw[8:] == want


class LibraryReducingBank(OracleBank):
    """Stand-in for a CherryBank after cb_allreduce_setup: its own entry point returns job-wide sums."""
    reduced = False

    def allreduce_setup(self, comm, fn, n_total):
        assert comm and fn and len(n_total) == 1
        self.reduced = True

    def loss_grad_torch(self, Q, pi, normalize=False, want_grad=True):
        loss, dQ = super().loss_grad_torch(Q, pi, normalize=normalize, want_grad=want_grad)
        if self.reduced:
            dist.all_reduce(loss)
            dist.all_reduce(dQ)
        return loss, dQ


def _inlib_worker(rank, world, port, lib, outdir, break_rank):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAKE_RCCL_OUT=outdir,
                      CHERRYML_AMD_RCCL_LIB=(lib + ".missing") if rank == break_rank else lib)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cherryml_amd.distributed import ShardedBank
        g = load_golden("traj_lgbank.npz")
        t, C = g["t"][::8], g["C"][::8]
        sb = ShardedBank(t, C, make_bank=lambda tt, CC: LibraryReducingBank(tt, CC))
        res = {}
        try:
            sb.enable_in_library_allreduce()
            Q = torch.tensor(g["init"], dtype=torch.float64, requires_grad=True)
            loss = sb.loss(Q, torch.full((20,), 0.05, dtype=torch.float64), normalize=True)[0]
            loss.backward()
            res = dict(loss=loss.item(), grad=Q.grad)
        except RuntimeError as exc:
            res = dict(error=str(exc))
        torch.save(res, os.path.join(outdir, f"i{rank}.pt"))
        sb.close()
    finally:
        dist.destroy_process_group()


def _fake_rccl(tmp_path):
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from conftest import ROOT
    lib = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", lib, os.path.join(ROOT, "tests", "fixtures", "fake_rccl.c")])
    return lib


def test_in_library_allreduce_is_not_reduced_twice(tmp_path):
    """after enable_in_library_allreduce() the bank itself returns job-wide sums, so
    ShardedBank.loss must not all-reduce them again (the loss came out world times too large)."""
    from oracle import ratelearn_oracle as orc
    lib = _fake_rccl(tmp_path)
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_inlib_worker, args=(2, _free_port(), lib, str(out), -1), nprocs=2, join=True)
    r0, r1 = torch.load(out / "i0.pt"), torch.load(out / "i1.pt")
    g = load_golden("traj_lgbank.npz")
    Q = torch.tensor(g["init"], dtype=torch.float64, requires_grad=True)
    ref = orc.bank_loss(Q, torch.tensor(g["t"][::8]), torch.tensor(g["C"][::8]))
    ref.backward()
    assert r0["loss"] == r1["loss"]
    assert abs(r0["loss"] - ref.item()) < 1e-12 * abs(ref.item())
    assert relerr(r0["grad"].numpy(), Q.grad.numpy()) < 1e-11


def test_communicator_failure_on_one_rank_raises_on_all(tmp_path):
    """A rank that cannot make its RCCL communicator must not strand its peers in ncclCommInitRank or
    leave the job mixing torch and in-library collectives: every rank raises."""
    lib = _fake_rccl(tmp_path)
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_inlib_worker, args=(2, _free_port(), lib, str(out), 1), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(out / f"i{r}.pt")
        assert "error" in res and "rank 1" in res["error"], res


def test_too_few_live_buckets_raises_on_every_rank():
    from cherryml_amd.distributed import ShardedBank
    C = np.zeros((4, 3, 3))
    with pytest.raises(ValueError, match="no counts"):
        ShardedBank(np.arange(1.0, 5.0), C, make_bank=lambda tt, CC: OracleBank(tt, CC))
    with pytest.raises(ValueError, match="no counts"):
        ShardedBank.from_rank_counts(np.arange(1.0, 5.0), C, make_bank=lambda tt, CC: OracleBank(tt, CC))


def _cache_worker(rank, world, port, out, cache_dir, mode):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cherryml_amd import caching
    caching.set_cache_dir(cache_dir)

    @caching.cached_computation(output_dirs=["output_dir"])
    def stage(x: int, output_dir=None):
        if mode == "body":
            raise ValueError("body broke")
        with open(os.path.join(output_dir, "result.txt"), "w") as f:
            f.write(str(x))

    res = {}
    try:
        res["dirs"] = stage(x=3)
    except Exception as exc:
        res["error"] = f"{type(exc).__name__}: {exc}"
    torch.save(res, os.path.join(out, f"c{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ok", "prepare", "body"])
def test_cached_stage_failures_on_rank_0_raise_on_every_rank(tmp_path, mode):
    """Rank 0's filesystem work of the cache decorator (directory creation, tokens) sits inside the failure
    exchange: when it raises (here: the cache root is a FILE), rank 1 raises too instead of waiting in the next
    broadcast for ever."""
    out = tmp_path / "out"
    out.mkdir()
    cache = tmp_path / "cache"
    if mode == "prepare":
        cache.write_text("not a directory")
    mp.spawn(_cache_worker, args=(2, _free_port(), str(out), str(cache), mode), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(out / f"c{r}.pt")
        if mode == "ok":
            assert os.path.exists(os.path.join(res["dirs"]["output_dir"], "result.success"))
        else:
            assert "error" in res and "rank 0" in res["error"], res
