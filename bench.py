#!/usr/bin/env python3
"""Benchmark of the composite-likelihood hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one optimiser epoch of the reference's `train_quantization`
(cherryml/estimation/_ratelearn/trainer.py:156-218) over one synthetic bank:
    Q(theta) -> loss = -sum_b <C_b, log expm(t_b Q)>/n -> dL/dtheta -> Adam step
with the bank resident in HBM.  metric = cherry-pairs/s = sum(C) / t_epoch
(every counted cherry x site pair adds exactly 1 to C).

Workloads (BASELINE.json configs):
  coevo400  co-evolution 400x400, B = 129 dense bank, sum C = 1,057,194 (the size of
            the reference's demo_data co-evolution bank)   [default]
  coevo400_demo  the reference's REAL demo_data co-evolution bank (all 32 families, counted by the
            reference: 43 of 129 buckets non-empty, 3.5 % dense); roofline denominators use the
            non-empty buckets
  lg20      LG 20x20, 1000 families x 200 sites x 64 cherries: sum C = 1.28e7
  siterm    SiteRM per-site 20x20, L sites (--sites, default 5000), B = 129
  counting  the stage that PRODUCES the LG bank: 1000 families x 64 cherries
            x 200 sites, step = one histogram pass over all cherries, inputs resident in HBM

  co_counting  the stage that PRODUCES the co-evolution bank (BASELINE config 5): 10,000 families x 64
            cherries x ~65 contacting site pairs per GPU into the [129][400][400] histogram

N > 1: `python bench.py --gpus N` started bare launches its own N rank processes (one per GPU,
torch.distributed / RCCL; under torch.distributed.run it is one of the ranks); `n_gpus` is the process
group's rank count and a mismatch with --gpus is an error.  coevo400 shards the buckets of ONE bank over
the ranks and all-reduces (loss, dL/dA) each epoch -- STRONG scaling, `value` = the bank's pairs / epoch
time, capped by the replicated eigensolver (the line's `amdahl` object); lg20 does not shard (replicas
only).  The workloads that shard without a replicated term ride along as `secondary_siterm` (sites x N,
no collective) and `secondary_co_counting` (families x N + the all-reduce of the integer counts): WEAK
scaling, each with its own `value`, `scaling`, `roofline`.
"""
import argparse
import json
import os
import sys
import time


def cpu_budget() -> int:
    """CPUs this process may really use: the affinity mask capped by the container's CFS quota (cgroup v2 cpu.max, v1
    cpu.cfs_quota_us).  The GPU boxes report 256 CPUs under a quota of 16: numpy / torch then start 256 threads, the cgroup
    is throttled (cpu.stat: ~35 throttled periods and 60 s of throttled thread time per bench run) and the throttled
    periods land on the thread that feeds the GPU -- 70-95 ms with nothing enqueued, anywhere in the first epochs, so that
    one run in five read 4.5 ms per epoch instead of 1.2 (found with CB_TRACE_SLOW)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path_q, path_p in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if path_p is None:
                q, p = open(path_q).read().split()[:2]
            else:
                q, p = open(path_q).read().strip(), open(path_p).read().strip()
            if q not in ("max", "-1") and int(p) > 0:
                n = min(n, max(1, int(q) // int(p)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


# host threads: the budget shared by the ranks of this job, a quarter of each rank's share (at least one core) left to the thread
# that feeds the GPU and the runtime's own threads (with one core left, 3 throttled periods in six driver commands and one of
# them 9 % slower); idle OpenMP workers sleep instead of spinning (set BEFORE numpy / torch load their thread pools)
CPU_BUDGET = cpu_budget()
_share = CPU_BUDGET // max(1, int(os.environ.get("WORLD_SIZE", "1")))
HOST_THREADS = max(1, _share - max(1, _share // 4))
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, str(HOST_THREADS))
os.environ.setdefault("KMP_BLOCKTIME", "0")
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np   # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_PEAK_TFLOPS = 78.6     # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (f64 MFMA = f64 vector rate)
PHASE_EVENTS_EVERY = 4   # the 400-state trainer's phase events (phase_ms, roofline) are recorded in every 4th timed epoch: six completion
                         # events cost an epoch 0.015 ms (profiles/r06_phase_event_cost.json); averages are over the recorded epochs
PREWARM_EPOCHS = 30   # throw-away epochs in front of the W warm-up epochs of the 400-state trainer (see main)
F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, 64 flop/clk/SIMD (exact f32)
LDS_ATOMIC_PEAK_G = 9830.4  # G lane-atomics/s: 16 lanes per clock per CU (LDS table: a 4-byte LDS write / atomic = 4 cycles per
                            # wave-instruction) x 256 CUs x 2.4 GHz, conflict-free
L2_GATHER_PEAK_GBS = 16800.0  # MI355X_MICROARCH.md, "Indexed rows": rows gathered from a table shared in the XCD's L2, chip-wide


# --------------------------------------------------------------------- inputs
def quantization_grid():
    """reference estimation_end_to_end/_cherry.py:267-272"""
    return np.array([float("%.8f" % (0.03 * 1.1 ** i)) for i in range(-64, 65)])


def bucket_weights(n_pairs: float, rng) -> np.ndarray:
    """cherry lengths ~ Exp(mean 0.4), snapped to the nearest grid point in log space"""
    grid = quantization_grid()
    lengths = rng.exponential(0.4, size=400000)
    lengths = lengths[(lengths >= grid[0]) & (lengths <= grid[-1])]
    idx = np.abs(np.log(lengths[:, None] / grid[None, :])).argmin(1)
    w = np.bincount(idx, minlength=grid.size).astype(np.float64)
    w = np.maximum(w, 1.0)  # dense bank: every bucket populated
    return w * (n_pairs / w.sum())


def lg_matrix():
    z = np.load(os.path.join(ROOT, "tests", "golden", "data_lg.npz"))
    return z["lg"]


def stationary(Q):
    w, v = np.linalg.eig(Q.T)
    p = v[:, int(np.argmin(np.abs(w.real)))].real
    return p / p.sum()


def expm_entrywise(Q, t):
    """expm(t Q) of a rate matrix with ENTRYWISE relative accuracy: uniformisation,
    expm(tQ) = e^{-mu t} expm(t (Q + mu I)) with Q + mu I >= 0, so the Taylor series and the squarings
    add non-negative numbers only -- no cancellation, whatever order a BLAS sums in.  The synthetic banks
    hold expected counts pi_i P_ij(t) down to 1e-14 (double substitutions at the shortest branch lengths)
    and the loss divides by those P_ij: an eigendecomposition-based expm leaves them at noise level, i.e.
    different on every machine; this one reproduces them to ~1e-13 relative on any LAPACK / thread count."""
    n = Q.shape[0]
    mu = float(np.max(-np.diag(Q)))
    s = max(0, int(np.ceil(np.log2(max(t * mu, 1e-300)))))
    h = t / 2.0 ** s
    X = h * (Q + mu * np.eye(n))
    term = np.eye(n)
    acc = np.eye(n)
    for k in range(1, 25):           # ||X|| <= 1: 1/24! ~ 1e-24
        term = term @ X / k
        acc += term
    P = np.exp(-h * mu) * acc
    for _ in range(s):
        P = P @ P
    return P


def reversible_bank(Q, pi, n_pairs, rng):
    """C_b = w_b diag(pi) expm(t_b Q), symmetrised (cherries are unordered)."""
    grid = quantization_grid()
    w = bucket_weights(n_pairs, rng)
    C = np.empty((grid.size,) + Q.shape)
    for b, t in enumerate(grid):
        J = pi[:, None] * expm_entrywise(Q, t)
        C[b] = w[b] * 0.5 * (J + J.T)
    return grid, C * (n_pairs / C.sum())


def coevolution_truth(rng):
    """400-state pair chain: two LG sites plus a symmetric log-normal coupling on
    the allowed (single-substitution) rates; mask as data/mask_matrices/aa_coevolution_mask."""
    lg = lg_matrix()
    pi1 = stationary(lg)
    I = np.eye(20)
    Q = np.kron(lg, I) + np.kron(I, lg)
    a = np.arange(400) // 20
    b = np.arange(400) % 20
    mask = ((a[:, None] == a[None, :]) | (b[:, None] == b[None, :])).astype(np.float64)
    pi = np.kron(pi1, pi1)
    d = np.sqrt(pi)
    R = d[:, None] * Q / d[None, :]
    noise = rng.normal(0.0, 0.3, size=(400, 400))
    noise = np.triu(noise, 1)
    R = R * np.exp(noise + noise.T)
    np.fill_diagonal(R, 0.0)
    R = R * mask
    Qc = R * d[None, :] / d[:, None]
    Qc -= np.diag(Qc.sum(1))
    return Qc, pi, mask


def make_workload(name, sites, rng):
    if name == "coevo400":
        Q, pi, mask = coevolution_truth(rng)
        n_pairs = 1057194.0
        t, C = reversible_bank(Q, pi, n_pairs, rng)
        return dict(kind="single", t=t, C=C, mask=mask, S=400, n_pairs=n_pairs,
                    desc="co-evolution 400x400, B=129 dense synthetic bank, sum C = 1,057,194 "
                         "(= reference demo_data co-evolution bank)")
    if name == "coevo400_demo":
        # BASELINE.json config 3 as it really is: the co-evolution bank of ALL 32 demo_data families,
        # counted by the reference itself (tests/golden/coevo_demo_full.npz, made by
        # tests/golden/make_golden_s400_full.py; stored sparse): 43 of 129 buckets non-empty, 3.5 % dense
        z = np.load(os.path.join(ROOT, "tests", "golden", "coevo_demo_full.npz"))
        C = np.zeros(tuple(z["C_shape"]))
        C[z["C_b"].astype(np.int64), z["C_i"].astype(np.int64), z["C_j"].astype(np.int64)] = z["C_quarters"] * 0.25
        mask = np.unpackbits(z["mask_packed"])[:160000].reshape(400, 400).astype(np.float64)
        live = int(np.count_nonzero(C.reshape(C.shape[0], -1).any(axis=1)))
        return dict(kind="single", t=z["t"], C=C, mask=mask, S=400, n_pairs=float(C.sum()), live=live,
                    desc=f"co-evolution 400x400, the reference's demo_data bank (32 families): B=129 of which {live} non-empty, "
                         f"{100.0 * np.count_nonzero(C) / C.size:.1f} % dense, sum C = {int(C.sum()):,}")
    if name == "lg20":
        Q = lg_matrix()
        n_pairs = 1000 * 200 * 64.0
        t, C = reversible_bank(Q, stationary(Q), n_pairs, rng)
        return dict(kind="single", t=t, C=C, mask=np.ones((20, 20)), S=20, n_pairs=n_pairs,
                    desc="LG 20x20, 1000 families x 200 sites x 64 cherries, B=129")
    if name == "siterm":
        lg = lg_matrix()
        pi = stationary(lg)
        grid = quantization_grid()
        equ = (np.ones((20, 20)) - np.eye(20)) / 19.0
        equ -= np.diag(equ.sum(1))
        banks = []
        for Qb, pb in ((lg, pi), (equ, np.full(20, 0.05))):
            banks.append(reversible_bank(Qb, pb, 2000.0, rng)[1])
        rates = rng.gamma(3.0, 1.0 / 3.0, size=sites)
        C = np.empty((sites, grid.size, 20, 20))
        T = np.empty((sites, grid.size))
        Q0 = np.empty((sites, 20, 20))
        for l in range(sites):
            C[l] = banks[l % 2]
            T[l] = grid / max(rates[l], 0.05)  # per-site grid (the reference's rate-scaled buckets)
            Q0[l] = (lg if l % 2 == 0 else equ) * 0.9
        return dict(kind="sites", t=T, C=C, S=20, n_pairs=float(C.sum()), init=Q0,
                    desc=f"SiteRM per-site 20x20, L={sites} sites, B=129")
    raise SystemExit(f"unknown workload {name}")


# ----------------------------------------------------------------- cpu baseline
def cpu_baseline(wl, name):
    """The oracle (reference algorithm: float32 torch.matrix_exp + autograd on the
    host cores) on a bounded sample of the same bank; scaled to one whole epoch."""
    import torch
    from oracle import ratelearn_oracle as orc
    from cherryml_amd.estimation import jtt_ipw_from_arrays

    cores = torch.get_num_threads()
    if wl["kind"] == "single":
        B = wl["C"].shape[0]
        nb = {400: 32, 20: 129}[wl["S"]]   # ~10-20 s of host work either way
        if "live" in wl:   # (the reference exponentiates all 129 buckets, empty or not: trainer.py:170)
            nb = min(nb, B)
        sel = np.linspace(0, B - 1, nb).round().astype(int)
        t, C = wl["t"][sel], wl["C"][sel]
        init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
        u, p = orc.invert_pande_reversible(init, wl["mask"])
        # >= 3 timed evaluations each way; the MEDIAN is the baseline and the spread is reported (a single evaluation of
        # the 400-state sample varied by 70 % between two runs of the same command in round 2)
        reps, inner = (7, 1) if wl["S"] == 400 else (5, 40)   # (400 states: ~0.6 s per evaluation of the 32-bucket sample on 15 threads)
        orc.evaluate(u, p, wl["mask"], t[:1], C[:1], torch.float32)  # warm up
        samples = []
        for _ in range(reps):
            t0 = time.time()
            for _ in range(inner):
                orc.evaluate(u, p, wl["mask"], t, C, torch.float32)
            samples.append((time.time() - t0) / inner * (B / nb))
        dt = float(np.median(samples))
        spread = dict(evaluations=reps * inner, seconds_per_epoch_min=min(samples), seconds_per_epoch_max=max(samples))
        sample = (f"{nb} of {B} buckets x {reps} timed evaluation(s)" + (f" of {inner} each" if inner > 1 else "") +
                  f", float32 expm as the reference, scaled to {B}; value = median")
    else:
        L = min(16, wl["C"].shape[0])
        samples = []
        for _ in range(3):
            Q = torch.tensor(wl["init"][:L], requires_grad=True)
            t0 = time.time()
            _, tot = orc.siterm_loss(Q, torch.tensor(wl["C"][:L]), torch.tensor(wl["t"][:L]))
            tot.backward()
            samples.append((time.time() - t0) * (wl["C"].shape[0] / L))
        dt = float(np.median(samples))
        spread = dict(evaluations=3, seconds_per_epoch_min=min(samples), seconds_per_epoch_max=max(samples))
        sample = f"{L} of {wl['C'].shape[0]} sites x 3 timed evaluations, float64 as the reference's SiteRM path, scaled; value = median"
    return dict(value=wl["n_pairs"] / dt, unit="cherry-pairs/s", cores=int(cores), kind="port",
                sample=sample, seconds_per_epoch=dt, **spread)


# -------------------------------------------------------------------- launcher
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, child=None, grace=15.0, poll=0.05, _attempt=0):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment): THIS process touches no GPU (it
    imports numpy only), starts N fresh rank processes of this script -- one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment, as `torch.distributed.run` would set them (the reference starts its
    own ranks from inside the stage the same way: counting/_count_transitions.py:295-316, `mpirun -np N`) --, relays
    rank 0's standard output (whose last line is the JSON line) and returns non-zero when any rank does.  A rank that
    fails takes the others down after `grace` seconds (they may be waiting for it in a collective).  Never an exec:
    the children are ordinary child processes.  `child` replaces the command (tests use a stub)."""
    import subprocess
    cmd = list(child) if child else [sys.executable, os.path.abspath(__file__)]
    # (the port is free NOW; another process can still take it before rank 0 binds it -- a job that dies within seconds of its
    # start is therefore started once more on another port: `_attempt`)
    t_start = time.time()
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CB_BENCH_LAUNCHER="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL across processes)
        for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):   # the ranks SHARE the CPU budget
            env[k] = str(max(1, CPU_BUDGET // n - max(1, CPU_BUDGET // n // 4)))
        # rank 0's stdout is the result; the other ranks' stdout joins the launcher's stderr
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    failed_at, first_bad = None, None
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
                if codes[r] not in (None, 0) and failed_at is None:
                    failed_at, first_bad = time.time(), codes[r]
                    print(f"bench launcher: rank {r} exited with code {codes[r]}", file=sys.stderr)
        if failed_at is not None and time.time() - failed_at > grace:
            for r, pr in enumerate(procs):   # exactly the processes started above
                if codes[r] is None:
                    pr.kill()
                    codes[r] = pr.wait()
                    print(f"bench launcher: rank {r} killed after rank failure", file=sys.stderr)
        time.sleep(poll)
    reader.join(timeout=5.0)
    out = (chunks[0] if chunks else b"").decode(errors="replace")
    if first_bad is not None:
        if _attempt == 0 and child is None and failed_at - t_start < 8.0:   # died at the rendezvous: once more, new port
            print("bench launcher: the job failed within seconds of its start; retrying once on another port", file=sys.stderr)
            return launch_ranks(n, argv, child, grace, poll, _attempt=1)
        sys.stderr.write(out)   # no result line on stdout from a failed job
        return first_bad if 0 < first_bad < 256 else 1
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks of ONE node (default: WORLD_SIZE when started under torch.distributed.run, else 1)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="coevo400",
                    choices=["coevo400", "coevo400_demo", "lg20", "siterm", "counting", "co_counting", "ble", "assembly",
                             "likelihood"])
    ap.add_argument("--sites", type=int, default=5000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "mixed"],
                    help="element type of the bank products (coevo400 only): f32 = float32 MFMA (cb_create dtype CB_F32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (ShardedBank + collectives) even at N = 1")
    ap.add_argument("--shard-of", type=int, default=0, metavar="N",
                    help="coevo400 / coevo400_demo on ONE GPU: rank 0's share of an N-rank job -- its buckets of the N-way deal "
                         "of the non-empty buckets + the replicated eigensolve, the C-driven loop with the in-library "
                         "all-reduce (a single-rank communicator) -- i.e. what one GPU of N does per epoch, measured")
    ap.add_argument("--torch-glue", action="store_true",
                    help="multi-GPU co-evolution: keep theta->Q / Adam in torch and the collective in "
                         "torch.distributed instead of the C-driven loop with the in-library ncclAllReduce")
    args = ap.parse_args()

    if args.gpus is None:   # left at its default: adopt the launcher's rank count (an EXPLICIT mismatch is still refused below)
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # started bare: be the launcher (no torch import, no GPU touched in this process)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    def refuse(found, what):
        # a line that says n_gpus = N must have run on N ranks: refuse (on every rank) instead of warning
        print(f"bench.py: --gpus {args.gpus} but {what} has {found} rank(s); start it bare (`python bench.py --gpus N` "
              "launches its own ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus", file=sys.stderr)
        sys.exit(2)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        refuse(world, "the environment (WORLD_SIZE)")

    import torch
    import torch.distributed as dist

    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # CB_BENCH_BACKEND=gloo is a TEST HOOK (never a measurement): with it several ranks may share one GPU (RCCL refuses
        # that), which is how the N > 1 control flow -- launcher, sharding, collectives on device tensors, the fall-back
        # when no raw RCCL communicator can be made, the JSON relay -- runs on a one-GPU box; the line says so in `data`
        backend = os.environ.get("CB_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:   # n_gpus in the line = the communicator's rank count
            n_found = dist.get_world_size()
            dist.destroy_process_group()
            refuse(n_found, "the process group")
        world = dist.get_world_size()
    if os.environ.get("CB_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)   # (test hook: ranks may share a device)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(workload, steps, warmup, with_cpu, shard_of=0):
        import cherryml_amd
        from cherryml_amd.distributed import ShardedBank
        from cherryml_amd.estimation import jtt_ipw_from_arrays

        rng = np.random.default_rng(0)
        wl = make_workload(workload, args.sites, rng)
        S = wl["S"]
        bank_dtype = args.dtype if S > 32 else "f64"   # the small-state kernels are float64
        resumed = False   # the timed epochs continue the warm-up's optimisation (C-driven 400-state loop)
        if shard_of and not (wl["kind"] == "single" and S > 32 and world == 1):
            raise SystemExit("--shard-of: one 400-state bank on one GPU (coevo400 / coevo400_demo, --gpus 1)")
        if wl["kind"] == "single" and S > 32 and (world > 1 or args.force_sharded or shard_of):
            # ---- co-evolution on N > 1 GPUs: torch keeps theta -> Q and Adam (the collective is
            #      torch.distributed's), HIP does loss + dL/dQ; buckets sharded over the ranks,
            #      one all-reduce of S^2 + 1 doubles per epoch
            init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])  # pipelines' default init
            module = cherryml_amd.RateMatrix(
                num_states=S, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True,
                initialization=init).to(dev)
            # STRONG scaling of ONE bank: the same 1,057,194 cherry x contact pairs whatever N is (the
            # epoch's cost does not depend on the pair count, SURVEY 8d, so "every GPU brings its own
            # pairs" would grow `value` N-fold by construction).  Families are sharded: every rank holds
            # the sufficient statistics of its 1/N of the families (here: 1/N of the bank's counts); ONE
            # reduce-scatter over the non-empty buckets sums them and leaves each rank with the buckets
            # it owns; then per epoch one all-reduce of S^2 + 1 doubles.
            if shard_of:
                # ONE rank's share of a shard_of-rank job, alone on this GPU: the buckets rank 0 would own (every
                # shard_of-th non-empty bucket), the whole replicated part, the same C-driven loop and enqueue sequence
                # (the two ncclAllReduce calls go to a single-rank communicator).  A self-consistent smaller problem: the
                # loss is normalised by the share's own count, so its Adam trajectory -- and with it the warm eigensolver's
                # work -- is that of a bank with an eighth of the buckets spread over the same branch lengths.
                sharded = ShardedBank(wl["t"], wl["C"], dtype=bank_dtype, emulate=(0, shard_of))
                n_pairs_total = sharded.total_count
                sharding = (f"rank 0's share of a {shard_of}-rank job, alone on one GPU: {len(sharded.local_buckets)} of "
                            f"{wl.get('live', wl['C'].shape[0])} non-empty buckets + the replicated eigensolver / K4 / "
                            "parameter step; in-library all-reduce on a single-rank communicator")
            else:
                sharded = ShardedBank.from_rank_counts(wl["t"], torch.tensor(wl["C"] / world, device=dev), dtype=bank_dtype)
                n_pairs_total = wl["n_pairs"]
                sharding = (f"one bank; families x{world} -> reduce-scatter of the counts over buckets (once), "
                            f"buckets x{world}, all-reduce(loss, dL/dA) per epoch; eigensolver replicated")
            bank = sharded.bank
            scaling = "strong"
            prewarm_ms = None
            B_local = len(sharded.local_buckets)
            in_library = not args.torch_glue
            if in_library:
                try:   # raises on EVERY rank when any rank cannot make its communicator (distributed.py)
                    sharded.enable_in_library_allreduce()
                except Exception as exc:  # no raw RCCL communicator: keep torch.distributed's
                    print(f"warning: in-library all-reduce unavailable ({exc}); torch glue", file=sys.stderr)
                    in_library = False
            if in_library:
                # ---- the whole sharded loop from C on every rank: theta->A, replicated eigensolve,
                #      this rank's buckets, ncclAllReduce(loss, dL/dA) on the handle's stream, Adam
                glue = ("whole loop on the device, driven from C on every rank (theta->A, eigh, own buckets, "
                        "in-library ncclAllReduce, Adam in HIP)")
                u0 = module.upper_diag.detach().cpu().numpy().copy()
                p0 = module._pi.detach().cpu().numpy().copy()
                # the K timed epochs CONTINUE the optimisation of the W warm-up epochs (CB_TRAIN_RESUME): epochs W .. W+K-1
                call = lambda E, resume=False: sharded.train_pande_reversible(  # noqa: E731
                    u0, p0, mask=wl["mask"], num_epochs=E, lr=0.1, resume=resume)
                if warmup > 0:
                    tp0 = time.perf_counter()
                    call(PREWARM_EPOCHS)   # (see the single-GPU branch below: one-time events of a fresh process)
                    prewarm_ms = (time.perf_counter() - tp0) * 1e3
                    call(warmup)
                bank.profile(True, every=PHASE_EVENTS_EVERY if steps >= 2 * PHASE_EVENTS_EVERY else 1)
                fence()
                t0 = time.perf_counter()
                r = call(steps, warmup > 0)
                resumed = warmup > 0
                fence()
                dt = time.perf_counter() - t0
                tm = bank.timing_means()
                bank.profile(False)
                final_loss = float(r["loss"][-1])
            else:
                opt = torch.optim.Adam(module.parameters(), lr=0.1)
                glue = "theta->Q and Adam in torch, loss + dL/dQ in HIP"

                def step():
                    opt.zero_grad()
                    loss = sharded.loss(module(), module.stationary(), normalize=True)[0]
                    loss.backward()
                    opt.step()
                    return loss

                for _ in range(warmup):
                    step()
                bank.profile(True)
                fence()
                t0 = time.perf_counter()
                for _ in range(steps):
                    last = step()
                fence()
                dt = time.perf_counter() - t0
                tm = bank.timing_means()
                bank.profile(False)
                final_loss = float(last.item())
            kernel_ms = None
            bank = sharded   # closed below (bank + communicator)
        else:
            # ---- the WHOLE epoch loop on the device (cb_train_pande_reversible /
            #      cb_train_siterm): K steps = K epochs of it, no torch in the loop.
            #      co-evolution on one GPU: C-driven kernel sequence (train_large.hip.h);
            #      LG: one 0.4 MB bank does not shard -> N independent replicas;
            #      SiteRM: every rank owns its own `--sites` sites (no collective).
            bank = cherryml_amd.CherryBank(wl["t"], wl["C"], device=local_rank, dtype=bank_dtype)
            n_pairs_total, scaling = wl["n_pairs"] * world, "weak"
            B_local = wl.get("live", wl["C"].shape[-3])   # empty buckets cost nothing (B_ne in the formulas)
            glue = ("whole loop on the device, driven from C (theta->A, eigh, bank, grads, Adam in HIP)" if S > 32
                    else "whole loop fused in one HIP kernel (theta->Q, eigh, bank, grads, Adam)")
            if wl["kind"] == "single":
                init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
                mod = cherryml_amd.RateMatrix(
                    num_states=S, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                    pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True,
                    initialization=init)
                u0 = mod.upper_diag.detach().numpy().copy()
                p0 = mod._pi.detach().numpy().copy()
                sharding = f"replicas only x{world}"
                # S > 32: the K timed epochs CONTINUE the optimisation of the W warm-up epochs (CB_TRAIN_RESUME), i.e. they
                # are epochs W .. W+K-1 of one training; the small-state trainers restart (their epoch cost does not
                # depend on the iterate)
                call = lambda E, resume=False: bank.train_pande_reversible(  # noqa: E731
                    u0, p0, mask=wl["mask"], num_epochs=E, lr=0.1, resume=resume and S > 32)
            else:
                from cherryml_amd._siterm._vectorized import _invert
                th0, Th0 = _invert(wl["init"])
                sharding = f"sites x{world} (no collective)"
                call = lambda E, resume=False: bank.train_siterm(th0, Th0, E, lr=0.1)  # noqa: E731
            prewarm_ms = None
            if S > 32 and warmup > 0:
                # One-time events of a fresh process (first launch of every kernel, the runtime's lazily grown pools: a host
                # stall of 70-95 ms once per process, at a random place in its first ~10 epochs: CB_TRACE_SLOW) are taken
                # out of the way by a throw-away optimisation BEFORE the W warm-up epochs; the W + K epochs that follow start
                # from scratch again (resume = False), so the timed epochs are still epochs W .. W+K-1 of one optimisation.
                tp0 = time.perf_counter()
                call(PREWARM_EPOCHS)
                prewarm_ms = (time.perf_counter() - tp0) * 1e3
            if warmup > 0:
                call(warmup)
            bank.profile(True, every=PHASE_EVENTS_EVERY if S > 32 and steps >= 2 * PHASE_EVENTS_EVERY else 1)
            fence()
            t0 = time.perf_counter()
            r = call(steps, warmup > 0)
            resumed = warmup > 0 and S > 32 and wl["kind"] == "single"
            fence()
            dt = time.perf_counter() - t0
            if S > 32:
                tm = bank.timing_means()
                kernel_ms = None
            else:
                kernel_ms = bank.last_timings()["small"]
                tm = {"small": kernel_ms / steps, "calls": 1}
            bank.profile(False)
            final_loss = float(np.sum(r["loss"][-1]) if "loss" in r
                               else np.sum(r["loss_per_epoch_per_site"][-1]))
        tm_max, rccl_ranks = None, None
        if world > 1 or args.force_sharded or shard_of:
            rccl_ranks = getattr(getattr(bank, "rccl", None), "count", None)   # ncclCommCount of the raw communicator
        if world > 1:
            tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
            dt = float(tdt.item())
            # the slowest rank sets the pace: per-phase maxima over the ranks beside rank 0's own phase times
            keys = [k for k in ("eigh", "k1", "k2", "k3", "k4", "allreduce", "small") if k in tm]
            tv = torch.tensor([float(tm[k]) for k in keys], dtype=torch.float64, device=dev)
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
            tm_max = {k: round(float(v), 4) for k, v in zip(keys, tv.tolist())}
        if rank != 0:
            bank.close()   # every rank releases its handle (and its RCCL communicator)
            return None
        traffic = _pmc_traffic()
        bank_form = (bank.bank if hasattr(bank, "bank") else bank).last_bank_form() if S > 32 else None
        if S > 32:
            # dominant MFMA kernels: K1/K2/K3 are one batched 2 S^3 B GEMM each
            flops = 2.0 * S ** 3 * B_local
            names = {"k1": "k1_pt_loss_gt", "k2": "k2_t_eq_g_u", "k3": "k3_w_phi"}
            tn_ = -(-S // 80)
            tri = (tn_ * (tn_ + 1) / 2) / float(tn_ * tn_)   # share of 80x80 tiles actually multiplied
            # whole epoch: SURVEY 8d's algorithmic flops of one epoch (6 B S^3 bank + ~13 S^3 eigensolver
            # and back-rotation; 49.6 GFLOP at S = 400, B = 129) over the WALL time of one step
            epoch_flops = 6.0 * (B_local if shard_of else wl.get("live", wl["C"].shape[0])) * S ** 3 + 13.0 * S ** 3
            epoch_tflops = epoch_flops / (dt / steps) / 1e12
            util, util_src = _mfma_util(bank_dtype)
            fused = bank_form["fused"]   # CB_T_K1 = the one span of the fused launch
            tbasis = bank_form.get("time_basis", False)
            # symmetric counts (every bench bank): the buckets are summed BEFORE the last product -- no third product per
            # bucket ("bucket sum first"): the launch then holds K1 and K2 only
            sum_first = bank_form["bucket_sum_first"]
            n_prod = 2.0 if sum_first else 3.0
            if tbasis:
                # The bank in a TIME BASIS (csrc/tbasis.hip.h): products on ns + nd forward and ng gradient VIRTUAL buckets,
                # one elementwise kernel over all B buckets in between.  Phases: k1 = tables + the forward products, k2 = the
                # elementwise kernel, k3 = the two gradient products (two launches), k4 = sum over the virtual buckets + K4.
                info = (bank.bank if hasattr(bank, "bank") else bank).time_basis_info()
                ns, nd, ng = info["forward_skeleton"], info["direct"], info["gradient_skeleton"]
                one = 2.0 * S ** 3                                   # one S^3 product
                LDp = -(-S // 16) * 16
                ew_pairs = float(LDp) * LDp * B_local                # (element, bucket) pairs the elementwise kernel visits
                ew_mfma = 2.0 * ew_pairs * ((16 if ns <= 16 else 24) + (32 if ng <= 32 else 48))   # its two contractions on MFMA
                kern = {
                    "k1": dict(kernel="k1_pt_loss_gt<double, RAW> (forward products: %d skeleton + %d long-branch buckets)" % (ns, nd),
                               bound="mfma", algorithmic=(ns + nd) * one, executed=(ns + nd) * one * tri, unit="TFLOP/s", peak=F64_PEAK_TFLOPS),
                    "k2": dict(kernel="tb_ew (every element of all %d buckets: P, log P, 1 / P, loss, G, the %d sums)" % (B_local, ng),
                               bound="hbm", algorithmic=8.0 * S * S * (B_local + ns + nd + ng), executed=None, unit="GB/s", peak=HBM_PEAK_GBS),
                    "k3": dict(kernel="k2_t_eq_g_u<%s> + k3_w_phi<%s> (gradient products on %d virtual buckets, two launches)"
                                      % (("double", "double", ng) if bank_dtype == "f64" else ("float", "float", ng)),
                               bound="mfma", algorithmic=2 * ng * one, executed=ng * one * (1.0 + tri), unit="TFLOP/s",
                               peak=F64_PEAK_TFLOPS if bank_dtype == "f64" else F32_PEAK_TFLOPS),   # (mixed: float32 MFMA)
                }
                for k, d in kern.items():
                    sc = 1e12 if d["unit"] == "TFLOP/s" else 1e9
                    d["ms"] = tm[k]
                    d["achieved"] = d["algorithmic"] / (tm[k] * 1e-3) / sc if tm[k] > 0 else 0.0
                    d["frac"] = d["achieved"] / d["peak"]
                    if d["executed"] is not None:
                        d["executed_frac"] = d["executed"] / (tm[k] * 1e-3) / sc / d["peak"] if tm[k] > 0 else 0.0
                kern["k2"]["fp64_pipe_frac"] = (ew_mfma + 60.0 * ew_pairs) / (tm["k2"] * 1e-3) / 1e12 / F64_PEAK_TFLOPS if tm["k2"] > 0 else 0.0
                dom = max(kern, key=lambda k: tm[k])
                t_bank = tm["k1"] + tm["k2"] + tm["k3"]
                d = kern[dom]
                roofline = dict(bound=d["bound"], kernel=d["kernel"], achieved=d["achieved"], peak=d["peak"], unit=d["unit"], frac=d["frac"],
                                **({"executed_frac": d["executed_frac"]} if "executed_frac" in d else {}),
                                epoch_frac=epoch_tflops / (F64_PEAK_TFLOPS * world), epoch_tflops=epoch_tflops, epoch_flops=epoch_flops,
                                traffic=traffic.get("tb:" + dom) if world == 1 else None,
                                traffic_source=_traffic_source() if world == 1 else None,
                                ms_per_launch=tm[dom], algorithmic_per_launch=d["algorithmic"],
                                time_basis=dict(forward_skeleton=ns, long_branch_buckets=nd, gradient_skeleton=ng, rho_max=info["rho_max"],
                                                bases_built=info["builds"], repeated_epochs=info["repeated_epochs"]),
                                bank_vs_per_bucket_pricing=dict(
                                    ms=round(t_bank, 4), flops_6BS3=3.0 * flops, tflops=3.0 * flops / (t_bank * 1e-3) / 1e12,
                                    x_f64_peak=3.0 * flops / (t_bank * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                                    executed_flops=kern["k1"]["executed"] + kern["k3"]["executed"] + ew_mfma,
                                    executed_frac=(kern["k1"]["executed"] + kern["k3"]["executed"] + ew_mfma) / (t_bank * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                                    note="SURVEY 8d prices the bank at 6 B S^3 (three products per bucket); the time basis needs "
                                         "(ns + nd) + 2 ng products instead of 3 B, so the bank as a whole runs ABOVE the f64 peak "
                                         "at that price (x_f64_peak) -- the algorithm got cheaper; executed_frac is what the "
                                         "matrix pipe multiplies in the three phases over their time"),
                                per_kernel={k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kern.items()},
                                note="the dominant bank phase of a time-basis evaluation (phase_ms: k1 = forward products, k2 = "
                                     "elementwise kernel, k3 = gradient products); achieved = that phase's OWN algorithmic work "
                                     "(S^3 products on its virtual buckets, Pt / W symmetric halves counted whole; bytes the "
                                     "elementwise kernel must move once) / its time; epoch_frac prices the WHOLE epoch at SURVEY's "
                                     "6 B S^3 + 13 S^3 whatever is executed")
            elif fused:
                # K1 -> K2 (-> K3) as ONE persistent launch (k123_bank): algorithmic flops = its products, 2 B S^3 each
                # (SURVEY 8d); executed = (tri + 1 [+ tri]) of 2 B S^3: Pt and, with symmetric counts, W are symmetric
                sym3 = 0.0 if sum_first else tri
                executed = flops * (tri + 1.0 + sym3)
                achieved = n_prod * flops / (tm["k1"] * 1e-3) / 1e12
                peak = F32_PEAK_TFLOPS if bank_dtype == "f32" else F64_PEAK_TFLOPS
                kname = {"f64": "k123_bank<double, double>", "f32": "k123_bank<float, float>",
                         "mixed": "k123_bank<double, float>"}[bank_dtype]
                # mixed: K1 on the f64 pipe, K2 / K3 on the f32 pipe (twice the rate): time-weighted peak of the launch
                if bank_dtype == "mixed":
                    t64, t32 = tri / F64_PEAK_TFLOPS, (1.0 + sym3) / F32_PEAK_TFLOPS
                    peak = (tri + 1.0 + sym3) / (t64 + t32)
                exec_tflops = executed / (tm["k1"] * 1e-3) / 1e12
                roofline = dict(bound="mfma", kernel=kname, achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak,
                                executed_tflops=exec_tflops, executed_frac=exec_tflops / peak,
                                epoch_frac=epoch_tflops / (peak * world), epoch_tflops=epoch_tflops, epoch_flops=epoch_flops,
                                traffic=traffic.get("k123_bank" + ("_f32" if bank_dtype == "f32" else "_mixed" if bank_dtype == "mixed" else ""))
                                if world == 1 else None,
                                traffic_source=_traffic_source() if world == 1 else None,
                                ms_per_launch=tm["k1"], flops_per_launch=n_prod * flops, executed_flops_per_launch=executed,
                                products_in_launch="K1, K2 (buckets summed before the last product: no K3)" if sum_first else "K1, K2, K3",
                                note="ONE persistent launch for the bank products (tickets per XCD, large_bank.hip.h); "
                                     "ms_per_launch = HIP events from the end of the eigensolver to the end of the launch "
                                     "(includes lg_tables, ~4 us).  achieved = the algorithmic 2 S^3 B flops (SURVEY 8d) of each "
                                     f"product IN the launch / launch time; executed = what the matrix pipe multiplies: {tri:.2f} "
                                     "of K1's tiles (Pt symmetric) + all of K2's" + ("" if sum_first else f" + {tri:.2f} of K3's") +
                                     "; executed_frac is the pipe's own utilisation.  epoch_frac prices the WHOLE epoch at "
                                     "SURVEY's 6 B S^3 + 13 S^3 whatever is executed",
                                mfma_util=next((v for k, v in util.items() if k.startswith("k123")), None), mfma_util_source=util_src)
            else:
                if sum_first:
                    names.pop("k3")   # (CB_T_K3 is then the bucket sums + their 7 single products + the combination: no B-fold product)
                dom = max(names, key=lambda k: tm[k])
                achieved = flops / (tm[dom] * 1e-3) / 1e12 if tm[dom] > 0 else 0.0
                # (mixed: the dominant kernel is then K1 in float64; K2 / K3 run on the f32 MFMA)
                dom_f32 = bank_dtype == "f32" or (bank_dtype == "mixed" and dom != "k1")
                peak = F32_PEAK_TFLOPS if dom_f32 else F64_PEAK_TFLOPS
                peak_of = lambda k: (F32_PEAK_TFLOPS if (bank_dtype == "f32" or (bank_dtype == "mixed" and k != "k1"))  # noqa: E731
                                     else F64_PEAK_TFLOPS)
                roofline = dict(bound="mfma", kernel=names[dom] + ("<float>" if dom_f32 else "<double>"),
                                achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak,
                                epoch_frac=epoch_tflops / (peak * world), epoch_tflops=epoch_tflops,
                                epoch_flops=epoch_flops,
                                traffic=traffic.get(names[dom] + ("_f32" if dom_f32 else "_mixed" if bank_dtype == "mixed" else ""))
                                if world == 1 else None,
                                traffic_source=_traffic_source() if world == 1 else None,
                                ms_per_launch=tm[dom], flops_per_launch=flops,
                                note="achieved = algorithmic 2 S^3 B flops (SURVEY 8d) / launch time; k1 (Pt symmetric) "
                                     f"and k3 (symmetric counts) multiply only the upper-triangular tiles, {tri:.2f} of "
                                     "those flops; k2 multiplies all of them",
                                per_kernel_tflops={k: round(flops / (tm[k] * 1e-3) / 1e12, 2) for k in names if tm[k] > 0},
                                # what the matrix pipe really executes: k1 / k3 run `tri` of the algorithmic flops
                                per_kernel_executed_tflops={k: round((flops if k == "k2" else flops * tri) / (tm[k] * 1e-3) / 1e12, 2)
                                                            for k in names if tm[k] > 0},
                                per_kernel_executed_frac={k: round((flops if k == "k2" else flops * tri) / (tm[k] * 1e-3) / 1e12 / peak_of(k), 3)
                                                          for k in names if tm[k] > 0},
                                per_kernel_mfma_util=util, mfma_util_source=util_src)
        else:
            Lb = wl["C"].shape[0] if wl["kind"] == "sites" else 1
            nbytes = float(Lb) * B_local * S * S * 8  # C streamed once per epoch
            achieved = nbytes / (tm["small"] * 1e-3) / 1e9 if tm["small"] > 0 else 0.0
            kname = ("sp_prepare + sp_bank + sp_finish (3 launches per epoch)" if Lb >= 64 else
                     "sp_step (= sp_finish + sp_prepare) + sp_bank (2 launches per epoch)")
            if Lb == 1:
                # ONE 20-state bank is not a roofline case ("launch / latency; report but not a roofline case"):
                # 0.41 MB against 8 TB/s says nothing -- the line carries `latency_budget` instead (below)
                roofline = dict(bound="latency", kernel=kname, achieved=None, peak=None, unit=None, frac=None, traffic=None,
                                ms_per_epoch_in_kernel=tm["small"], bytes_per_epoch=nbytes,
                                note="not a roofline case (SURVEY 8d): an epoch is two dependent launches of serial sections; "
                                     "see latency_budget")
            else:
                roofline = dict(bound="hbm", kernel=kname, achieved=achieved,
                                peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                                traffic=traffic.get("epoch:" + workload) if world == 1 else None,
                                traffic_source=_traffic_source() if world == 1 else None,
                                ms_per_epoch_in_kernel=tm["small"], bytes_per_epoch=nbytes,
                                compute_floor=_siterm_compute_floor(),
                                note="figures are per epoch; HBM only streams the counts once (counter traffic 1.02x algorithmic). "
                                     "What bounds sp_bank is the float64 pipe, which on gfx950 the f64 MFMA and the f64 vector "
                                     "instructions share (table logarithm, reciprocal, divided difference per count entry): "
                                     "compute_floor = its SQ counters' MFMA-busy + vector-issue cycles over the chip's SIMD cycles")
        out = {
            "metric": "cherry-pairs/sec (whole node) per EM iter",
            "value": n_pairs_total / (dt / steps), "unit": "cherry-pairs/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": bank_dtype,
            "data": "synthetic" if os.environ.get("CB_BENCH_BACKEND", "nccl") == "nccl" else
                    "synthetic; TEST HOOK CB_BENCH_BACKEND (ranks may share one GPU): control-flow check, not a measurement",
            "config": {"workload": wl["desc"], "states": S, "buckets": 129,
                       **({"non_empty_buckets": wl["live"]} if "live" in wl else {}), "sharding": sharding,
                       "epoch": glue,
                       **({"timed_epochs": f"{warmup} .. {warmup + steps - 1} of one optimisation from the JTT-IPW start "
                                           "(the warm-up epochs are its first ones; CB_TRAIN_RESUME)",
                           "prewarm": f"{PREWARM_EPOCHS} throw-away epochs of the same optimisation in front of the warm-up "
                                      "(one-time host stalls of a fresh process: EXPERIMENTS.md section 11)"} if resumed else {}),
                       **({"arithmetic": "float32 operands + float32 MFMA accumulation in P_b, G_b U, (T_b^T U) o Phi_b; "
                                         "eigensolver, loss sums, divided differences, bucket sum, Adam in float64"}
                          if bank_dtype == "f32" else {})},
            "roofline": roofline,
            "epochs_per_s": steps / dt,   # the epoch's cost does not depend on the pair count
            "phase_ms": {k: round(v, 4) for k, v in tm.items()},
            **({"phase_events": f"recorded in every {PHASE_EVENTS_EVERY}th epoch of the timed region"}
               if S > 32 and steps >= 2 * PHASE_EVENTS_EVERY else {}),
            **({"bank_form": bank_form} if bank_form else {}),
            "final_loss": final_loss,
        }
        if resumed and prewarm_ms is not None:
            # the throw-away optimisation in front of the warm-up takes the one-off costs of a fresh process out of
            # the timed window; what a FIRST call in a fresh process pays is reported beside it (wall time of that call:
            # uploads, first launches of every kernel, the cold eigensolve, its 30 epochs, the read-back)
            out["prewarm"] = {"epochs": PREWARM_EPOCHS, "first_call_ms": round(prewarm_ms, 3),
                              "first_call_ms_per_epoch": round(prewarm_ms / PREWARM_EPOCHS, 4),
                              "note": "a throw-away optimisation of the same bank before the W warm-up epochs; ms_per_step "
                                      "excludes it, first_call_ms is what a fresh process pays for its first 30 epochs"}
        if S > 32:
            # planned (device-controlled) warm eigensolves of the timed call: how many, continuations, host spins
            out["eigh"] = (bank.bank if hasattr(bank, "bank") else bank).eigh_counters()
        if tm_max is not None:
            out["phase_ms_max_over_ranks"] = tm_max
        if shard_of:
            out["shard_of"] = shard_of
            out["config"]["share_buckets"] = int(B_local)
        if world > 1 or args.force_sharded or shard_of:
            # which transport carried the per-epoch all-reduce: the raw RCCL communicator's own rank count (None: torch's)
            out["rccl_ranks"] = rccl_ranks
        if S <= 32 and wl["kind"] == "single":
            # one 20-state bank is not a roofline case: an epoch is two dependent launches whose serial sections
            # (theta -> A, the warm eigensolve, partial sums, parameter step: one workgroup) are pure latency
            out["latency_budget"] = {
                "us_per_epoch": round(tm["small"] * 1e3, 2), "target_us": 35.0, "launches_per_epoch": 2,
                "launch_floor_us": 2.9, "algorithmic_bytes_per_epoch": float(B_local * S * S * 8),
                "note": "sp_step (finish of epoch e-1 + prepare of epoch e, ONE workgroup: ~30 us of dependent latency) + "
                        "sp_bank (one quad per wave over the chip: ~13 us); launch floor = a dependent empty kernel on this "
                        "box (profiles/tools/launch_probe2)"}
        if S > 32:
            # the Amdahl arithmetic of sharding ONE bank over ranks: the eigensolver and K4 run on the
            # whole matrix on every rank, K1-K3 on the rank's buckets only; `other` = theta -> A, parameter kernels,
            # the all-reduce (N > 1) and launch gaps
            rep, shd = tm["eigh"] + tm["k4"], tm["k1"] + tm["k2"] + tm["k3"]
            other = max(dt / steps * 1e3 - tm["total"], 0.0)
            out["amdahl"] = {
                "replicated_ms": round(rep, 4), "sharded_ms": round(shd, 4), "other_ms": round(other, 4),
                "measured_on_ranks": world,
                "note": "per epoch on rank 0: replicated = eigensolver + K4 (every rank, whole matrix), sharded = K1 + K2 + K3 "
                        "(this rank's buckets), other = theta->A, parameter kernels, all-reduce, gaps",
                **({"projected_speedup_at_8_arithmetic": round((rep + shd + other) / (rep + shd / 8.0 + other), 2),
                    "projection_arithmetic": "(replicated + sharded + other) / (replicated + sharded / 8 + other), all-reduce "
                                             "not included: an upper bound for one bank on 8 GPUs"}
                   if world == 1 and not shard_of else {})}
            if shard_of:
                out["amdahl"]["measured_rank_ms"] = round(dt / steps * 1e3, 4)
        if with_cpu:
            out["cpu_baseline"] = cpu_baseline(wl, workload)
        bank.close()
        return out

    def finish(out):
        """The JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio,
        which (on a pipe) would otherwise be flushed after Python's line at exit."""
        if dist.is_initialized():
            dist.destroy_process_group()
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if rank == 0:
            sys.stdout.write(json.dumps(out) + "\n")
            sys.stdout.flush()

    defaults = {"co_counting": (10, 2), "coevo400": (200, 5), "coevo400_demo": (200, 5), "lg20": (500, 50), "siterm": (100, 2), "counting": (20, 3), "ble": (5, 1), "assembly": (5, 1),
                "likelihood": (5, 1)}
    steps = args.steps if args.steps is not None else defaults[args.workload][0]
    warmup = args.warmup if args.warmup is not None else defaults[args.workload][1]
    if args.workload == "assembly":
        finish(run_assembly(steps, warmup, world, rank, local_rank, fence, world == 1 and not args.no_cpu_baseline))
        return
    if args.workload == "likelihood":
        finish(run_likelihood(steps, warmup, world, rank, local_rank, fence, world == 1 and not args.no_cpu_baseline))
        return
    if args.workload == "ble":
        finish(run_ble(steps, warmup, world, rank, local_rank, fence, world == 1 and not args.no_cpu_baseline))
        return
    if args.workload == "co_counting":
        finish(run_co_counting(steps, warmup, world, rank, local_rank, fence, world == 1 and not args.no_cpu_baseline))
        return
    if args.workload == "counting":
        out = run_counting(steps, warmup, world, rank, local_rank, fence,
                           world == 1 and not args.no_cpu_baseline)
        finish(out)
        return
    out = run(args.workload, steps, warmup, world == 1 and not args.no_cpu_baseline and not args.shard_of, shard_of=args.shard_of)
    keep = ("value", "unit", "n_gpus", "scaling", "ms_per_step", "kernel_ms_per_step", "steps", "warmup", "dtype", "config", "roofline",
            "cpu_baseline", "phase_ms", "shard_of", "amdahl", "eigh", "prewarm", "bank_form")
    if args.workload == "coevo400" and not args.no_secondary:
        # every rank takes part (the multi-rank runs end in collectives); rank 0 attaches the lines
        extra = {}
        if world == 1 and not args.shard_of:
            # what ONE rank of an 8-rank job does per epoch, measured on this GPU (same steps / warm-up as the headline):
            # the projection of the strong-scaling speed-up of one bank comes from this line, not from dividing phase times
            sh = run(args.workload, steps, warmup, False, shard_of=8)
            extra["secondary_shard8"] = sh
            if rank == 0 and sh is not None and "amdahl" in out:
                rank_ms = sh["ms_per_step"]
                out["amdahl"].update({
                    "measured_rank_ms": round(rank_ms, 4),
                    "measured_rank_phase_ms": sh["phase_ms"],
                    "projected_speedup_at_8": round(out["ms_per_step"] / rank_ms, 2),
                    "projection": "ms_per_step of the whole bank on this GPU / ms_per_step of rank 0's share of an 8-rank job "
                                  "measured alone on this GPU (secondary_shard8: every 8th non-empty bucket + the replicated "
                                  "eigensolver, K4, parameter step, the loop's own enqueue sequence); the 1.28 MB all-reduce "
                                  "over xGMI is NOT in it (single-rank communicator): an upper bound"})
        if world == 1 and not args.shard_of:
            # BASELINE.json config 3 AS IT IS: the co-evolution bank of all 32 demo_data families counted by the reference
            # itself (tests/golden/coevo_demo_full.npz: 43 of 129 buckets non-empty, 3.5 % dense), same steps / warm-up window
            # as the headline
            extra["secondary_demo43"] = run("coevo400_demo", steps, warmup, not args.no_cpu_baseline)
        if world == 1:
            # BASELINE.json's metric names both sizes: the line always carries the 20x20 LG configuration
            # too (its own epoch counts: an LG epoch takes < 0.1 ms)
            extra["secondary"] = run("lg20", *defaults["lg20"], not args.no_cpu_baseline)
        # BASELINE.json configs 4 and 5 -- the workloads that shard WITHOUT a replicated term (weak scaling):
        # SiteRM (sites x N, no collective) and co-transition counting (families x N + one integer all-reduce)
        extra["secondary_siterm"] = run("siterm", *defaults["siterm"], world == 1 and not args.no_cpu_baseline)
        extra["secondary_co_counting"] = run_co_counting(*defaults["co_counting"], world, rank, local_rank, fence,
                                                         world == 1 and not args.no_cpu_baseline)
        # the other widened rows of SURVEY 8f (each a few tens of milliseconds of GPU time; per family, so they shard like
        # SiteRM): LG counting, FastCherries branch lengths, SiteRM assembly, held-out likelihood
        cpu = world == 1 and not args.no_cpu_baseline
        extra["secondary_counting"] = run_counting(*defaults["counting"], world, rank, local_rank, fence, cpu)
        extra["secondary_ble"] = run_ble(*defaults["ble"], world, rank, local_rank, fence, cpu)
        extra["secondary_assembly"] = run_assembly(*defaults["assembly"], world, rank, local_rank, fence, cpu)
        extra["secondary_likelihood"] = run_likelihood(*defaults["likelihood"], world, rank, local_rank, fence, cpu)
        if rank == 0:
            for name, sec in extra.items():
                if sec is not None:
                    out[name] = {k: sec[k] for k in keep + ("latency_budget", "phase_ms_max_over_ranks") if k in sec}
    finish(out)


def run_counting(steps, warmup, world, rank, local_rank, fence, with_cpu):
    """cb_count_transitions on a config-2-shaped synthetic input: 1000 families x 64 cherries
    x 200 sites per rank (families shard over ranks like the reference's MPI ranks; the
    integer counts are all-reduced once per pass)."""
    import ctypes
    import torch
    import torch.distributed as dist
    from cherryml_amd import _lib
    from cherryml_amd.counting._stage import PAIR_DTYPE

    rng = np.random.default_rng(1 + rank)
    F, NCH, L, S = 1000, 64, 200, 20
    grid = quantization_grid()
    B = grid.size
    seqs = rng.integers(0, S, size=(F, 2 * NCH, L), dtype=np.int8)
    same = rng.random((F, NCH, L)) < 0.7          # leaves of a cherry mostly agree
    seqs[:, 1::2][same] = seqs[:, 0::2][same]
    seqs[rng.random(seqs.shape) < 0.05] = -1       # gaps
    rates = np.round(rng.gamma(3.0, 1.0 / 3.0, size=(F, L)) + 0.01, 5)
    pairs = np.zeros(F * NCH, dtype=PAIR_DTYPE)
    fam = np.repeat(np.arange(F), NCH)
    ch = np.tile(np.arange(NCH), F)
    pairs["seq_a"] = (fam * 2 * NCH + 2 * ch) * L
    pairs["seq_b"] = (fam * 2 * NCH + 2 * ch + 1) * L
    pairs["aux"] = fam * L
    pairs["n"] = L
    pairs["len_a"] = rng.exponential(0.2, size=F * NCH)
    pairs["len_b"] = rng.exponential(0.2, size=F * NCH)
    dev = torch.device("cuda", local_rank)
    d_seqs = torch.from_numpy(seqs.reshape(-1)).to(dev)
    d_rates = torch.from_numpy(rates.reshape(-1)).to(dev)
    d_grid = torch.from_numpy(grid).to(dev)
    d_pairs = torch.from_numpy(pairs.view(np.uint8)).to(dev)
    d_counts = torch.zeros(B * S * S, dtype=torch.int64, device=dev)
    lib = _lib.load()

    def step():
        d_counts.zero_()
        rc = lib.cb_count_transitions(local_rank, S, B, d_grid.data_ptr(), d_seqs.data_ptr(),
                                      d_seqs.numel(), d_rates.data_ptr(), d_rates.numel(),
                                      d_pairs.data_ptr(), len(pairs), 1, _lib.CB_PTR_DEVICE | (L << 8),
                                      d_counts.data_ptr())
        _lib.check(rc, "cb_count_transitions")
        if world > 1:
            dist.all_reduce(d_counts, op=dist.ReduceOp.SUM)

    for _ in range(warmup):
        step()
    fence()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(steps):
        step()
    ev[1].record()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
    if rank != 0:
        return None
    counted = float(d_counts.sum().item()) * 0.5   # all ranks' pairs after the all-reduce
    kernel_ms = ev[0].elapsed_time(ev[1]) / steps  # default stream: zero + histogram (+ all-reduce)
    # algorithmic bytes of one pass: both code bytes of every pair x site, the pair records,
    # the site rates once per family and the count tensor once (atomics are not "bytes moved")
    nbytes = 2.0 * len(pairs) * L + pairs.nbytes + rates.nbytes + B * S * S * 8
    achieved = nbytes / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "cherry-pairs/sec (whole node) counted per pass", "value": counted / (dt / steps),
        "unit": "cherry-pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "i8/u64", "data": "synthetic",
        "config": {"workload": f"counting: {F} families x {NCH} cherries x {L} sites per GPU, "
                               "B=129 grid, cherry (symmetric) mode", "states": S, "buckets": B,
                   "sharding": f"families x{world}, all-reduce of integer counts"},
        # The pass is bound by LDS atomics, not by HBM (30.7 MB of algorithmic bytes in 85 us = 4.6 % of the HBM peak says
        # nothing): every valid (cherry, site) is two ds_add_u32 into the workgroup's histogram.  Peak: an LDS atomic is a
        # 4-byte LDS write cycle-wise (MI355X_MICROARCH.md, LDS table: ds_write_b32 = 4 cycles per wave-instruction,
        # i.e. 16 lanes per clock per CU) x 256 CUs x 2.4 GHz = 9.83e12 lane-atomics / s when conflict-free.
        "roofline": {"bound": "lds_atomic", "kernel": "count_transitions_lds_kernel",
                     "achieved": 2.0 * counted / world / (kernel_ms * 1e-3) / 1e9, "peak": LDS_ATOMIC_PEAK_G,
                     "unit": "G lane-atomics/s", "frac": 2.0 * counted / world / (kernel_ms * 1e-3) / 1e9 / LDS_ATOMIC_PEAK_G,
                     "hbm_achieved_GBs": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
                     "traffic": _pmc_traffic().get("pass:counting"), "ms_per_launch": kernel_ms, "bytes_per_launch": nbytes,
                     "note": "per-workgroup LDS histogram (packed 16-bit bins) + slab reduction; achieved = 2 LDS atomics per "
                             "counted pair-site; what keeps it far from the LDS peak is the branch-free grid search in front "
                             "of every atomic (~25 % of the pass) and the slab write + reduction"},
        "counted_pairs": counted,
    }
    if with_cpu:
        from oracle import counting_oracle as co
        npairs = 150
        t0 = time.time()
        C = np.zeros((B, S, S))
        for p in pairs[:npairs]:
            sa = seqs.reshape(-1)[p["seq_a"]:p["seq_a"] + L]
            sb = seqs.reshape(-1)[p["seq_b"]:p["seq_b"] + L]
            rt = rates.reshape(-1)[p["aux"]:p["aux"] + L]
            total = p["len_a"] + p["len_b"]
            for k in range(L):
                q = co.quantization_idx(total * rt[k], grid)
                if q is None or sa[k] < 0 or sb[k] < 0:
                    continue
                C[q, sa[k], sb[k]] += 0.5
                C[q, sb[k], sa[k]] += 0.5
        cdt = time.time() - t0
        out["cpu_baseline"] = {"value": float(C.sum()) / cdt, "unit": "cherry-pairs/s", "cores": 1,
                               "kind": "port", "sample": f"{npairs} of {len(pairs)} cherries x {L} sites, "
                               "the oracle's per-site loop (the reference's Python counter)"}
    return out


def make_co_counting_input(rng, F, NCH=64, L=200):
    """A config-5-shaped synthetic input (BASELINE.json: co-evolution, 10k families, family-sharded): F families of NCH
    cherries (2 NCH leaves) x L sites, 20 letters, per family a set of disjoint contacting site pairs >= 7 apart (what
    the reference's maximal matching of a contact map leaves: _maximal_matching.py:70-93), cherry lengths ~ 2 Exp(0.2)."""
    from cherryml_amd.counting._stage import PAIR_DTYPE
    S = 20
    seqs = rng.integers(0, S, size=(F, 2 * NCH, L), dtype=np.int8)
    same = rng.integers(0, 10, size=(F, NCH, L), dtype=np.int8) < 7     # the leaves of a cherry mostly agree (70 %)
    np.copyto(seqs[:, 1::2], seqs[:, 0::2], where=same)
    del same
    np.copyto(seqs, np.int8(-1), where=rng.integers(0, 20, size=seqs.shape, dtype=np.int8) < 1)   # 5 % gaps
    perm = np.argsort(rng.random((F, L)), axis=1).astype(np.int32).reshape(F, L // 2, 2)
    perm.sort(axis=2)
    ok = perm[:, :, 1] - perm[:, :, 0] >= 7
    want = rng.integers(40, 90, size=F)
    ok &= np.cumsum(ok, axis=1) <= want[:, None]
    n_c = ok.sum(axis=1)
    contacts = perm[ok].reshape(-1)                           # family-major, [sum n_c][2]
    c_off = np.concatenate([[0], np.cumsum(n_c)[:-1]])
    pairs = np.zeros(F * NCH, dtype=PAIR_DTYPE)
    fam = np.repeat(np.arange(F), NCH)
    ch = np.tile(np.arange(NCH), F)
    pairs["seq_a"] = (fam * 2 * NCH + 2 * ch) * L
    pairs["seq_b"] = (fam * 2 * NCH + 2 * ch + 1) * L
    pairs["aux"] = c_off[fam]
    pairs["n"] = n_c[fam]
    pairs["len_a"] = rng.exponential(0.2, size=F * NCH)
    pairs["len_b"] = rng.exponential(0.2, size=F * NCH)
    return S, seqs.reshape(-1), contacts.astype(np.int32), pairs, int(n_c.max())


def run_co_counting(steps, warmup, world, rank, local_rank, fence, with_cpu, families=10000):
    """cb_count_co_transitions on a config-5-shaped synthetic input: `families` families x 64 cherries x ~65 contacting
    site pairs PER RANK (families shard over the ranks exactly as the reference's MPI ranks take them,
    _count_co_transitions.cpp:626-628), a step = one pass over this rank's families into the resident
    [129][400][400] 8-byte histogram (165 MB) and, at N > 1, the all-reduce of those integer counts (what one EM
    iteration exchanges: the summed sufficient statistics).  Weak scaling."""
    import torch
    import torch.distributed as dist
    from cherryml_amd import _lib

    rng = np.random.default_rng(11 + rank)
    grid = quantization_grid()
    B = grid.size
    S, seqs, contacts, pairs, max_n = make_co_counting_input(rng, families)
    S2 = S * S
    dev = torch.device("cuda", local_rank)
    d_seqs = torch.from_numpy(seqs).to(dev)
    d_contacts = torch.from_numpy(contacts).to(dev)
    d_grid = torch.from_numpy(grid).to(dev)
    d_pairs = torch.from_numpy(pairs.view(np.uint8)).to(dev)
    d_counts = torch.zeros(B * S2 * S2, dtype=torch.int64, device=dev)
    lib = _lib.load()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def step(timed=False):
        d_counts.zero_()
        if timed:
            ev[0].record()
        rc = lib.cb_count_co_transitions(local_rank, S, B, d_grid.data_ptr(), d_seqs.data_ptr(), d_seqs.numel(),
                                         d_contacts.data_ptr(), d_contacts.numel() // 2, d_pairs.data_ptr(), len(pairs), 1,
                                         _lib.CB_PTR_DEVICE | (max_n << 8), d_counts.data_ptr())
        _lib.check(rc, "cb_count_co_transitions")
        if timed:
            ev[1].record()
        if world > 1:
            dist.all_reduce(d_counts, op=dist.ReduceOp.SUM)
        if timed:
            ev[2].record()

    for _ in range(warmup):
        step()
    fence()
    kms, ams = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
        torch.cuda.synchronize()
        kms.append(ev[0].elapsed_time(ev[1]))
        ams.append(ev[1].elapsed_time(ev[2]))
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
    if rank != 0:
        return None
    counted = float(d_counts.sum().item()) * 0.25            # all ranks' (cherry, contact) events after the all-reduce
    kernel_ms = float(np.mean(kms))
    n_events = int(pairs["n"].sum())
    # algorithmic bytes of one pass (this rank): every leaf sequence once, the pair records, the contact lists, and the
    # count tensor cleared and written once (the reference's C++ keeps the same dense [B][400][400] tensor per rank)
    nbytes = float(seqs.nbytes + pairs.nbytes + contacts.nbytes + 2 * B * S2 * S2 * 8)
    achieved = nbytes / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "cherry-pairs/sec (whole node): cherry x contacting-site-pair events counted per pass",
        "value": counted / (dt / steps), "unit": "cherry-pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8/u64", "data": "synthetic",
        "config": {"workload": f"co-transition counting: {families} families x 64 cherries x {n_events / len(pairs):.0f} "
                               "contacting site pairs per GPU, 20 letters (400 pair states), B=129, cherry mode",
                   "states": S2, "buckets": B,
                   "sharding": f"families x{world}" + (", all-reduce of the 165 MB integer count tensor per pass" if world > 1
                                                       else " (no collective)")},
        "roofline": {"bound": "hbm", "kernel": "co_bucket + co_plan + co_expand + co_count_lds (one pass)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": _pmc_traffic().get("pass:co_counting") if world == 1 else None,
                     "traffic_source": _traffic_source() if world == 1 else None,
                     "ms_per_launch": kernel_ms, "bytes_per_launch": nbytes,
                     "note": "events binned by bucket, 100-row x 400-column LDS histograms per (bucket, row block, event "
                             "chunk), non-zero bins added to the 8-byte global bins once; time excludes the clear"},
        "phase_ms": {"count": kernel_ms, "allreduce": float(np.mean(ams))},
        "counted_events": counted,
    }
    if with_cpu:
        from oracle import counting_oracle as co
        npairs = 40
        codes = seqs.astype(np.int64)
        t0 = time.perf_counter()
        C = np.zeros((B, S2, S2))
        for p in pairs[:npairs]:
            q = co.quantization_idx(float(p["len_a"] + p["len_b"]), grid)
            if q is None:
                continue
            L = 200
            ij = contacts[2 * p["aux"]: 2 * (p["aux"] + p["n"])].reshape(-1, 2).tolist()
            co.co_count_pair(C[q], codes[p["seq_a"]: p["seq_a"] + L].tolist(), codes[p["seq_b"]: p["seq_b"] + L].tolist(),
                             ij, S, True)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(C.sum()) / cdt, "unit": "cherry-pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{npairs} of {len(pairs)} cherries, the oracle's per-contact loop (the reference's "
                                         "Python counter, _count_co_transitions.py:108-140)"}
    return out


def run_ble(steps, warmup, world, rank, local_rank, fence, with_cpu):
    """FastCherries branch lengths / site rates (cb_ble) on one large synthetic family per rank
    (2048 cherries x 512 sites, 129 grid points x 20 rate categories, LG): a step is one whole
    coordinate ascent.  CPU baseline: the reference's own C++ compiled into oracle/_ref."""
    import torch
    import torch.distributed as dist
    from cherryml_amd.phylogeny_estimation import compute_log_transition_matrices, estimate_branch_lengths_and_site_rates
    rng = np.random.default_rng(100 + rank)
    n, L, R = 2048, 512, 20
    Q = lg_matrix()
    grid = quantization_grid()
    rates = np.geomspace(1.0 / R, float(R), R)
    weights = np.arange(1, R + 1) / R
    bank = compute_log_transition_matrices(Q, grid, rates, device=local_rank, stationary_distribution=stationary(Q))
    true_rate = rng.choice(R, size=L, p=np.diff(np.concatenate([[0.0], weights])))
    t_idx = rng.integers(40, 100, size=n)
    anc = rng.integers(0, 20, size=(n, L))
    cy = anc.copy()
    cum = np.cumsum(np.exp(bank), axis=3)               # [T,R,S,S] row-wise cdf
    u = rng.random((n, L))
    rows = cum[t_idx[:, None], true_rate[None, :], anc]  # [n,L,S]
    cy = (u[:, :, None] > rows).sum(axis=2).clip(0, 19)
    cx = anc.copy()
    gaps = rng.random((2, n, L)) < 0.05
    cx[gaps[0]] = -1
    cy[gaps[1]] = -1
    seqs = np.concatenate([cx, cy])
    prof = {}
    # the caller's view (VERDICT r5 "missing 3"): the bank resident on the device, one call per family
    # (cherryml_amd.phylogeny_estimation.BleBank = cb_ble_bank_create / cb_ble_bank_run); ms_per_step is the WALL time of a call
    # -- sequences uploaded, transposed, site statistics, the coordinate ascent, results read back
    from cherryml_amd.phylogeny_estimation import BleBank
    cx8, cy8, seqs8 = (np.ascontiguousarray(a, dtype=np.int8) for a in (cx, cy, seqs))
    resident = BleBank(bank, grid, rates, device=local_rank)
    call = lambda: resident.estimate(cx8, cy8, seqs8, weights, 50, profile=prof)  # noqa: E731
    # (the per-call entry with host arrays of the reference's own dtype: bank re-uploaded, workspace re-allocated, three host passes)
    t0 = time.perf_counter()
    estimate_branch_lengths_and_site_rates(cx, cy, seqs, bank, grid, rates, weights, 50, device=local_rank)
    per_call_entry_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(warmup):
        call()
    fence()
    kms, iters = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        lengths, srates = call()
        kms.append(prof["kernel_ms"])
        iters.append(prof["iterations"])
    fence()
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean(kms))
    call_ms = dt / steps * 1e3
    resident.close()
    if world > 1:
        tdt = torch.tensor([kernel_ms, call_ms], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        kernel_ms, call_ms = (float(v) for v in tdt.tolist())
    if rank != 0:
        return None
    valid = int(((cx >= 0) & (cy >= 0)).sum())
    passes = 1 + 2 * int(np.mean(iters))                # first branch-length pass + 2 per iteration
    # bytes a pass must at least touch: both code bytes of every cherry x site, per bisection step
    steps_bl, steps_sr = int(np.ceil(np.log2(len(grid)))), int(np.ceil(np.log2(R)))
    nbytes = 2.0 * n * L * (steps_bl * (1 + int(np.mean(iters))) + steps_sr * int(np.mean(iters)))
    gather_bytes = 16.0 * n * L * (steps_bl * (1 + int(np.mean(iters))) + steps_sr * int(np.mean(iters)))
    out = {
        "metric": "cherry-pairs/sec (whole node): cherry x site pairs fitted per coordinate ascent",
        "value": valid * world / (call_ms * 1e-3), "unit": "cherry-pairs/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": call_ms, "kernel_ms_per_step": kernel_ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"FastCherries branch lengths + site rates: {n} cherries x {L} sites per GPU, "
                               f"129 grid points x {R} rate categories, LG", "iterations": int(np.mean(iters)),
                   "sharding": f"families x{world} (no collective)",
                   "entry": "BleBank.estimate (cb_ble_bank_run): log-transition bank resident on the device, sequences uploaded per call; "
                            "ms_per_step = wall time of a call, kernel_ms_per_step = the coordinate ascent alone (HIP events)",
                   "per_call_entry_ms": per_call_entry_ms},
        # Bound: 8-byte gathers from the L2-resident 8 MB log-transition bank -- two per (cherry, site, bisection step) --
        # not HBM.  Peak: rows gathered from a table every workgroup shares in the XCD's L2, 16.8 TB/s chip-wide
        # (MI355X_MICROARCH.md, "Indexed rows"); an 8-byte word costs a 64-byte sector there, which `sector_frac` prices.
        "roofline": {"bound": "l2_gather", "kernel": "ble_branch_lengths_kernel + ble_site_rates_kernel",
                     "achieved": gather_bytes / (kernel_ms * 1e-3) / 1e9, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s",
                     "frac": gather_bytes / (kernel_ms * 1e-3) / 1e9 / L2_GATHER_PEAK_GBS,
                     "sector_frac": 8.0 * gather_bytes / (kernel_ms * 1e-3) / 1e9 / L2_GATHER_PEAK_GBS,
                     "hbm_frac": nbytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "bytes_per_step": gather_bytes, "code_bytes_per_step": nbytes, "passes": passes,
                     "note": "achieved = 16 gathered bytes per (cherry, site, bisection step) over the whole coordinate ascent, "
                             "including its one-off bank kernels; wave-uniform bisection with a wavefront reduction per step: "
                             "the dependent gather -> reduce -> compare chain, not bandwidth, sets the time"},
    }
    if with_cpu:
        from oracle import ble_oracle as bo
        if bo.ref_available():
            sub = 256                                    # bounded sample: the first 256 cherries
            t0 = time.perf_counter()
            bo.ref_ble(cx[:sub], cy[:sub], np.concatenate([cx[:sub], cy[:sub]]), bank, grid, rates, weights, 50)
            dtc = (time.perf_counter() - t0) * (n / sub)
            out["cpu_baseline"] = dict(value=valid / dtc, unit="cherry-pairs/s", cores=1, kind="reference",
                                       sample=f"the reference's ble() (oracle/_ref/libref_ble.so) on {sub} of {n} "
                                              "cherries, scaled", seconds_per_step=dtc)
    return out


def run_assembly(steps, warmup, world, rank, local_rank, fence, with_cpu):
    """SiteRM count / pseudocount assembly (cb_siterm_assemble) of one large synthetic family per rank:
    2048 cherries x 512 sites, 20 states, 129 buckets -> the [512,129,20,20] float64 count tensor
    (211 MB), left on the device.  A step is one whole assembly (clear + count + pseudocount mix)."""
    import torch
    import torch.distributed as dist
    from cherryml_amd._siterm._assembly import _assemble, get_count_prior_probability_matrices
    from cherryml_amd.counting._stage import PAIR_DTYPE
    rng = np.random.default_rng(200 + rank)
    n, L, S = 2048, 512, 20
    grid = quantization_grid()
    Q = lg_matrix()
    prior = get_count_prior_probability_matrices(Q, list(grid))
    codes = rng.integers(0, 20, size=(2 * n, L)).astype(np.int8)
    codes[rng.random(codes.shape) < 0.05] = -1
    pairs = np.zeros(n, dtype=PAIR_DTYPE)
    lengths = rng.exponential(0.4, size=n)
    for k in range(n):
        pairs[k] = (2 * k * L, (2 * k + 1) * L, 0, L, 0, lengths[k], 0.0)
    rates = rng.gamma(3.0, 1.0 / 3.0, size=L)
    prof = {}
    call = lambda: _assemble(pairs, codes, grid, rates, prior, 0.5, True, S, local_rank, True, profile=prof)  # noqa: E731
    for _ in range(warmup):
        call()
    fence()
    kms = []
    for _ in range(steps):
        out_t = call()
        kms.append(prof["kernel_ms"])
    fence()
    kernel_ms = float(np.mean(kms))
    if world > 1:
        tdt = torch.tensor([kernel_ms], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        kernel_ms = float(tdt.item())
    if rank != 0:
        return None
    valid = int(((codes[0::2] >= 0) & (codes[1::2] >= 0)).sum())
    nbytes = float(out_t.numel() * 8 + codes.nbytes)        # the tensor written once + the codes read once
    out = {
        "metric": "cherry-pairs/sec (whole node): cherry x site transitions assembled into SiteRM count tensors",
        "value": valid * world / (kernel_ms * 1e-3), "unit": "cherry-pairs/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": kernel_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8/f64", "data": "synthetic",
        "config": {"workload": f"SiteRM assembly: {n} cherries x {L} sites per GPU, 20 states, 129 buckets, lambda 0.5",
                   "sharding": f"families x{world} (no collective)"},
        "roofline": {"bound": "hbm", "kernel": "memset + siterm_raw_counts_kernel + siterm_mix_kernel", "achieved":
                     nbytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": nbytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "bytes_per_step": nbytes,
                     "note": "the [L,B,S,S] tensor is cleared (1 write) and its live (site, bucket) matrices rewritten"},
    }
    if with_cpu:
        from oracle import siterm_assembly_oracle as sa
        sub = 64
        aa = "ARNDCQEGHILKMFPSTWYV"
        dec = lambda row: "".join(aa[c] if c >= 0 else "-" for c in row)  # noqa: E731
        tr = [(dec(codes[2 * k]), dec(codes[2 * k + 1]), float(lengths[k])) for k in range(sub)]
        t0 = time.perf_counter()
        sa.raw_count_matrices(tr, list(grid), list(aa), True)
        dtc = (time.perf_counter() - t0) * (n / sub)
        out["cpu_baseline"] = dict(value=valid / dtc, unit="cherry-pairs/s", cores=1, kind="port",
                                   sample=f"oracle raw_count_matrices (the reference's Python loop) on {sub} of {n} cherries, "
                                          "scaled; pseudocount mix not included", seconds_per_step=dtc)
    return out


def _pmc_traffic():
    """HBM-side bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    corrected as MI355X_MICROARCH.md prescribes), committed under profiles/."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f).get("bytes_per_launch", {})


def _traffic_source():
    """`roofline.traffic` is NOT measured by the run that prints the line (a counter pass cannot share a run with the
    timing): it is read from the committed PMC summary; this says which one."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    return (f"profiles/pmc_traffic.json ({d.get('round', 'r02')}, collected {d.get('collected', '2026-10-03')}; "
            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes), not measured in this run")


def _siterm_compute_floor():
    """sp_bank's float64-pipe occupancy from its tracked SQ counters (profiles/r06_sp_bank_sq_counters.json, collected by
    profiles/tools/r6_sq_counters_sp_bank.sh; not measured in this run)"""
    path = os.path.join(ROOT, "profiles", "r06_sp_bank_sq_counters.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f).get("derived", {}).get("sp_bank<5, true, true>")
    if not d:
        return None
    return {"kernel": "sp_bank<5, true, true>", "fp64_pipe_frac": round(d["fp64_pipe_frac"], 3),
            "mfma_pipe_busy_frac": round(d["mfma_pipe_busy_frac"], 3), "valu_issue_frac": round(d["valu_issue_frac"], 3),
            "floor_ms": round(d["compute_floor_ms"], 3), "measured_ms": round(d["avg_duration_us"] * 1e-3, 3),
            "resident_waves_per_simd": round(d["mean_resident_waves_per_simd"], 2),
            "source": "profiles/r06_sp_bank_sq_counters.json (rocprofv3 --pmc SQ_*, three passes; not measured in this run)"}


def _mfma_util(dtype):
    """measured MFMA-pipe utilisation of K1 / K2 / K3 (SQ_VALU_MFMA_BUSY_CYCLES, profiles/mfma_util.json) per kernel"""
    path = os.path.join(ROOT, "profiles", "mfma_util.json")
    if not os.path.exists(path):
        return {}, None
    with open(path) as f:
        d = json.load(f)
    out = {}
    for key, v in d.get("kernels", {}).items():
        name, dt = key.split(":")[:2]   # kernel : arithmetic [: bank form]
        if dt == dtype:
            form = key.split(":")[2] if key.count(":") >= 2 else ""
            out[name.split("_")[0] + (":" + form if form else "")] = round(v["mfma_util"], 3)
    return out, f"profiles/mfma_util.json ({d.get('round', 'r02')}, collected {d.get('collected', '2026-10-03')})"


def _bench_tree(rng, n_leaves):
    """Random binary tree by successive joins of random pairs of open subtrees."""
    from cherryml_amd.io import Tree
    tree = Tree()
    names = [f"l{i}" for i in range(n_leaves)]
    tree.add_nodes(names)
    pool, k = list(names), 0
    while len(pool) > 1:
        i, j = sorted(rng.choice(len(pool), size=2, replace=False), reverse=True)
        a, b = pool.pop(i), pool.pop(j)
        v = f"i{k}"
        k += 1
        tree.add_node(v)
        tree.add_edge(v, a, float(rng.uniform(0.01, 0.6)))
        tree.add_edge(v, b, float(rng.uniform(0.01, 0.6)))
        pool.append(v)
    return tree, names


def run_likelihood(steps, warmup, world, rank, local_rank, fence, with_cpu):
    """Held-out log-likelihood (cb_tree_likelihood) of one synthetic family per rank: 1024 leaves (2047
    nodes), 512 sites = 256 independent sites (LG, 20 rate categories) + 128 contacting pairs
    (400-state coevolution model).  A step is one whole `dp_likelihood_computation`."""
    import torch
    import torch.distributed as dist
    from cherryml_amd.evaluation import dp_likelihood_computation
    rng = np.random.default_rng(300 + rank)
    n_leaves, n_single, n_pairs = 1024, 256, 128
    L = n_single + 2 * n_pairs
    aa = list("ARNDCQEGHILKMFPSTWYV")
    lg = lg_matrix()
    pi1 = stationary(lg)
    Q2, pi2, _ = coevolution_truth(rng)
    Q2 = Q2 / -(pi2 * np.diag(Q2)).sum() * 2.0
    tree, names = _bench_tree(rng, n_leaves)
    alphabet = np.array(aa + ["-"])
    msa = {n: "".join(rng.choice(alphabet, size=L, p=np.r_[np.full(20, 0.0475), 0.05])) for n in names}
    cm = np.zeros((L, L), dtype=int)
    for k in range(n_pairs):
        i, j = n_single + 2 * k, n_single + 2 * k + 1
        cm[i, j] = cm[j, i] = 1
    rates = list(rng.choice(np.geomspace(0.05, 8.0, 20), size=L))
    prof = {}

    # the caller's view (VERDICT r5 "missing 3"): both models resident on the device (LikelihoodModel = cb_tl_model_create /
    # cb_tl_model_run), one call per family; ms_per_step is the WALL time of a call -- the family's tree arrays and state codes
    # built and uploaded, the transition banks, the pruning, the per-site results read back
    from cherryml_amd.evaluation import LikelihoodModel
    model_1 = LikelihoodModel(lg, pi1, pairs=False, device=local_rank)
    model_2 = LikelihoodModel(Q2, pi2, pairs=True, alphabet_size=20, device=local_rank)

    def call():
        prof.clear()
        return dp_likelihood_computation(tree, msa, cm, rates, aa, pi1, lg, pi_2=pi2, Q_2=Q2, device=local_rank,
                                         profile=prof, model_1=model_1, model_2=model_2)
    t0 = time.perf_counter()    # (the per-call entry: models made, used and freed by the call)
    dp_likelihood_computation(tree, msa, cm, rates, aa, pi1, lg, pi_2=pi2, Q_2=Q2, device=local_rank)
    per_call_entry_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(max(warmup, 1)):
        call()
    fence()
    kms, pms, sms = [], [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        ll, lls = call()
        kms.append(prof["kernel_ms"])
        pms.append(prof["prune_ms_pairs"])
        sms.append(prof["prune_ms_sites"])
    fence()
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean(kms))
    call_ms = dt / steps * 1e3
    model_1.close()
    model_2.close()
    if world > 1:
        tdt = torch.tensor([kernel_ms, call_ms], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        kernel_ms, call_ms = (float(v) for v in tdt.tolist())
    if rank != 0:
        return None
    n_nodes = 2 * n_leaves - 1
    flops = 2.0 * 400 * 400 * n_pairs * (n_nodes - 1)          # P_v W over all non-root nodes
    prune_pairs = float(np.mean(pms))
    ach = flops / (prune_pairs * 1e-3) / 1e12
    out = {
        "metric": "sites/sec (whole node): held-out log-likelihood, sites evaluated per second",
        "value": L * world / (call_ms * 1e-3), "unit": "sites/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": call_ms, "kernel_ms_per_step": kernel_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"held-out log-likelihood: {n_leaves} leaves ({n_nodes} nodes), {n_single} independent "
                               f"sites (LG, 20 rate categories) + {n_pairs} contacting pairs (400 states) per GPU",
                   "sharding": f"families x{world} (no collective)", "log_likelihood": ll,
                   "prune_ms_pairs": prune_pairs, "prune_ms_sites": float(np.mean(sms)),
                   "bank_ms": kernel_ms - prune_pairs - float(np.mean(sms)),
                   "entry": "dp_likelihood_computation on two resident LikelihoodModels (cb_tl_model_run); ms_per_step = wall time "
                            "of a call, kernel_ms_per_step = expm banks + pruning (HIP events)",
                   "per_call_entry_ms": per_call_entry_ms},
        "roofline": {"bound": "mfma", "kernel": "tl_leaf_mfma_kernel + tl_mfma_kernel (all heights of the tree)", "achieved": ach,
                     "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F64_PEAK_TFLOPS, "traffic": None,
                     "flops_per_step": flops,
                     "note": "ALGORITHMIC 2 S^2 flops per (non-root node, pair) / time of all pruning launches of the pair "
                             "model; a leaf's message is one [S x S] x [S x pairs] product with the model's eigenvectors as the "
                             "streamed operand (tl_leaf_mfma_kernel: no P_v is formed for the leaves -- half of the nodes), internal "
                             "nodes read their P_v (1.28 MB) once per 32-pair block (tl_mfma_kernel); the transition bank "
                             "(config.bank_ms) holds the internal nodes only"},
    }
    if with_cpu:
        from oracle import likelihood_oracle as lo
        srng = np.random.default_rng(7)
        s_leaves, s_single, s_pairs = 64, 32, 8
        stree, snames = _bench_tree(srng, s_leaves)
        sL = s_single + 2 * s_pairs
        smsa = {n: msa[f"l{i}"][:s_single] + msa[f"l{i}"][n_single:n_single + 2 * s_pairs] for i, n in enumerate(snames)}
        scm = np.zeros((sL, sL), dtype=int)
        for k in range(s_pairs):
            scm[s_single + 2 * k, s_single + 2 * k + 1] = scm[s_single + 2 * k + 1, s_single + 2 * k] = 1
        srates = rates[:s_single] + rates[n_single:n_single + 2 * s_pairs]
        t0 = time.perf_counter()
        lo.log_likelihood(stree, smsa, scm, srates, aa, pi1, lg, pi2, Q2)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=sL / dtc * ((2 * s_leaves - 1) / n_nodes), unit="sites/s", cores=1, kind="port",
                                   sample=f"the oracle on a {s_leaves}-leaf tree, {s_single} sites + {s_pairs} pairs of the "
                                          f"same alignment ({dtc:.1f} s); rate scaled by the node ratio {2 * s_leaves - 1}/"
                                          f"{n_nodes} (cost is linear in nodes)")
    return out


if __name__ == "__main__":
    main()
