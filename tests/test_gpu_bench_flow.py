"""The N > 1 control flow of bench.py on ONE GPU: `CB_BENCH_BACKEND=gloo` (a test hook; bench.py marks such a line in its
`data` field) lets two ranks share the device, which RCCL refuses.  Everything but the RCCL transport runs for real: the
launcher starts two rank processes, the bank is sharded over them (family counts -> collective over the non-empty buckets ->
per-epoch all-reduce of (loss, dL/dA)), the raw RCCL communicator cannot be made on a duplicated GPU and EVERY rank falls back
to torch's collective together (the failure protocol of DESIGN.md section 7), the weak-scaling secondaries run with their
collectives on device tensors, and rank 0's JSON line is relayed.  A real N > 1 run over RCCL / xGMI is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_the_gpu_through_the_gloo_hook():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CB_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "TEST HOOK" in d["data"]
    assert "buckets x2" in d["config"]["sharding"] and d["amdahl"]["measured_on_ranks"] == 2
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]
    # two ranks on one device: no raw RCCL communicator -> both ranks chose torch's collective together
    assert "torch glue" in r.stderr and "Adam in torch" in d["config"]["epoch"]
    sr, cc = d["secondary_siterm"], d["secondary_co_counting"]
    assert sr["n_gpus"] == 2 and sr["scaling"] == "weak" and "sites x2" in sr["config"]["sharding"]
    assert cc["n_gpus"] == 2 and cc["scaling"] == "weak" and cc["phase_ms"]["allreduce"] > 0
    # weak scaling: both ranks' events are in the counted total (10,000 families x 64 cherries x ~65 contacts each)
    assert cc["value"] * cc["ms_per_step"] * 1e-3 > 6.0e7
    assert "secondary" not in d      # the LG replica line belongs to N = 1
    # two ranks on ONE device = two persistent bank launches (k123_bank) sharing it: each must make progress with whatever
    # share of the slots it gets (every ticket drawn or claimed by the workgroup that runs it).  The first fused version, which
    # handed tickets to workgroups by index, took 7.9 SECONDS per epoch here -- the two launches waited for each other's
    # non-resident workgroups until the scheduler's preemption timer let them through.
    assert d["ms_per_step"] < 100.0, d["ms_per_step"]
    # round 4: the N > 1 line is self-diagnosing -- which transport carried the per-epoch all-reduce (None: torch's, because
    # the raw RCCL communicator could not be made here), per-phase maxima over the ranks, and every widened row of SURVEY 8f
    assert "rccl_ranks" in d and d["rccl_ranks"] is None
    mx = d["phase_ms_max_over_ranks"]
    assert all(mx[k] >= d["phase_ms"][k] - 1e-9 for k in ("eigh", "k1", "k2", "k3")) and "allreduce" in mx
    for name in ("secondary_counting", "secondary_ble", "secondary_assembly", "secondary_likelihood"):
        assert d[name]["n_gpus"] == 2 and d[name]["value"] > 0 and d[name]["roofline"]["bound"] in ("hbm", "mfma", "lds_atomic", "l2_gather"), name
    assert d["secondary_counting"]["roofline"]["bound"] == "lds_atomic" and d["secondary_ble"]["roofline"]["bound"] == "l2_gather"


def test_shard_of_line_measures_one_ranks_share_on_one_gpu():
    """`bench.py --shard-of 8`: rank 0's buckets of an 8-way deal of the bench bank (17 of 129) + the whole replicated part,
    through the C-driven sharded loop with a single-rank RCCL communicator -- what ONE GPU of eight does per epoch."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shard-of", "8", "--steps", "6", "--warmup", "2",
                        "--no-secondary"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["shard_of"] == 8 and d["config"]["share_buckets"] == 17
    assert d["rccl_ranks"] == 1 and "in-library ncclAllReduce" in d["config"]["epoch"]
    assert d["amdahl"]["measured_rank_ms"] == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert 0 < d["ms_per_step"] < 5.0 and d["final_loss"] == d["final_loss"]
    assert "cpu_baseline" not in d      # a share of a bank has no CPU twin
    # the share's bank launch is a sixth of the whole bank's work or less: well under the 0.66 ms of the 129-bucket launch
    bank_ms = d["phase_ms"]["k1"] + d["phase_ms"]["k2"] + d["phase_ms"]["k3"]
    assert 0 < bank_ms < 0.4, d["phase_ms"]


_RESIDENT_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from cherryml_amd.estimation_end_to_end import coevolution_fit_resident
z = np.load(sys.argv[2], allow_pickle=True)
r = coevolution_fit_resident(tree_dir=str(z["tree"]), msa_dir=str(z["msa"]), contact_map_dir=str(z["cm"]),
                             families=[str(f) for f in z["families"]], amino_acids=list("ARNDCQEGHILKMFPSTWYV"),
                             quantization_points=[float(q) for q in z["grid"]], edge_or_cherry="cherry++",
                             minimum_distance_for_nontrivial_contact=7, mask=z["mask"], num_epochs=int(z["epochs"]))
np.savez(sys.argv[3] + f".{rank}.npz", Q_best=r["Q_best"], loss=r["loss"], init=r["initialization"], n_pairs=r["n_pairs"])
if world > 1:
    dist.destroy_process_group()
'''


def test_resident_chain_two_ranks_equal_one_rank(tmp_path):
    """`coevolution_fit_resident` with the demo families dealt over TWO ranks (gloo on the one GPU: each rank counts its
    families, the JTT-IPW start comes from the all-reduced S x S sums, the counts meet in the collective over the non-empty
    buckets, no raw RCCL communicator -> torch's collective) gives the one-rank result: same pair total and initialiser,
    loss curves and learned Q to rounding."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden, relerr
    from test_gpu_demo_e2e import _materialise
    from cherryml_amd import caching
    from cherryml_amd.estimation_end_to_end import create_maximal_matching_contact_map
    z = load_golden("demo_e2e.npz")
    dirs, fams = _materialise(tmp_path, z)
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        cm_dir = create_maximal_matching_contact_map(i_contact_map_dir=dirs["contact_map"], families=fams,
                                                     minimum_distance_for_nontrivial_contact=7, num_processes=1)["o_contact_map_dir"]
    finally:
        caching.set_cache_dir(None)
    mask = np.unpackbits(z["co_mask_packed"])[:160000].reshape(400, 400).astype(np.float64)
    spec = tmp_path / "spec.npz"
    np.savez(spec, tree=dirs["tree"], msa=dirs["msa"], cm=cm_dir, families=np.array(fams), grid=z["quantization_points"].astype(float),
             mask=mask, epochs=int(z["co_epochs"]))
    script = tmp_path / "worker.py"
    script.write_text(_RESIDENT_WORKER)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    # one rank holds all live buckets, each of two ranks half of them: large_eval would sum the buckets before the last product
    # on the one (>= 24 buckets) and not on the two.  The comparison is about the sharding, so both keep the per-bucket form.
    base.update(CB_TEST_HOOKS="1", CB_BANK_K3="1")
    subprocess.run([sys.executable, str(script), ROOT, str(spec), str(tmp_path / "one")], env=dict(base, RANK="0", WORLD_SIZE="1"),
                   check=True, timeout=600)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(spec), str(tmp_path / "two")],
                              env=dict(base, RANK=str(r), WORLD_SIZE="2")) for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    one = np.load(str(tmp_path / "one") + ".0.npz")
    for r in range(2):
        two = np.load(str(tmp_path / "two") + f".{r}.npz")
        assert float(two["n_pairs"]) == float(one["n_pairs"]) == float(z["co_counts_val"].sum())
        assert np.allclose(two["init"], one["init"], rtol=1e-12, atol=1e-300)
        assert np.allclose(two["loss"], one["loss"], rtol=1e-10, atol=0)
        assert relerr(two["Q_best"], one["Q_best"]) < 1e-8
    assert relerr(one["Q_best"], z["co_Q_best_f64"]) < 1e-6


_INLIB_WORKER = r'''
import ctypes as C, glob, os, sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from cherryml_amd import CherryBank
from cherryml_amd.distributed import ShardedBank
z = np.load(sys.argv[2])
hip = C.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*"))[0])
calls = []

@C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
def allreduce(send, recv, count, dtype, op, comm, stream):
    """ncclAllReduce's signature; the transport is gloo through the host (two ranks share the GPU: RCCL cannot)."""
    assert dtype == 8 and op == 0, (dtype, op)          # ncclFloat64, ncclSum
    buf = np.empty(count)
    assert hip.hipStreamSynchronize(C.c_void_p(stream)) == 0
    assert hip.hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(send), C.c_size_t(count * 8), 2) == 0
    t = torch.from_numpy(buf)
    dist.all_reduce(t)
    assert hip.hipMemcpy(C.c_void_p(recv), C.c_void_p(buf.ctypes.data), C.c_size_t(count * 8), 1) == 0
    calls.append(int(count))
    return 0

C_rank = torch.tensor(z["C"] * (0.3 if rank == 0 else 0.7), device="cuda")   # this rank's share of every family count
sb = ShardedBank.from_rank_counts(z["t"], C_rank)
class _NoComm:                       # stands for the RcclCommunicator the bank would own
    def destroy(self): pass
sb.rccl = _NoComm()
sb.bank.allreduce_setup(1, C.cast(allreduce, C.c_void_p).value, [sb.total_count])
E = int(z["E"])
if os.environ.get("FAULT_RANK") == str(rank):
    os.environ["CB_FAULT_INJECT"] = "3"      # this rank's evaluation "fails" at epoch 3 (read by the library at the call)
try:
    r = sb.train_pande_reversible(z["u0"], z["p0"], mask=z["mask"], num_epochs=E, lr=0.1)
    form, info = sb.bank.last_bank_form(), sb.bank.time_basis_info()
    np.savez(sys.argv[3] + f".{rank}.npz", loss=r["loss"], Q_last=r["Q_last"], Q_best=r["Q_best"], calls=np.array(calls),
             local=np.array(sb.local_buckets), time_basis=int(form["time_basis"]), builds=info["builds"],
             repeated=info["repeated_epochs"])
except Exception as exc:
    np.savez(sys.argv[3] + f".{rank}.npz", error=str(exc), calls=np.array(calls))
sb.close()
dist.destroy_process_group()
'''


def test_c_driven_sharded_loop_with_two_real_ranks(tmp_path):
    """`cb_allreduce_setup` + `cb_train_pande_reversible` (S > 32): the WHOLE sharded epoch loop from C on two ranks that
    really hold different shares of the counts and different buckets (gloo processes on the one GPU; the function handed
    over in place of `ncclAllReduce` has its signature and moves the bytes through gloo).  Both ranks reproduce the
    single-process trajectory of the full bank; every epoch made exactly two collective calls (1 and LD^2 doubles) after
    the set-up's two."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden, relerr
    from cherryml_amd import CherryBank
    import bench
    z = load_golden("coevo_dense_traj.npz")
    rng = np.random.default_rng(0)
    wl = bench.make_workload("coevo400", 0, rng)
    sel = z["sel"]
    t, Cm, mask = wl["t"][sel], wl["C"][sel], wl["mask"]
    E = 8
    with CherryBank(t, Cm) as bank:
        one = bank.train_pande_reversible(z["upper_diag0"], z["log_pi0"], mask=mask, num_epochs=E, lr=0.1)
    spec = tmp_path / "spec.npz"
    np.savez(spec, t=t, C=Cm, mask=mask, u0=z["upper_diag0"], p0=z["log_pi0"], E=E)
    script = tmp_path / "worker.py"
    script.write_text(_INLIB_WORKER)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(spec), str(tmp_path / "two")],
                              env=dict(base, RANK=str(r), WORLD_SIZE="2")) for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    got = [np.load(str(tmp_path / "two") + f".{r}.npz") for r in range(2)]
    assert set(got[0]["local"]).isdisjoint(got[1]["local"]) and len(got[0]["local"]) + len(got[1]["local"]) == len(sel)
    for g in got:
        assert np.allclose(g["loss"], one["loss"], rtol=1e-11, atol=0)
        assert relerr(g["Q_last"], one["Q_last"]) < 1e-9 and relerr(g["Q_best"], one["Q_best"]) < 1e-9
        calls = list(g["calls"])
        assert calls[-2 * E:] == [1, 400 * 400] * E, calls
    assert np.array_equal(got[0]["Q_last"], got[1]["Q_last"])      # identical steps on both ranks


@pytest.mark.parametrize("growth_hook", [False, True])
def test_two_real_ranks_each_in_its_own_time_basis(tmp_path, growth_hook):
    """the ranks of a sharded job run THEIR buckets in a time basis of their own since round 5 (every second bucket of
    an ascending grid is one) and nothing tested a communicator and a time basis together.  The bench bank's 129 buckets over two
    real ranks (65 / 64, both above the 28-bucket threshold): both ranks report the time-basis form, reproduce the single-process
    trajectory (which runs the whole bank in ONE basis: other skeletons, other sums, hence rounding) and take identical steps.
    With CB_TB_TEST_GROWTH (a basis with no room to grow) the device-side guard fires on both ranks in the same epochs -- sigma is
    the same everywhere --, the epochs are repeated with per-bucket products, and the collectives still pair up."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden, relerr
    from cherryml_amd import CherryBank
    import bench
    z = load_golden("coevo_dense_traj_full.npz")
    wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
    t, Cm, mask = wl["t"], wl["C"], wl["mask"]
    E = 8
    with CherryBank(t, Cm) as bank:
        one = bank.train_pande_reversible(z["upper_diag0"], z["log_pi0"], mask=mask, num_epochs=E, lr=0.1)
        assert bank.last_bank_form()["time_basis"]
    spec = tmp_path / "spec.npz"
    np.savez(spec, t=t, C=Cm, mask=mask, u0=z["upper_diag0"], p0=z["log_pi0"], E=E)
    script = tmp_path / "worker.py"
    script.write_text(_INLIB_WORKER)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CB_TB_TEST_GROWTH")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29581" if growth_hook else "29580", CB_TEST_HOOKS="1")
    if growth_hook:
        base["CB_TB_TEST_GROWTH"] = "1.0005"
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(spec), str(tmp_path / "tb")],
                              env=dict(base, RANK=str(r), WORLD_SIZE="2")) for r in range(2)]
    assert [p.wait(timeout=900) for p in procs] == [0, 0]
    got = [np.load(str(tmp_path / "tb") + f".{r}.npz") for r in range(2)]
    assert "error" not in got[0].files and "error" not in got[1].files, [str(g["error"]) for g in got if "error" in g.files]
    assert sorted(len(g["local"]) for g in got) == [64, 65]
    for g in got:
        # (with the hook the LAST evaluation may be a repeated one, i.e. per-bucket products: the builds say the basis was in use)
        assert int(g["builds"]) >= (2 if growth_hook else 1) and (growth_hook or int(g["time_basis"]) == 1)
        assert (int(g["repeated"]) >= 2) if growth_hook else (int(g["repeated"]) == 0)
        assert np.allclose(g["loss"], one["loss"], rtol=1e-10, atol=0)
        assert relerr(g["Q_last"], one["Q_last"]) < 1e-8 and relerr(g["Q_best"], one["Q_best"]) < 1e-8
    assert int(got[0]["repeated"]) == int(got[1]["repeated"])
    assert np.array_equal(got[0]["Q_last"], got[1]["Q_last"]) and list(got[0]["calls"]) == list(got[1]["calls"])
    print(f"two ranks in their own time bases ({'growth hook' if growth_hook else 'default'}): loss to the single process "
          f"{np.max(np.abs(got[0]['loss'] - one['loss']) / np.abs(one['loss'])):.1e}, Q_last {relerr(got[0]['Q_last'], one['Q_last']):.1e}, "
          f"repeated epochs {int(got[0]['repeated'])}")


def test_a_failing_rank_takes_its_peer_down_after_the_same_collectives(tmp_path):
    """The failure protocol of the C-driven loop with two real ranks: rank 1's evaluation fails at epoch 3
    (CB_FAULT_INJECT); it keeps its place in every remaining all-reduce with NaN payloads, so rank 0's parameters turn
    NaN, its own eigensolver refuses them, and BOTH ranks return an error -- after the same number of collective calls,
    nobody left waiting."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    import bench
    z = load_golden("coevo_dense_traj.npz")
    wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
    sel = z["sel"]
    E = 8
    spec = tmp_path / "spec.npz"
    np.savez(spec, t=wl["t"][sel], C=wl["C"][sel], mask=wl["mask"], u0=z["upper_diag0"], p0=z["log_pi0"], E=E)
    script = tmp_path / "worker.py"
    script.write_text(_INLIB_WORKER)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CB_FAULT_INJECT")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", FAULT_RANK="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(spec), str(tmp_path / "f")],
                              env=dict(base, RANK=str(r), WORLD_SIZE="2")) for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    got = [np.load(str(tmp_path / "f") + f".{r}.npz") for r in range(2)]
    assert "error" in got[0].files and "error" in got[1].files, [g.files for g in got]
    assert "injected fault" in str(got[1]["error"]) and "NaN" in str(got[1]["error"])
    assert len(got[0]["calls"]) == len(got[1]["calls"])          # same number of collectives on both ranks


_COUNT_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
import cherryml_amd
z = np.load(sys.argv[2], allow_pickle=True)
fams = [str(f) for f in z["families"]]
AA = list("ARNDCQEGHILKMFPSTWYV")
cherryml_amd.count_transitions(tree_dir=str(z["tree"]), msa_dir=str(z["msa"]), site_rates_dir=str(z["rates"]), families=fams,
                               amino_acids=AA, quantization_points=[str(q) for q in z["grid"]], edge_or_cherry="cherry++",
                               output_count_matrices_dir=sys.argv[3] + "/single")
cherryml_amd.count_co_transitions(tree_dir=str(z["tree"]), msa_dir=str(z["msa"]), contact_map_dir=str(z["cm"]), families=fams,
                                  amino_acids=AA, quantization_points=[str(q) for q in z["grid"]], edge_or_cherry="cherry++",
                                  minimum_distance_for_nontrivial_contact=7, output_count_matrices_dir=sys.argv[3] + "/co")
dist.destroy_process_group()
'''


def test_counting_stages_with_families_dealt_over_two_ranks(tmp_path):
    """`count_transitions` / `count_co_transitions` under torch.distributed: families dealt round-robin to two ranks (the
    reference's MPI scheme, _count_transitions.cpp:626-628), each rank counts its own on the GPU, the integer counts are
    all-reduced, rank 0 writes -- the files equal the reference's counts of the demo families bit for bit."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    from test_gpu_demo_e2e import _materialise
    from cherryml_amd import caching
    from cherryml_amd.estimation_end_to_end import create_maximal_matching_contact_map
    from cherryml_amd.io import read_count_matrices_arrays
    z = load_golden("demo_e2e.npz")
    dirs, fams = _materialise(tmp_path, z)
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        cm_dir = create_maximal_matching_contact_map(i_contact_map_dir=dirs["contact_map"], families=fams,
                                                     minimum_distance_for_nontrivial_contact=7, num_processes=1)["o_contact_map_dir"]
    finally:
        caching.set_cache_dir(None)
    spec = tmp_path / "spec.npz"
    np.savez(spec, tree=dirs["tree"], msa=dirs["msa"], rates=dirs["site_rates"], cm=cm_dir, families=np.array(fams),
             grid=np.array([str(q) for q in z["quantization_points"]]))
    script = tmp_path / "worker.py"
    script.write_text(_COUNT_WORKER)
    out = tmp_path / "out"
    out.mkdir()
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29580")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(spec), str(out)], env=dict(base, RANK=str(r), WORLD_SIZE="2"))
             for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    q, C, _ = read_count_matrices_arrays(str(out / "single" / "result.txt"))
    assert np.array_equal(q, z["lg_t"]) and np.array_equal(C, z["lg_counts"])
    q, C, _ = read_count_matrices_arrays(str(out / "co" / "result.txt"))
    ref = np.zeros_like(C)
    ref[tuple(z["co_counts_nz"].T)] = z["co_counts_val"]
    assert np.array_equal(q, z["co_t"]) and np.array_equal(C, ref)
