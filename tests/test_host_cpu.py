"""CPU-side tests: wire formats, caching convention, parameterisation mirror
vs the oracle, C-ABI symbol table, loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import ROOT, load_golden, relerr
from oracle import ratelearn_oracle as orc

import cherryml_amd
from cherryml_amd import _lib, caching
from cherryml_amd.estimation._ratelearn._rate_matrix import RateMatrix
from cherryml_amd.io import (read_count_matrices, read_count_matrices_arrays, read_mask_matrix,
                             read_rate_matrix, write_count_matrices, write_rate_matrix)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cherrybank.h")).read()
    declared = set(re.findall(r"\b(cb_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(os.path.join(ROOT, "cherryml_amd", "libcherrybank.so"))
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().cb_version() >= 1


def test_no_gpu_means_loud_failure_not_fallback():
    if _lib.load().cb_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.CherryBankError):
        cherryml_amd.CherryBank(np.ones(2), np.ones((2, 3, 3)))


def test_no_gpu_fails_loudly_whatever_the_device_argument_says(tmp_path):
    """`device` keeps the reference's spellings and default ("cpu"), but there is ONE execution target:
    without an MI355X every entry point raises CherryBankError -- never a CPU fallback.  (With a GPU,
    "cpu" runs there and warns: tests/test_gpu_api.py.)"""
    from cherryml_amd import _lib
    if _lib.load().cb_device_count() > 0:
        pytest.skip("a GPU is visible")
    g = load_golden("eval_toy3_init.npz")
    states = list("ABC")
    cpath = str(tmp_path / "c.txt")
    write_count_matrices([(float(t), pd.DataFrame(C, index=states, columns=states))
                          for t, C in zip(g["t"], g["C"])], cpath)
    for device in ("cpu", "cuda"):
        with pytest.raises(_lib.CherryBankError, match="no CPU fallback"):
            cherryml_amd.quantized_transitions_mle(
                count_matrices_path=cpath, initialization_path=None, mask_path=None,
                output_rate_matrix_dir=str(tmp_path / "out"), device=device, num_epochs=1)
        with pytest.raises(_lib.CherryBankError, match="no CPU fallback"):
            cherryml_amd.quantized_transitions_mle_vectorized_over_sites(
                np.ones((1, 1, 4, 4)), np.ones((1, 1)), 1, device=device)
    with pytest.raises(ValueError):
        cherryml_amd.quantized_transitions_mle_vectorized_over_sites(
            np.ones((1, 1, 4, 4)), np.ones((1, 1)), 1, device="tpu")
    from cherryml_amd.estimation_end_to_end import coevolution_fit_resident
    with pytest.raises(_lib.CherryBankError, match="no CPU fallback"):   # the resident chain too
        coevolution_fit_resident(tree_dir=str(tmp_path), msa_dir=str(tmp_path), contact_map_dir=str(tmp_path), families=[],
                                 amino_acids=list("AC"), quantization_points=[0.1, 0.2], edge_or_cherry="cherry",
                                 minimum_distance_for_nontrivial_contact=3, num_epochs=1)


def test_positional_arguments_refused():
    with pytest.raises(caching.CacheUsageError):
        cherryml_amd.quantized_transitions_mle("a", None, None, "out")


def test_cached_computation_convention(tmp_path):
    calls = []

    @caching.cached_computation(output_dirs=["out_dir"], exclude_args=["device"])
    def stage(x: int, out_dir=None, device="cpu"):
        calls.append(x)
        open(os.path.join(out_dir, "result.txt"), "w").write(str(x))

    caching.set_cache_dir(str(tmp_path))
    try:
        r1 = stage(x=1)
        r2 = stage(x=1, device="cuda")  # excluded from the key -> cache hit
        r3 = stage(x=2)
    finally:
        caching.set_cache_dir(None)
    assert calls == [1, 2]
    assert r1 == r2 and r1 != r3
    assert os.path.exists(os.path.join(r1["out_dir"], "result.success"))
    assert open(os.path.join(r3["out_dir"], "result.txt")).read() == "2"


def test_count_matrix_roundtrip_and_reference_format(tmp_path):
    g = load_golden("eval_s20_mask.npz")
    states = [str(s) for s in load_golden("data_lg.npz")["states"]]
    path = str(tmp_path / "counts.txt")
    mats = [(float(t), pd.DataFrame(C, index=states, columns=states))
            for t, C in zip(g["t"], g["C"])]
    write_count_matrices(mats, path)
    q, C, st = read_count_matrices_arrays(path)
    assert st == states and np.array_equal(q, g["t"]) and np.array_equal(C, g["C"])
    back = read_count_matrices(path)
    assert back[1][0] == g["t"][1] and np.array_equal(back[1][1].to_numpy(), g["C"][1])
    # space separated variant with a leading blank header, as the reference's test files
    lines = open(path).read().replace("\t", " ")
    open(path, "w").write(lines)
    assert np.array_equal(read_count_matrices_arrays(path)[1], g["C"])
    # malformed header
    open(path, "w").write("2 matrixes\n20 states\n")
    with pytest.raises(Exception):
        read_count_matrices_arrays(path)


def test_rate_and_mask_matrix_roundtrip(tmp_path):
    lg = load_golden("data_lg.npz")
    states = [str(s) for s in lg["states"]]
    path = str(tmp_path / "d" / "lg.txt")
    write_rate_matrix(lg["lg"], states, path)
    back = read_rate_matrix(path)
    assert list(back.index) == states and np.array_equal(back.to_numpy(), lg["lg"])
    mpath = str(tmp_path / "mask.txt")
    m = (np.arange(400).reshape(20, 20) % 3 > 0).astype(int)
    pd.DataFrame(m, index=states, columns=states).to_csv(mpath, sep=" ")
    assert np.array_equal(read_mask_matrix(mpath).to_numpy(), m)


@pytest.mark.parametrize("case", ["toy3_init", "toy3_mask", "s20_symmask", "s20_mask"])
def test_rate_matrix_module_matches_reference_Q(case):
    """theta -> Q of the mirror == the reference's Q (golden) for the same params."""
    g = load_golden(f"eval_{case}.npz")
    S = g["mask"].shape[0]
    torch.manual_seed(0)
    mod = RateMatrix(num_states=S, mode="pande_reversible", mask=torch.tensor(g["mask"]),
                     pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True,
                     initialization=g["init"] if "init" in g else None)
    if "init" in g:  # parameters recovered from the initialisation (reference stores them f32)
        fin = np.isfinite(g["upper_diag"])
        assert np.allclose(mod.upper_diag.detach().numpy()[fin], g["upper_diag"][fin],
                           rtol=1e-6, atol=1e-6)
        assert np.array_equal(np.isfinite(mod.upper_diag.detach().numpy()), fin)
    else:  # the seed-0 float32 draw of the reference
        assert np.array_equal(mod.upper_diag.detach().numpy(), g["upper_diag"])
    with torch.no_grad():
        mod.upper_diag.copy_(torch.tensor(g["upper_diag"]))
        mod._pi.copy_(torch.tensor(g["log_pi"]))
    Q = mod()
    assert relerr(Q.detach().numpy(), g["Q_f64"]) < 1e-14
    assert mod.is_reversible() == (case != "s20_mask")
    # and the parameter gradients through the mirror equal the reference's, given dL/dQ
    Q.backward(torch.tensor(g["dQ_f64"]))
    fin = np.isfinite(g["upper_diag"])
    assert relerr(mod.upper_diag.grad.numpy()[fin], g["d_upper_f64"][fin]) < 1e-11
    assert relerr(mod._pi.grad.numpy(), g["d_log_pi_f64"]) < 1e-10


def test_rate_matrix_init_errors():
    g = load_golden("eval_toy3_init.npz")
    bad_mask = load_golden("eval_toy3_mask.npz")["mask"]
    with pytest.raises(ValueError):
        RateMatrix(num_states=3, mode="pande_reversible", mask=torch.tensor(bad_mask),
                   pi=torch.ones(3, dtype=torch.float64) / 3, initialization=g["init"])
    with pytest.raises(ValueError):
        RateMatrix(num_states=3, mode="default", mask=torch.ones(3, 3),
                   pi=torch.ones(3, dtype=torch.float64) / 3, initialization=g["init"])


def test_other_modes_have_zero_row_sums():
    torch.manual_seed(1)
    for mode in ["default", "pande", "stationary", "stationary_reversible"]:
        mod = RateMatrix(num_states=5, mode=mode, mask=torch.ones(5, 5),
                         pi=torch.tensor([0.1, 0.2, 0.3, 0.25, 0.15], dtype=torch.float64))
        Q = mod().detach().numpy()
        if mode in ("default", "pande"):
            assert np.abs(Q.sum(1)).max() < 1e-14
        else:  # "stationary*" modes return R diag(pi) with pi-weighted zero row sums
            pi = mod.stationary().detach().numpy()
            assert np.abs((Q / pi[None, :]) @ pi).max() < 1e-14


def test_jtt_ipw_reference_goldens(tmp_path):
    """reference tests/estimation_tests/jtt_ipw_test.py:12-74 (exact goldens): the closed form on top of the two S x S
    sums (host logic; the sums themselves are the GPU pass cb_jtt_ipw_stats, checked on the same goldens through the stage
    function in tests/test_gpu_api.py) and the oracle; without a GPU the stage fails loudly."""
    from cherryml_amd.estimation import jtt_ipw_from_statistics
    g = load_golden("jtt_ipw_toy.npz")
    t, C = g["t"], g["C"].astype(np.float64)
    Cs = 0.5 * (C + C.transpose(0, 2, 1))
    F, R = Cs.sum(0), (Cs / t[:, None, None]).sum(0)
    for key, m, ipw in [("Q1_JTT_IPW_on_toy_matrix", None, True),
                        ("Q1_JTT_IPW_on_toy_matrix_mask", g["mask"].astype(float), True),
                        ("Q1_JTT_on_toy_matrix", None, False),
                        ("Q1_JTT_on_toy_matrix_mask", g["mask"].astype(float), False)]:
        got = jtt_ipw_from_statistics(F, R, t, m, use_ipw=ipw)
        np.testing.assert_almost_equal(got, g[key], decimal=7)
        np.testing.assert_almost_equal(got, orc.jtt_ipw(t, C, m, use_ipw=ipw), decimal=12)
    if _lib.load().cb_device_count() <= 0:
        states = list("ABC")
        cpath = str(tmp_path / "c.txt")
        write_count_matrices([(float(tt), pd.DataFrame(Cb, index=states, columns=states)) for tt, Cb in zip(t, C)], cpath)
        os.makedirs(str(tmp_path / "o"))
        with pytest.raises(_lib.CherryBankError):
            cherryml_amd.jtt_ipw(count_matrices_path=cpath, mask_path=None, use_ipw=True,
                                 output_rate_matrix_dir=str(tmp_path / "o"))


def test_newick_conversion_and_standard_site_rate_grid():
    """convert_newick_to_CherryML_Tree names internal nodes in pre-order like the reference
    (io/_tree.py:266-320, whose own test expects `internal-2` for the inner node); the standard
    site-rate grid / prior of the SiteRM public API (_learn_site_rate_matrix.py:933-952)."""
    from cherryml_amd.io import convert_newick_to_CherryML_Tree
    from cherryml_amd._siterm import get_standard_site_rate_grid, get_standard_site_rate_prior
    t = convert_newick_to_CherryML_Tree("((Homo_sapiens:0.00655,Pan_troglodytes:0.00684):0.00422);")
    assert t.root() == "internal-1"
    assert t.edges() == [("internal-1", "internal-2", 0.00422), ("internal-2", "Homo_sapiens", 0.00655),
                         ("internal-2", "Pan_troglodytes", 0.00684)]
    t = convert_newick_to_CherryML_Tree("((a:1,'b c':2)90:0.5,(d,e)x:3)root;")
    assert t.leaves() == ["a", "b c", "d", "e"] and t.root() == "internal-1"
    assert t.children("internal-3") == [("d", 1.0), ("e", 1.0)]      # missing lengths are 1.0 (ete3's default)
    with pytest.raises(ValueError):
        convert_newick_to_CherryML_Tree("((a,b);")
    g = get_standard_site_rate_grid(20)
    assert len(g) == 20 and abs(g[0] - 0.05) < 1e-15 and abs(g[-1] - 20.0) < 1e-12
    assert all(g[i] < g[i + 1] for i in range(19))
    golden = load_golden("siterm_learn.npz")
    assert np.allclose(g, golden["pub_grid"], rtol=1e-15) and np.allclose(get_standard_site_rate_prior(20), golden["pub_prior"], rtol=1e-13)


def test_native_count_matrix_parser_equals_python_float(tmp_path, monkeypatch):
    """cb_parse_count_matrices (host threads, Clinger fast path + strtod) against the Python tokeniser it
    replaces: awkward numbers, ragged whitespace, and the reference's error cases."""
    import cherryml_amd.io._formats as F
    rng = np.random.default_rng(5)
    S, B = 7, 5
    states = ["A", "R", "N", "D", "gap-", "X1", "Zz"]
    specials = ["0.0", "0", "1e-05", "1.5e+16", "-3.25", "123456789012345678", "0.1234567890123456789", "4.9e-324",
                "1.7976931348623157e+308", "2.5E3", "+7", ".5", "5.", "1e22", "1e23", "9007199254740993", "0.30000000000000004",
                "1e-300", "123.456e-2", "inf", "nan"]
    toks = []
    path = tmp_path / "cm.txt"
    with open(path, "w") as f:
        f.write(f"{B} matrices\n{S} states\n")
        for b in range(B):
            f.write(f"{0.03 * 1.1 ** b!r}\n" if b != 2 else "  6.729e-05 \r\n")
            f.write("\t" + "\t".join(states) + ("\n" if b % 2 else "   \n\n"))
            for r in range(S):
                row = [specials[int(rng.integers(len(specials)))] if rng.random() < 0.5 else repr(float(rng.normal() * 10.0 ** int(rng.integers(-8, 8))))
                       for _ in range(S)]
                toks.append(row)
                f.write(states[r] + ("\t" if r % 2 else "  ") + ("\t" if b % 2 else " ").join(row) + "\n")
    q, C, st = F.read_count_matrices_arrays(str(path))
    want = np.array([[float(t) for t in row] for row in toks]).reshape(B, S, S)
    assert st == states and q[2] == 6.729e-05 and q[0] == 0.03
    assert np.array_equal(C, want, equal_nan=True)                    # bit-identical to Python's float()
    monkeypatch.setattr(F, "_parse_native", lambda *a: None)          # the pure-Python route gives the same
    q2, C2, st2 = F.read_count_matrices_arrays(str(path))
    assert np.array_equal(q, q2) and np.array_equal(C, C2, equal_nan=True) and st == st2
    monkeypatch.undo()
    text = open(path).read()
    bad = tmp_path / "bad.txt"
    bad.write_text(text.replace("\nX1", "\nX2", 1).replace("\tX1\t", "\tX1\t", 1))    # a row label of one matrix differs
    with pytest.raises(Exception, match="state labels"):
        F.read_count_matrices_arrays(str(bad))
    bad.write_text(text + " 1.0\n")                                                     # one token too many
    with pytest.raises(Exception, match="tokens"):
        F.read_count_matrices_arrays(str(bad))
    lines = text.split("\n")
    lines[5] = lines[5].replace(lines[5].split()[1], "12..5", 1)                        # not a number
    bad.write_text("\n".join(lines))
    with pytest.raises(Exception, match="not a number"):
        F.read_count_matrices_arrays(str(bad))


def test_native_formatter_writes_the_bytes_of_repr(tmp_path):
    """cb_format_matrix_rows (std::to_chars digits laid out by Python's rule) against
    "\\t".join(map(repr, row)) on random bit patterns, the notation boundaries (1e16, 1e-4), signed zeros,
    denormals and infinities; and the two writers that use it round-trip through the readers."""
    import cherryml_amd.io._formats as F
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2 ** 63, size=60000, dtype=np.int64).astype(np.uint64) | \
        (rng.integers(0, 2, size=60000).astype(np.uint64) << np.uint64(63))
    v = bits.view(np.float64)
    v = v[~np.isnan(v)]
    spec = np.array([0.0, -0.0, 1.0, 12.0, 1e16, 1e15, 9999999999999998.0, 1e-4, 1e-5, 0.00012, 123456.789, 5e-324,
                     1.7976931348623157e308, 2.5, 1e22, 1e23, -3.5e-7, float("inf"), float("-inf"), 100.0,
                     1234567890123456.0, 12345678901234567.0, 0.1, 0.30000000000000004, 99999999999999.98])
    decs = rng.normal(size=20000) * 10.0 ** rng.integers(-10, 10, size=20000)
    allv = np.concatenate([v, spec, decs, np.round(decs, 3), rng.integers(0, 1000, size=3000) * 0.25])
    M = allv[:(allv.size // 9) * 9].reshape(-1, 9)
    labels = [f"row {i}" if i % 3 else "Ä" for i in range(M.shape[0])]
    got = F._format_rows_native(M, labels)
    want = "".join(l + "\t" + "\t".join(map(repr, row)) + "\n" for l, row in zip(labels, M.tolist())).encode("utf-8")
    assert got == want
    states = ["A", "C", "G", "T"]
    Q = rng.normal(size=(4, 4)) * 1e-3
    F.write_rate_matrix(Q, states, str(tmp_path / "q.txt"))
    assert np.array_equal(F.read_rate_matrix(str(tmp_path / "q.txt")).to_numpy(), Q)
    assert open(tmp_path / "q.txt", "rb").read() == pd.DataFrame(Q, index=states, columns=states).to_csv(sep="\t").encode()
    cm = [(0.03, pd.DataFrame(rng.integers(0, 9, size=(4, 4)) * 0.5, index=states, columns=states)),
          (6.729e-05, pd.DataFrame(rng.random((4, 4)), index=states, columns=states))]
    F.write_count_matrices(cm, str(tmp_path / "c.txt"))
    q, C, st = F.read_count_matrices_arrays(str(tmp_path / "c.txt"))
    assert st == states and list(q) == [0.03, 6.729e-05]
    assert np.array_equal(C[0], cm[0][1].to_numpy()) and np.array_equal(C[1], cm[1][1].to_numpy())


def test_jtt_ipw_from_two_reduced_sums_equals_the_tensor_form():
    """The estimator is computed from sum_b sym(C_b) and sum_b sym(C_b) / t_b (the GPU pass; also what the ranks of the
    resident chain all-reduce instead of the 165 MB tensor): the same matrix as the reference's tensor form (the oracle),
    with and without a mask, symmetrised or not, both rate estimators."""
    from cherryml_amd.estimation import jtt_ipw_from_statistics
    from cherryml_amd.estimation_end_to_end import jtt_ipw_from_reduced_statistics
    rng = np.random.default_rng(3)
    B, S = 17, 12
    t = np.sort(rng.uniform(0.01, 3.0, B))
    C = rng.poisson(2.0, size=(B, S, S)).astype(np.float64)
    C[::4] = 0.0
    mask = (rng.random((S, S)) < 0.6).astype(np.float64)
    mask = np.maximum(mask, mask.T)
    np.fill_diagonal(mask, 1.0)
    Cs = 0.5 * (C + C.transpose(0, 2, 1))
    for m in (None, mask):
        want = orc.jtt_ipw(t, C, m)
        got = jtt_ipw_from_reduced_statistics(Cs.sum(0), (Cs / t[:, None, None]).sum(0), t, m)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-14)
        for ipw in (True, False):
            for sym in (True, False):
                X = Cs if sym else C
                got = jtt_ipw_from_statistics(X.sum(0), (X / t[:, None, None]).sum(0), t, m, use_ipw=ipw)
                assert np.allclose(got, orc.jtt_ipw(t, C, m, use_ipw=ipw, symmetrize=sym), rtol=1e-12, atol=1e-14)


def test_bank_ticket_protocol_finishes_with_any_number_of_resident_workgroups():
    """A model of the ticket protocol of the fused bank launch (csrc/large_bank.hip.h, k123_bank): per-queue counters, the
    reserved first tickets with claim flags, the per-bucket completion counters, deferred announcements and the `help` path.
    Only R of the launch's workgroups are ever resident (persistent workgroups leave only when no ticket is left, so the others
    never start -- two launches sharing a GPU); a random scheduler advances one resident workgroup at a time.  Every tile must
    run exactly once, after its inputs, and the launch must end: with tickets handed out by index (the first fused version)
    this model deadlocks as the GPU did."""
    import random

    def run(B, t1, t2, t3, grid, resident, seed, claimable=True):
        NQ = 8
        rng = random.Random(seed)
        per = t1 + t2 + t3
        nb = [B * (q + 1) // NQ - B * q // NQ for q in range(NQ)]
        b0 = [B * q // NQ for q in range(NQ)]
        qlen = [n * per for n in nb]
        homes = [(grid - q + NQ - 1) // NQ for q in range(NQ)]
        reserved = [min(homes[q], nb[q] * t1) for q in range(NQ)]
        tick = list(reserved)
        claim = [[0] * max(reserved[q], 1) for q in range(NQ)]
        done1, done2 = [0] * B, [0] * B
        ran = {}

        def decode(q, idx):
            n = nb[q]
            if idx < n * t1:
                return 0, b0[q] + idx // t1, idx % t1
            if idx < n * (t1 + t2):
                i = idx - n * t1
                return 1, b0[q] + i // t2, i % t2
            i = idx - n * (t1 + t2)
            return 2, b0[q] + i // t3, i % t3

        class WG:
            def __init__(self, bid):
                self.bid, self.home, self.first = bid, bid % NQ, True
                self.q, self.ticket, self.stash, self.pending, self.done = bid % NQ, None, None, None, False

            def flush(self):
                if self.pending is not None:
                    arr, b = self.pending
                    arr[b] += 1
                    self.pending = None

            def step(self):
                if self.ticket is None:
                    if self.stash is not None:
                        self.ticket, self.stash = self.stash, None
                    else:
                        idx, q = -1, self.q
                        if self.first:
                            i0 = self.bid // NQ
                            if i0 < reserved[self.home]:
                                if not claimable or claim[self.home][i0] == 0:
                                    claim[self.home][i0] = 1
                                    idx = i0
                            self.first = False
                        if idx < 0 and q == self.home:
                            idx, tick[q] = tick[q], tick[q] + 1
                        if idx >= qlen[q]:
                            idx = -1
                        tries = 0
                        while idx < 0 and tries < NQ:
                            q = (q + 1) % NQ
                            tries += 1
                            if tick[q] >= qlen[q]:
                                continue
                            i, tick[q] = tick[q], tick[q] + 1
                            if i < qlen[q]:
                                idx = i
                        if idx < 0:
                            self.flush()
                            self.done = True
                            return
                        self.q = q
                        self.ticket = decode(q, idx) + (q,)
                    return
                stage, b, tile, q = self.ticket
                ready = stage == 0 or (stage == 1 and done1[b] >= t1) or (stage == 2 and done2[b] >= t2)
                if not ready:
                    self.flush()
                    if stage == 1 and claimable:   # help: an unclaimed reserved K1 ticket of this bucket
                        bl = b - b0[q]
                        for i in range(bl * t1, (bl + 1) * t1):
                            if i < reserved[q] and claim[q][i] == 0:
                                claim[q][i] = 1
                                self.stash, self.ticket = self.ticket, (0, b, i - bl * t1, q)
                                return
                    return   # spin
                assert (stage, b, tile) not in ran
                if stage == 1:
                    assert done1[b] == t1
                if stage == 2:
                    assert done2[b] == t2
                ran[(stage, b, tile)] = True
                self.flush()   # (the device announces the previous tile from inside this one)
                self.pending = (done1, b) if stage == 0 else (done2, b) if stage == 1 else None
                self.ticket = None

        # without claim flags the reserved tickets belong to their workgroups by index, resident or not
        wgs = [WG(i) for i in range(grid)]
        running = wgs[:resident]
        waiting = wgs[resident:]
        for _ in range(400000):
            if not running:
                break
            w = rng.choice(running)
            w.step()
            if w.done:
                running.remove(w)
                if waiting:
                    running.append(waiting.pop(0))
        return not running and not waiting and len(ran) == B * per

    for B, t1, t2, t3, grid in ((129, 15, 25, 15, 1024), (9, 1, 1, 1, 27), (3, 3, 4, 3, 30), (20, 6, 9, 6, 420)):
        for resident in (grid, max(1, grid // 2), max(1, grid // 7), 1):
            for seed in (1, 2):
                assert run(B, t1, t2, t3, grid, resident, seed), (B, grid, resident, seed)
    # the first fused version (tickets by index, no claim flags): half the workgroups resident is a deadlock
    assert not run(129, 15, 25, 15, 1024, 512, 1, claimable=False)
