"""FastCherries end to end on the GPU (pairing on the host, branch lengths / site rates by cb_ble) against
the outputs of the reference's own C++ program (tests/golden/make_golden_fast_cherries.py), the stage's
files, and `learn_site_rate_matrices(tree=None)` against the reference's two halves glued as its wrapper
glues them (tests/golden/make_golden_siterm_learn.py, case "fc")."""
import os

import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [0, 1, 2])
def test_family_matches_the_reference_program(k):
    from cherryml_amd.phylogeny_estimation import fast_cherries_family
    g = load_golden("fast_cherries.npz")
    prof = {}
    cherries, lengths, rates = fast_cherries_family(
        [str(n) for n in g[f"fam{k}_names"]], [str(s) for s in g[f"fam{k}_seqs"]], g["Q"],
        [str(a) for a in g["alphabet"]], num_rate_categories=int(g[f"fam{k}_rcat"]),
        max_iters=int(g[f"fam{k}_iters"]), seed=int(g[f"fam{k}_seed"]), profile=prof)
    assert cherries == [(str(a), str(b)) for a, b in g[f"fam{k}_cherries"]]
    assert np.array_equal(lengths, g[f"fam{k}_lengths"])         # grid values x mean rate, through the text format
    assert np.array_equal(rates, g[f"fam{k}_site_rates"])
    assert prof["iterations"] >= 1 and "pairing_time" in prof


def test_stage_writes_the_reference_files(tmp_path):
    from cherryml_amd.io import read_tree, write_rate_matrix
    from cherryml_amd.phylogeny_estimation import fast_cherries
    g = load_golden("fast_cherries.npz")
    k = 1
    alphabet = [str(a) for a in g["alphabet"]]
    msa_dir = tmp_path / "msas"
    msa_dir.mkdir()
    names = [str(n) for n in g[f"fam{k}_names"]]
    with open(msa_dir / "famA.txt", "w") as f:
        f.write("".join(f">{n}\n{s}\n" for n, s in zip(names, g[f"fam{k}_seqs"])))
    qpath = str(tmp_path / "Q.txt")
    write_rate_matrix(g["Q"], alphabet, qpath)
    out = {d: str(tmp_path / d) for d in ("trees", "rates", "lls")}
    fast_cherries(msa_dir=str(msa_dir), families=["famA"], rate_matrix_path=qpath,
                  num_rate_categories=int(g[f"fam{k}_rcat"]), max_iters=int(g[f"fam{k}_iters"]), num_processes=1,
                  output_tree_dir=out["trees"], output_site_rates_dir=out["rates"], output_likelihood_dir=out["lls"],
                  verbose=False, seed=int(g[f"fam{k}_seed"]))
    tree = read_tree(os.path.join(out["trees"], "famA.txt"))
    want = []
    for i, ((a, b), d) in enumerate(zip(g[f"fam{k}_cherries"], g[f"fam{k}_lengths"])):
        want += [("root", f"internal-{i}", 1.0), (f"internal-{i}", str(a), d / 2.0), (f"internal-{i}", str(b), d / 2.0)]
    assert tree.edges() == want and sorted(tree.leaves()) == sorted(names)
    lines = open(os.path.join(out["rates"], "famA.txt")).read().split("\n")
    assert lines[0] == f"{len(g[f'fam{k}_site_rates'])} sites"
    assert np.array_equal(np.array([float(x) for x in lines[1].split()]), g[f"fam{k}_site_rates"])
    assert open(os.path.join(out["lls"], "famA.txt")).read() == "0.0"
    assert os.path.exists(os.path.join(out["trees"], "famA.profiling"))


def test_learn_site_rate_matrices_without_a_tree():
    import pandas as pd
    from cherryml_amd._siterm import learn_site_rate_matrices
    g = load_golden("siterm_learn.npz")
    dna, alpha5 = ["A", "C", "G", "T"], ["A", "C", "G", "T", "-"]

    def equ(states):
        n = len(states)
        Q = np.full((n, n), 1.0 / (n - 1))
        np.fill_diagonal(Q, -1.0)
        return pd.DataFrame(Q, index=states, columns=states)

    msa = {str(n): str(s) for n, s in zip(g["fc_msa_names"], g["fc_msa_seqs"])}
    kw = dict(tree=None, leaf_states=msa, alphabet=alpha5, regularization_rate_matrix=equ(alpha5),
              regularization_strength=0.5, alphabet_for_site_rate_estimation=dna,
              rate_matrix_for_site_rate_estimation=equ(dna), num_epochs=20, quantization_grid_num_steps=16)
    r = learn_site_rate_matrices(**kw)
    assert np.array_equal(np.asarray(r["learnt_site_rates"]), g["fc_site_rates"])
    want_edges = [(str(u), str(v), float(t)) for u, v, t in zip(g["fc_edges_u"], g["fc_edges_v"], g["fc_edges_t"])]
    assert r["learnt_tree"].edges() == want_edges
    for l in range(g["fc_res"].shape[0]):
        assert relerr(r["learnt_rate_matrices"][l], g["fc_res"][l]) < 1e-6, l
    j = learn_site_rate_matrices(just_run_fast_cherries=True, **kw)
    assert j["learnt_rate_matrices"] is None and j["learnt_tree"].edges() == want_edges
    assert np.array_equal(np.asarray(j["learnt_site_rates"]), g["fc_site_rates"])


def test_public_api_just_run_fast_cherries():
    """The reference's own `test_just_run_fast_cherries` (_siterm_public_api.py:212-231)."""
    import pandas as pd
    import cherryml_amd
    dna = ["A", "C", "G", "T"]
    Q = pd.DataFrame(np.full((4, 4), 1.0 / 3.0) - np.eye(4) * (4.0 / 3.0), index=dna, columns=dna)
    r = cherryml_amd.learn_site_specific_rate_matrices(
        tree=None, msa={"leaf_1": "AAAA", "leaf_2": "AAAT", "leaf_3": "TTTA", "leaf_4": "TTTT"}, alphabet=dna,
        regularization_rate_matrix=Q, just_run_fast_cherries=True)
    assert r["learnt_site_rates"] is not None and len(r["learnt_site_rates"]) == 4
    assert r["learnt_tree"] is not None and sorted(r["learnt_tree"].leaves()) == ["leaf_1", "leaf_2", "leaf_3", "leaf_4"]
    assert r["learnt_rate_matrices"] is None


def test_many_families_in_one_call_equal_the_per_family_calls(tmp_path):
    """`cb_ble_batch` / `fast_cherries_families` (the reference's pool over families as ONE device call: shared
    bank uploaded once, lockstep coordinate ascents, one flag read-back per round): family by family bit-identical
    to `fast_cherries_family` -- ragged sizes, a family of two identical sequences (converges at once), the three
    golden families of the reference's own program -- and the stage writes one file set per family."""
    from cherryml_amd.io import write_rate_matrix
    from cherryml_amd.phylogeny_estimation import fast_cherries, fast_cherries_families, fast_cherries_family
    g = load_golden("fast_cherries.npz")
    alphabet = [str(a) for a in g["alphabet"]]
    rng = np.random.default_rng(17)
    msas = [([str(n) for n in g[f"fam{k}_names"]], [str(s) for s in g[f"fam{k}_seqs"]]) for k in range(3)]
    for n, L in ((2, 9), (7, 33), (40, 120), (13, 5)):
        base = rng.integers(0, len(alphabet), size=L)
        seqs = []
        for _ in range(n):
            s = base.copy()
            flip = rng.random(L) < 0.25
            s[flip] = rng.integers(0, len(alphabet), size=int(flip.sum()))
            txt = [alphabet[c] for c in s]
            for j in np.flatnonzero(rng.random(L) < 0.05):
                txt[j] = "-"
            seqs.append("".join(txt))
        msas.append(([f"s{n}_{i}" for i in range(n)], seqs))
    msas.append((["a", "b"], ["".join(alphabet[:8]), "".join(alphabet[:8])]))
    kw = dict(num_rate_categories=20, max_iters=50, seed=1234)
    prof = {}
    batch = fast_cherries_families(msas, g["Q"], alphabet, profile=prof, **kw)
    assert len(batch) == len(msas) and len(prof["iterations"]) == len(msas) and prof["kernel_ms"] > 0
    for (names, seqs), (ch, le, ra) in zip(msas, batch):
        one_prof = {}
        ch1, le1, ra1 = fast_cherries_family(names, seqs, g["Q"], alphabet, profile=one_prof, **kw)
        assert ch == ch1 and np.array_equal(le, le1) and np.array_equal(ra, ra1)
    # the golden family whose settings these are reproduces the reference program through the batch too
    for k in range(3):
        if int(g[f"fam{k}_rcat"]) == 20 and int(g[f"fam{k}_seed"]) == 1234 and int(g[f"fam{k}_iters"]) == 50:
            assert np.array_equal(batch[k][1], g[f"fam{k}_lengths"]) and np.array_equal(batch[k][2], g[f"fam{k}_site_rates"])
    # the stage function: all families through one batch, one file set each
    msa_dir = tmp_path / "msas"
    msa_dir.mkdir()
    fams = [f"fam{i}" for i in range(len(msas))]
    for fam, (names, seqs) in zip(fams, msas):
        with open(msa_dir / f"{fam}.txt", "w") as f:
            f.write("".join(f">{n}\n{s}\n" for n, s in zip(names, seqs)))
    qpath = str(tmp_path / "Q.txt")
    write_rate_matrix(g["Q"], alphabet, qpath)
    out = {d: str(tmp_path / d) for d in ("trees", "rates", "lls")}
    fast_cherries(msa_dir=str(msa_dir), families=fams, rate_matrix_path=qpath, num_processes=4, output_tree_dir=out["trees"],
                  output_site_rates_dir=out["rates"], output_likelihood_dir=out["lls"], verbose=False, **kw)
    for fam, (_, _, ra) in zip(fams, batch):
        lines = open(os.path.join(out["rates"], fam + ".txt")).read().split("\n")
        assert np.array_equal(np.array([float(x) for x in lines[1].split()]), ra)
        assert os.path.exists(os.path.join(out["trees"], fam + ".txt"))
