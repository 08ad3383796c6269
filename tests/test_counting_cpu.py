"""Counting stage: the oracle against the reference's own tiny fixtures and the
reference-generated synthetic golden; host-side parsing/pairing of the product against
the oracle (CPU only)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import counting_oracle as co

CNT = os.path.join(GOLDEN, "counting")
AA = list("ARNDCQEGHILKMFPSTWYV")


def _read_expected(path):
    from cherryml_amd.io import read_count_matrices_arrays
    q, C, states = read_count_matrices_arrays(path)
    return q, C, states


# (dataset, families, mode, alphabet, quantization points, expected dir)  -- the reference's
# tests/counting_tests/counting_test.py cases on the tiny data
TINY_SINGLE = [
    ("tiny", ["fam1", "fam2", "fam3"], "edge", "count_matrices_dir_edges"),
    ("tiny", ["fam1", "fam2", "fam3"], "cherry", "count_matrices_dir_cherries"),
    ("tiny_2", ["fam1", "fam2", "fam3"], "cherry++", "count_matrices_dir_cherries_plus_plus"),
    ("tiny_3", ["fam1"], "cherry++", "count_matrices_dir_cherries_plus_plus"),
]


@pytest.mark.parametrize("ds,fams,mode,exp", TINY_SINGLE)
def test_oracle_single_site_reference_fixtures(ds, fams, mode, exp):
    q, C, states = _read_expected(os.path.join(CNT, ds, exp, "result.txt"))
    got = co.count_transitions(f"{CNT}/{ds}/tree_dir", f"{CNT}/{ds}/msa_dir",
                               f"{CNT}/{ds}/site_rates_dir", fams, list(states), list(q), mode)
    assert np.array_equal(got, C)


TINY_CO = [
    ("tiny", ["fam1", "fam2", "fam3"], "edge", "count_co_matrices_dir_edges"),
    ("tiny", ["fam1", "fam2", "fam3"], "cherry", "count_co_matrices_dir_cherries"),
    ("tiny_2", ["fam1", "fam2", "fam3"], "cherry++", "count_co_matrices_dir_cherries_plus_plus"),
    ("tiny_4", ["fam1"], "cherry++", "count_co_matrices_dir_cherries_plus_plus"),
]


def _alphabet_of_pairs(states):
    aa = []
    for st in states:
        if st[0] not in aa:
            aa.append(st[0])
    assert [a + b for a in aa for b in aa] == list(states)
    return aa


@pytest.mark.parametrize("ds,fams,mode,exp", TINY_CO)
def test_oracle_co_transitions_reference_fixtures(ds, fams, mode, exp):
    q, C, states = _read_expected(os.path.join(CNT, ds, exp, "result.txt"))
    aa = _alphabet_of_pairs(states)
    # minimum_distance_for_nontrivial_contact = 2 in the reference's tests
    got = co.count_co_transitions(f"{CNT}/{ds}/tree_dir", f"{CNT}/{ds}/msa_dir",
                                  f"{CNT}/{ds}/contact_map_dir", fams, aa, list(q), mode, 2)
    assert np.array_equal(got, C)


@pytest.mark.parametrize("mode", ["edge", "cherry", "cherry++"])
def test_oracle_matches_reference_on_synthetic_families(mode):
    g = np.load(os.path.join(GOLDEN, "counting_synth.npz"))
    d = os.path.join(CNT, "synth")
    fams = [str(f) for f in g["families"]]
    got = co.count_transitions(f"{d}/tree_dir", f"{d}/msa_dir", f"{d}/site_rates_dir", fams, AA,
                               list(g["grid"]), mode)
    assert np.array_equal(got, g[f"single_{mode}"])
    co_got = co.count_co_transitions(f"{d}/tree_dir", f"{d}/msa_dir", f"{d}/contact_map_dir", fams,
                                     AA, list(g["grid"]), mode, 3)
    want = np.zeros_like(co_got)
    want[tuple(g[f"co_{mode}_idx"])] = g[f"co_{mode}_val"]
    assert np.array_equal(co_got, want)


def test_quantization_idx_edge_cases():
    grid = np.array([1.0, 2.0, 4.0])
    assert co.quantization_idx(0.99, grid) is None and co.quantization_idx(4.01, grid) is None
    assert co.quantization_idx(1.0, grid) == 0 and co.quantization_idx(4.0, grid) == 2
    # tie in relative error goes right: b/1 - 1 == 2/b - 1  at b = sqrt(2)
    b = float(np.sqrt(2.0))
    assert co.quantization_idx(np.nextafter(b, 0), grid) == 0
    assert co.quantization_idx(np.nextafter(b, 9), grid) == 1


@pytest.mark.parametrize("mode", ["edge", "cherry", "cherry++"])
def test_product_host_pairing_matches_oracle(mode):
    """cherryml_amd.counting._host (parsing, iterative pairing, encoding) vs the oracle."""
    from cherryml_amd.counting import _host
    for ds, fams in [("synth", ["famA", "famB", "famC", "famD"]), ("tiny_2", ["fam1", "fam2", "fam3"])]:
        if ds == "tiny_2" and mode == "edge":
            continue  # that MSA has no internal sequences
        for fam in fams:
            tp = f"{CNT}/{ds}/tree_dir/{fam}.txt"
            names, children, root = _host.read_tree_arrays(tp)
            onodes, ochildren, oroot = co.read_tree(tp)
            assert names == onodes and names[root] == oroot
            got = [(names[a], names[b], la, lb) for a, b, la, lb in _host.build_pairs(children, root, mode)]
            assert got == co.transition_pairs(onodes, ochildren, oroot, mode)
            msa = _host.read_msa(f"{CNT}/{ds}/msa_dir/{fam}.txt")
            assert msa == co.read_msa(f"{CNT}/{ds}/msa_dir/{fam}.txt")
            codes = _host.encode_msa(msa, sorted(msa), AA)
            for r, nm in enumerate(sorted(msa)):
                assert [AA[c] if c >= 0 else None for c in codes[r]] == \
                    [ch if ch in AA else None for ch in msa[nm]]
            assert list(_host.read_site_rates(f"{CNT}/{ds}/site_rates_dir/{fam}.txt")) == \
                co.read_site_rates(f"{CNT}/{ds}/site_rates_dir/{fam}.txt")
    cm = _host.read_contact_map(f"{CNT}/synth/contact_map_dir/famB.txt")
    assert np.array_equal(cm, co.read_contact_map(f"{CNT}/synth/contact_map_dir/famB.txt"))


def test_counting_bad_inputs(tmp_path):
    from cherryml_amd.counting import _host
    p = tmp_path / "t.txt"
    p.write_text("2 nodez\na\nb\n1 edges\na b 1.0\n")
    with pytest.raises(Exception):
        _host.read_tree_arrays(str(p))
    p.write_text("3 nodes\na\nb\nc\n2 edges\na c 1.0\nb c 1.0\n")
    with pytest.raises(Exception):
        _host.read_tree_arrays(str(p))  # c has two parents
    with pytest.raises(ValueError):
        _host.build_pairs([[]], 0, "cherries")


def test_maximal_matching_equals_networkx(tmp_path):
    """create_maximal_matching_contact_map == networkx.maximal_matching on the same graph
    construction as the reference (evaluation/_maximal_matching.py:70-93)."""
    nx = pytest.importorskip("networkx")
    from cherryml_amd.estimation_end_to_end import create_maximal_matching_contact_map
    from cherryml_amd.counting import _host
    src = f"{CNT}/synth/contact_map_dir"
    fams = ["famA", "famB", "famD"]
    out = str(tmp_path / "mm")
    create_maximal_matching_contact_map(i_contact_map_dir=src, families=fams,
                                        minimum_distance_for_nontrivial_contact=3, num_processes=1,
                                        o_contact_map_dir=out)
    for fam in fams:
        cm = _host.read_contact_map(f"{src}/{fam}.txt")
        n = cm.shape[0]
        G = nx.Graph()
        G.add_nodes_from(range(n))
        G.add_edges_from([(i, j) for i, j in zip(*np.where(cm == 1)) if i < j and abs(i - j) >= 3])
        want = np.zeros((n, n), dtype=int)
        for u, v in nx.maximal_matching(G):
            want[u, v] = want[v, u] = 1
        got = _host.read_contact_map(f"{out}/{fam}.txt")
        assert np.array_equal(got, want)
        assert got.sum(axis=0).max() <= 1
