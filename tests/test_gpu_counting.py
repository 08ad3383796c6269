"""Counting on the MI355X (cb_count_transitions / cb_count_co_transitions through the
mirrored stage functions) against the reference tests' own expected files, the
reference-generated synthetic golden and the oracle: BIT-EXACT."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import counting_oracle as co
from test_counting_cpu import AA, CNT, TINY_CO, TINY_SINGLE, _alphabet_of_pairs, _read_expected

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ds,fams,mode,exp", TINY_SINGLE)
def test_count_transitions_reference_fixtures(ds, fams, mode, exp, tmp_path):
    import cherryml_amd
    q, C, states = _read_expected(os.path.join(CNT, ds, exp, "result.txt"))
    out = str(tmp_path / "out")
    cherryml_amd.count_transitions(
        tree_dir=f"{CNT}/{ds}/tree_dir", msa_dir=f"{CNT}/{ds}/msa_dir",
        site_rates_dir=f"{CNT}/{ds}/site_rates_dir", families=fams, amino_acids=list(states),
        quantization_points=list(q), edge_or_cherry=mode, output_count_matrices_dir=out,
        num_processes=3)
    q2, C2, st2 = _read_expected(os.path.join(out, "result.txt"))
    assert list(st2) == list(states) and np.array_equal(q2, q) and np.array_equal(C2, C)
    assert open(os.path.join(out, "profiling.txt")).read().startswith("Total time: ")


@pytest.mark.parametrize("ds,fams,mode,exp", TINY_CO)
def test_count_co_transitions_reference_fixtures(ds, fams, mode, exp, tmp_path):
    import cherryml_amd
    q, C, states = _read_expected(os.path.join(CNT, ds, exp, "result.txt"))
    out = str(tmp_path / "out")
    cherryml_amd.count_co_transitions(
        tree_dir=f"{CNT}/{ds}/tree_dir", msa_dir=f"{CNT}/{ds}/msa_dir",
        contact_map_dir=f"{CNT}/{ds}/contact_map_dir", families=fams,
        amino_acids=_alphabet_of_pairs(states), quantization_points=list(q), edge_or_cherry=mode,
        minimum_distance_for_nontrivial_contact=2, output_count_matrices_dir=out)
    q2, C2, st2 = _read_expected(os.path.join(out, "result.txt"))
    assert list(st2) == list(states) and np.array_equal(C2, C)


@pytest.mark.parametrize("mode", ["edge", "cherry", "cherry++"])
def test_counting_synthetic_families_vs_reference_golden(mode, tmp_path):
    import cherryml_amd
    g = np.load(os.path.join(GOLDEN, "counting_synth.npz"))
    d = os.path.join(CNT, "synth")
    fams = [str(f) for f in g["families"]]
    out = str(tmp_path / "single")
    cherryml_amd.count_transitions(
        tree_dir=f"{d}/tree_dir", msa_dir=f"{d}/msa_dir", site_rates_dir=f"{d}/site_rates_dir",
        families=fams, amino_acids=AA, quantization_points=[str(x) for x in g["grid"]],
        edge_or_cherry=mode, output_count_matrices_dir=out)
    _, C, _ = _read_expected(os.path.join(out, "result.txt"))
    assert np.array_equal(C, g[f"single_{mode}"])
    out = str(tmp_path / "co")
    cherryml_amd.count_co_transitions(
        tree_dir=f"{d}/tree_dir", msa_dir=f"{d}/msa_dir", contact_map_dir=f"{d}/contact_map_dir",
        families=fams, amino_acids=AA, quantization_points=list(g["grid"]), edge_or_cherry=mode,
        minimum_distance_for_nontrivial_contact=3, output_count_matrices_dir=out)
    _, C, states = _read_expected(os.path.join(out, "result.txt"))
    want = np.zeros_like(C)
    want[tuple(g[f"co_{mode}_idx"])] = g[f"co_{mode}_val"]
    assert np.array_equal(C, want)
    assert list(states) == [str(s) for s in g[f"co_{mode}_states"]]


def test_quantisation_ties_and_range_on_device(tmp_path):
    """branch lengths sitting exactly on grid points, on relative-error ties and outside the
    grid, through the device quantiser vs the oracle's."""
    import cherryml_amd
    grid = [1.0, 2.0, 4.0, 8.0]
    tie = float(np.sqrt(2.0))
    lengths = [0.5, 1.0, np.nextafter(tie, 0), tie, np.nextafter(tie, 9), 2.0, 2.9, 7.99, 8.0,
               np.nextafter(8.0, 9), 1e-300]
    d = tmp_path
    for sub in ["tree_dir", "msa_dir", "site_rates_dir"]:
        os.makedirs(d / sub)
    n = len(lengths)
    with open(d / "tree_dir" / "f.txt", "w") as f:
        f.write(f"{n + 1} nodes\nroot\n" + "".join(f"l{i}\n" for i in range(n)))
        f.write(f"{n} edges\n" + "".join(f"root l{i} {repr(float(x))}\n" for i, x in enumerate(lengths)))
    with open(d / "msa_dir" / "f.txt", "w") as f:
        f.write(">root\nAC\n" + "".join(f">l{i}\nCA\n" for i in range(n)))
    with open(d / "site_rates_dir" / "f.txt", "w") as f:
        f.write("2 sites\n1.0 1.0")
    out = str(d / "out")
    cherryml_amd.count_transitions(
        tree_dir=str(d / "tree_dir"), msa_dir=str(d / "msa_dir"), site_rates_dir=str(d / "site_rates_dir"),
        families=["f"], amino_acids=["A", "C"], quantization_points=grid, edge_or_cherry="edge",
        output_count_matrices_dir=out)
    _, C, _ = _read_expected(os.path.join(out, "result.txt"))
    want = co.count_transitions(str(d / "tree_dir"), str(d / "msa_dir"), str(d / "site_rates_dir"),
                                ["f"], ["A", "C"], grid, "edge")
    assert np.array_equal(C, want)
    assert C.sum() == 2 * sum(1 for x in lengths if grid[0] <= x <= grid[-1])
