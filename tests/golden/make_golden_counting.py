#!/usr/bin/env python3
"""Golden vectors for the counting stage, produced by RUNNING the
reference's Python counters (use_cpp_implementation=False) on a synthetic family set.

tests/golden/counting/tiny* are the reference tests' own data files (inputs and expected
result.txt, tests/counting_tests/test_input_data/tiny*), copied as data.
This script adds tests/golden/counting/synth/ (inputs, written here with a fixed seed) and
tests/golden/counting_synth.npz (the reference's outputs on them).

Usage (build container only):  python tests/golden/make_golden_counting.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _prepare_scratch  # noqa: E402

AA = list("ARNDCQEGHILKMFPSTWYV")


def random_tree(rng, n_leaves, prefix):
    """random rooted tree with multifurcations; returns (nodes, edges) in file order"""
    nodes = [f"{prefix}root"]
    edges = []
    frontier = [nodes[0]]
    leaves = []
    k = 0
    while len(leaves) + len(frontier) < n_leaves:
        u = frontier.pop(rng.integers(len(frontier)))
        for _ in range(int(rng.choice([2, 2, 2, 3]))):
            k += 1
            v = f"{prefix}n{k}"
            nodes.append(v)
            edges.append((u, v, float(np.round(rng.exponential(0.15) + 1e-4, 6))))
            frontier.append(v)
    return nodes, edges


def main():
    out_dir = os.path.join(HERE, "counting", "synth")
    rng = np.random.default_rng(7)
    fams = ["famA", "famB", "famC", "famD"]
    for d in ["tree_dir", "msa_dir", "site_rates_dir", "contact_map_dir"]:
        os.makedirs(os.path.join(out_dir, d), exist_ok=True)
    for f, (n_leaves, L) in zip(fams, [(17, 23), (40, 31), (9, 12), (64, 40)]):
        nodes, edges = random_tree(rng, n_leaves, "")
        with open(os.path.join(out_dir, "tree_dir", f + ".txt"), "w") as fh:
            fh.write(f"{len(nodes)} nodes\n" + "".join(n + "\n" for n in nodes))
            fh.write(f"{len(edges)} edges\n" + "".join(f"{u} {v} {l}\n" for u, v, l in edges))
        alphabet = AA + ["-", "X"]
        p = np.array([1.0] * 20 + [2.0, 0.5])
        p /= p.sum()
        with open(os.path.join(out_dir, "msa_dir", f + ".txt"), "w") as fh:
            for n in sorted(nodes):
                fh.write(f">{n}\n" + "".join(rng.choice(alphabet, size=L, p=p)) + "\n")
        rates = np.round(rng.gamma(3.0, 1 / 3.0, size=L) + 0.01, 5)
        with open(os.path.join(out_dir, "site_rates_dir", f + ".txt"), "w") as fh:
            fh.write(f"{L} sites\n" + " ".join(str(float(r)) for r in rates))
        cm = (rng.random((L, L)) < 0.12).astype(int)
        cm = np.triu(cm, 1)
        cm = cm + cm.T + np.eye(L, dtype=int)
        with open(os.path.join(out_dir, "contact_map_dir", f + ".txt"), "w") as fh:
            fh.write(f"{L} sites\n" + "".join("".join(str(x) for x in row) + "\n" for row in cm))

    _prepare_scratch()
    from cherryml.counting import count_co_transitions, count_transitions
    from cherryml.io import read_count_matrices

    grid = [float("%.8f" % (0.03 * 1.1 ** i)) for i in range(-64, 65)]
    res = {}
    for mode in ["edge", "cherry", "cherry++"]:
        with tempfile.TemporaryDirectory() as d:
            count_transitions(tree_dir=f"{out_dir}/tree_dir", msa_dir=f"{out_dir}/msa_dir",
                              site_rates_dir=f"{out_dir}/site_rates_dir", families=fams,
                              amino_acids=AA, quantization_points=grid, edge_or_cherry=mode,
                              output_count_matrices_dir=d, num_processes=1,
                              use_cpp_implementation=False)
            cms = read_count_matrices(os.path.join(d, "result.txt"))
            res[f"single_{mode}"] = np.stack([m.to_numpy() for _, m in cms])
        with tempfile.TemporaryDirectory() as d:
            count_co_transitions(tree_dir=f"{out_dir}/tree_dir", msa_dir=f"{out_dir}/msa_dir",
                                 contact_map_dir=f"{out_dir}/contact_map_dir", families=fams,
                                 amino_acids=AA, quantization_points=grid, edge_or_cherry=mode,
                                 minimum_distance_for_nontrivial_contact=3,
                                 output_count_matrices_dir=d, num_processes=1,
                                 use_cpp_implementation=False)
            cms = read_count_matrices(os.path.join(d, "result.txt"))
            C = np.stack([m.to_numpy() for _, m in cms])
            nz = np.nonzero(C)
            res[f"co_{mode}_idx"] = np.stack(nz).astype(np.int32)
            res[f"co_{mode}_val"] = C[nz]
            res[f"co_{mode}_states"] = np.array(list(cms[0][1].index))
    np.savez_compressed(os.path.join(HERE, "counting_synth.npz"), grid=np.array(grid),
                        families=np.array(fams), **res)
    for k, v in res.items():
        if not k.endswith("states"):
            print(k, v.shape, float(np.sum(v)) if "idx" not in k else "")


if __name__ == "__main__":
    main()
