"""FastCherries host side (no GPU): the seeded divide-and-conquer pairing and the initial site-rate
weights against vectors produced by the reference's own C++ (tests/golden/make_golden_fast_cherries.py,
oracle/_ref/libref_fc.so), and -- where that library is present -- against the compiled reference
directly on fresh random inputs."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

from cherryml_amd.phylogeny_estimation._fast_cherries import (_Mt19937, _through_text, _uniform_index,
                                                              cherries_to_tree, divide_and_pair,
                                                              get_weights_for_initial_site_rates,
                                                              rate_categories_ble)

REF = os.path.join(ROOT, "oracle", "_ref", "libref_fc.so")


def test_mersenne_twister_is_std_mt19937():
    """std::mt19937's 10000th output for the default seed 5489 is 4123659995 (the C++ standard's own check);
    numpy's legacy RandomState seeds the same generator the same way."""
    rng = _Mt19937(5489)
    for _ in range(9999):
        rng()
    assert rng() == 4123659995
    mine, ref = _Mt19937(1234), np.random.RandomState(1234)
    assert [mine() for _ in range(700)] == [int(x) for x in ref.randint(0, 2 ** 32, size=700, dtype=np.uint64)]


def test_uniform_index_schemes_stay_in_range():
    for scheme in ("lemire", "gcc10"):
        rng = _Mt19937(3)
        for n in (1, 2, 3, 7, 1000, 2 ** 31 + 5):
            assert all(0 <= _uniform_index(rng, n, scheme) < n for _ in range(50))
    with pytest.raises(ValueError):
        _uniform_index(_Mt19937(1), 5, "other")


def test_pairing_matches_reference_goldens():
    g = load_golden("fast_cherries.npz")
    k = 0
    while f"pair{k}_seqs" in g:
        got = divide_and_pair(g[f"pair{k}_seqs"], seed=int(g[f"pair{k}_seed"]))           # libcherrybank's host routine
        assert got == [tuple(int(v) for v in p) for p in g[f"pair{k}_pairs"]], k
        assert divide_and_pair(g[f"pair{k}_seqs"], seed=int(g[f"pair{k}_seed"]), _force_python=True) == got   # numpy route
        n = g[f"pair{k}_seqs"].shape[0]
        assert len(got) == n // 2 and len({i for p in got for i in p}) == 2 * (n // 2)   # a matching
        k += 1
    assert k >= 5


def test_initial_weights_and_rate_categories():
    g = load_golden("fast_cherries.npz")
    r = rate_categories_ble(20)
    assert abs(r[0] - 0.05) < 1e-16 and abs(r[-1] - 20.0) < 1e-10 and len(r) == 20
    assert np.array_equal(np.array(get_weights_for_initial_site_rates(r)), g["w20"])   # bit for bit
    assert rate_categories_ble(1) == [1.0]


def test_text_round_trip_and_tree_rule():
    assert _through_text([0.1234567890123456789, 6.729e-05 * 1.03])[1] == float("%.17f" % (6.729e-05 * 1.03))
    t = cherries_to_tree(["a", "b", "c", "d", "e"], [("a", "c"), ("e", "b")], [0.5, 0.25])
    assert t.root() == "root"
    assert t.edges() == [("root", "internal-0", 1.0), ("internal-0", "a", 0.25), ("internal-0", "c", 0.25),
                         ("root", "internal-1", 1.0), ("internal-1", "e", 0.125), ("internal-1", "b", 0.125),
                         ("root", "d", 1.0)]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libref_fc.so not built (needs /root/reference)")
def test_pairing_and_weights_against_the_compiled_reference():
    lib = C.CDLL(REF)
    rng = np.random.default_rng(123)
    for trial in range(25):
        n, L = int(rng.integers(2, 120)), int(rng.integers(1, 50))
        seqs = np.tile(rng.integers(0, 20, size=L), (n, 1))
        flip = rng.random((n, L)) < rng.uniform(0.02, 0.7)
        seqs[flip] = rng.integers(0, 20, size=int(flip.sum()))
        seqs[rng.random((n, L)) < 0.15] = -1
        s32 = np.ascontiguousarray(seqs, dtype=np.int32)
        out = np.zeros(2 * n, dtype=np.int32)
        m = lib.ref_divide_and_pair(s32.ctypes.data_as(C.c_void_p), n, L, 7 + trial, out.ctypes.data_as(C.c_void_p))
        want = [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(m)]
        assert divide_and_pair(seqs, seed=7 + trial) == want
        assert divide_and_pair(seqs, seed=7 + trial, _force_python=True) == want
        if trial < 5:   # the older libstdc++ rule: both of our routes agree with each other
            assert (divide_and_pair(seqs, seed=3, rng_scheme="gcc10")
                    == divide_and_pair(seqs, seed=3, rng_scheme="gcc10", _force_python=True))
    for R in (2, 5, 20, 41):
        r = np.array(rate_categories_ble(R))
        w = np.zeros(R)
        lib.ref_initial_weights(r.ctypes.data_as(C.c_void_p), R, w.ctypes.data_as(C.c_void_p))
        assert np.array_equal(np.array(get_weights_for_initial_site_rates(r)), w)
