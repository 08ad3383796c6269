"""FastCherries end to end (reference cherryml/phylogeny_estimation/_fast_cherries.py:185-281 and the C++
program FastCherries/fast_cherries.cpp:170-321): divide-and-conquer cherry pairing, then branch lengths and
site rates by coordinate ascent, then the star-of-cherries tree.

* pairing (`pairing_algorithms.cpp:77-175`): on the host -- `cb_fc_divide_and_pair` in libcherrybank (C++,
  int8 sequences, index vectors: 5x the reference's own C++, which hashes sequence names for every
  distance), with the numpy restatement below as the fallback and cross-check.  It is a seeded recursion over
  shrinking subsets whose only arithmetic is Hamming distances of one sequence against a subset (three
  passes per level), O(n L log n) byte compares in total.  The random pivot follows the reference bit for bit: `std::mt19937(seed)` + libstdc++'s
  `uniform_int_distribution<size_t>` (GCC >= 11: Lemire's multiply-shift with rejection; the older
  scale-and-reject rule is available as `rng_scheme="gcc10"`).
* branch lengths / site rates (`ble`, branch_length_estimation.cpp:146-241): on the GPU (`cb_ble`), the
  log-transition bank from the expm kernels of the hot path (`cb_ble_log_bank`).
* the reference round-trips lengths and rates through text files written with 17 fixed decimals
  (io_helpers.cpp:75-101); `_through_text` reproduces that rounding so that results are equal bit for bit."""
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ..io._tree import Tree, write_tree
from ._ble import compute_log_transition_matrices, estimate_branch_lengths_and_site_rates


# ----------------------------------------------------------------------------- the reference's random pivot
class _Mt19937:
    """std::mt19937 (32-bit Mersenne twister, `init_genrand` seeding): raw 32-bit draws."""

    def __init__(self, seed: int):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt, self.idx = mt, 624

    def __call__(self) -> int:
        if self.idx >= 624:
            mt = self.mt
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def _uniform_index(rng: _Mt19937, n: int, scheme: str = "lemire") -> int:
    """`std::uniform_int_distribution<size_t>(0, n - 1)(rng)` of libstdc++ for a 32-bit engine."""
    if n <= 0 or n > 0xFFFFFFFF:
        raise ValueError("range")
    if scheme == "lemire":      # GCC >= 11 (bits/uniform_int_dist.h, _S_nd): multiply-shift, reject the biased low part
        product = rng() * n
        low = product & 0xFFFFFFFF
        if low < n:
            threshold = ((1 << 32) - n) % n
            while low < threshold:
                product = rng() * n
                low = product & 0xFFFFFFFF
        return product >> 32
    if scheme == "gcc10":       # older libstdc++: scale and reject
        scaling = 0xFFFFFFFF // n
        past = n * scaling
        while True:
            r = rng()
            if r < past:
                return r // scaling
    raise ValueError("rng_scheme must be 'lemire' or 'gcc10'")


# ----------------------------------------------------------------------------- pairing
def _neg_hamming(A: np.ndarray, x: np.ndarray) -> np.ndarray:
    """pairing_algorithms.cpp:14-38: -(mismatches / compared sites) over the sites where both are known."""
    valid = (A != -1) & (x[None, :] != -1)
    count = valid.sum(axis=1)
    dist = ((A != x[None, :]) & valid).sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = (dist * -1.0) / count
    return np.where(count == 0, 0.0, d)


def _divide_and_pair_native(seqs, seed: int, rng_scheme: str):
    """libcherrybank's host implementation (cb_fc_divide_and_pair; 5-40x the reference's C++, which
    hashes sequence names for every distance); None when the library is not there."""
    if rng_scheme not in ("lemire", "gcc10"):
        raise ValueError("rng_scheme must be 'lemire' or 'gcc10'")
    try:
        from .. import _lib
        lib = _lib.load()
    except Exception:
        return None
    s8 = np.ascontiguousarray(seqs, dtype=np.int8)
    if s8.ndim != 2 or not np.array_equal(s8, np.asarray(seqs)):
        return None                                   # states beyond int8: the numpy route
    n, L = s8.shape
    out = np.zeros(2 * max(n // 2, 1), dtype=np.int32)
    m = lib.cb_fc_divide_and_pair(s8.ctypes.data, n, L, int(seed) & 0xFFFFFFFF, 0 if rng_scheme == "lemire" else 1,
                                  out.ctypes.data)
    if m < 0:
        _lib.check(m, "cb_fc_divide_and_pair")
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(m)]


def divide_and_pair(seqs: np.ndarray, seed: int = 1234, rng_scheme: str = "lemire",
                    _force_python: bool = False) -> List[Tuple[int, int]]:
    """`divide_and_pair` (:166-175): cherries as index pairs into `seqs` [n, L] (state index, -1 unknown),
    in the reference's order.  Every sequence but at most one ends up in a cherry."""
    if not _force_python:
        native = _divide_and_pair_native(seqs, seed, rng_scheme)
        if native is not None:
            return native
    seqs = np.ascontiguousarray(seqs, dtype=np.int16)
    rng = _Mt19937(seed)

    def divide(ids: List[int]) -> Tuple[Optional[int], List[Tuple[int, int]]]:
        if len(ids) == 2:
            return None, [(ids[0], ids[1])]
        if len(ids) == 1:
            return ids[0], []
        if len(ids) == 0:
            return None, []
        A = seqs[ids]
        x = _uniform_index(rng, len(ids), rng_scheme)
        x = int(np.argmin(_neg_hamming(A, A[x])))            # farthest from the random pivot (first minimum)
        dist_x = _neg_hamming(A, A[x])
        y = int(np.argmin(dist_x))                           # farthest from that one
        closer_x = dist_x >= _neg_hamming(A, A[y])
        close_x = [ids[i] for i in range(len(ids)) if closer_x[i] and i != y]
        close_y = [ids[i] for i in range(len(ids)) if not (closer_x[i] and i != y)]
        ux, cx = divide(close_x)
        uy, cy = divide(close_y)
        cherries = cx + cy
        unpaired = None
        if ux is not None and uy is not None:
            cherries.append((ux, uy))
        else:
            unpaired = ux if ux is not None else uy
        return unpaired, cherries

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 4 * len(seqs) + 100))     # a degenerate split peels one sequence per level
    try:
        return divide(list(range(len(seqs))))[1]
    finally:
        sys.setrecursionlimit(old)


# ----------------------------------------------------------------------------- initial site-rate weights
def _ln_gamma(alpha: float) -> float:
    """fast_cherries.cpp:59-78 (Pike & Hill 1966, Algorithm 291)."""
    import math
    x, f = alpha, 0.0
    if x < 7:
        f, z = 1.0, x - 1
        while True:
            z += 1
            if not z < 7:
                break
            f *= z
        x = z
        f = -math.log(f)
    z = 1 / (x * x)
    return (f + (x - 0.5) * math.log(x) - x + .918938533204673
            + (((-.000595238095238 * z + .000793650793651) * z - .002777777777778) * z + .083333333333333) / x)


def _incomplete_gamma(x: float, alpha: float, ln_gamma_alpha: float) -> float:
    """fast_cherries.cpp:80-139 (Bhattacharjee 1970, AS32): series for x <= 1 or x < alpha, else the
    continued fraction; accuracy 1e-8 as in the reference."""
    import math
    p, g = alpha, ln_gamma_alpha
    accurate, overflow = 1e-8, 1e30
    if x == 0:
        return 0.0
    if x < 0 or p <= 0:
        return -1.0
    factor = math.exp(p * math.log(x) - x - g)
    if not (x > 1 and x >= p):
        gin, term, rn = 1.0, 1.0, p
        while True:
            rn += 1
            term *= x / rn
            gin += term
            if not term > accurate:
                break
        return gin * (factor / p)      # `gin *= factor / p`
    a, b, term = 1 - p, 1 - p + x + 1, 0.0
    pn = [1.0, x, x + 1, x * b, 0.0, 0.0]
    gin = pn[2] / pn[3]
    while True:
        a += 1
        b += 2
        term += 1
        an = a * term
        for i in range(2):
            pn[i + 4] = b * pn[i + 2] - an * pn[i]
        if pn[5] != 0:
            rn = pn[4] / pn[5]
            dif = abs(gin - rn)
            if not dif > accurate and dif <= accurate * rn:
                return 1 - factor * gin
            gin = rn
        for i in range(4):
            pn[i] = pn[i + 2]
        if not abs(pn[4]) < overflow:
            for i in range(4):
                pn[i] /= overflow


def get_weights_for_initial_site_rates(rate_categories: Sequence[float]) -> List[float]:
    """fast_cherries.cpp:147-167: Gamma(shape 3, rate 3) mass below the geometric midpoints of the rate
    categories (the last weight is 1)."""
    import math
    r = list(rate_categories)
    shape = 3.0
    w = [_incomplete_gamma(math.sqrt(r[i - 1] * r[i]) * shape, shape, _ln_gamma(shape)) for i in range(1, len(r))]
    return w + [1.0]


def rate_categories_ble(num_rate_categories: int) -> List[float]:
    """fast_cherries.cpp:236-243: geometric from 1/R to R."""
    R = int(num_rate_categories)
    start = 1.0 / R
    if R == 1:
        return [start]
    ratio = (R / start) ** (1.0 / (R - 1))
    out = [start]
    for _ in range(1, R):
        out.append(out[-1] * ratio)
    return out


def quantization_points_ble(center: float, step: float, num_steps: int) -> List[float]:
    """io_helpers.cpp:176-194: the grid grown from the centre by repeated multiplication / division in
    `long double` (x87 80-bit on the reference's platforms; numpy's longdouble is the same type on x86-64
    Linux), rounded to double at the end -- NOT `center * step ** i`, which differs in the last bit."""
    ld = np.longdouble
    pts = [ld(0)] * (2 * num_steps + 1)
    pts[num_steps] = ld(center)
    for i in range(1, num_steps + 1):
        pts[num_steps + i] = pts[num_steps + i - 1] * ld(step)
        pts[num_steps - i] = pts[num_steps - i + 1] / ld(step)
    return [float(p) for p in pts]


def _through_text(values) -> np.ndarray:
    """What the reference reads back from the files its C++ program wrote with 17 fixed decimals."""
    return np.array([float("%.17f" % v) for v in np.asarray(values, dtype=np.float64)], dtype=np.float64)


# ----------------------------------------------------------------------------- families, in memory
def _encode_and_pair(sequences: Sequence[str], alphabet: Sequence[str], seed: int, rng_scheme: str):
    code = np.full(256, -1, dtype=np.int8)
    for i, s in enumerate(alphabet):
        if len(s) != 1 or ord(s) >= 256:
            raise ValueError("FastCherries reads one byte per state (io_helpers.cpp:12-16)")
        code[ord(s)] = i
    if len(set(len(s) for s in sequences)) > 1:
        raise ValueError("sequences of one MSA must have equal length")
    seqs = np.stack([code[np.frombuffer(s.encode("latin-1"), dtype=np.uint8)] for s in sequences])
    pairs = divide_and_pair(seqs, seed=seed, rng_scheme=rng_scheme)
    if not pairs:
        raise ValueError("FastCherries needs at least two sequences")
    return seqs, pairs


def _normalise(lengths, site_rates):
    """fast_cherries.cpp:262-304: lengths multiplied and rates divided by the mean rate, both through the
    reference's 17-decimal text files."""
    mean_rate = 0.0
    for r in site_rates:                    # the reference's left-to-right sum
        mean_rate += float(r)
    mean_rate /= len(site_rates)
    return (_through_text(np.array([float(x) * mean_rate for x in lengths])),
            _through_text(np.array([float(r) / mean_rate for r in site_rates])))


def fast_cherries_family(names: Sequence[str], sequences: Sequence[str], rate_matrix: np.ndarray,
                         alphabet: Sequence[str], num_rate_categories: int = 20, max_iters: int = 50,
                         seed: int = 1234, quantization_grid_center: float = 0.03,
                         quantization_grid_step: float = 1.1, quantization_grid_num_steps: int = 64,
                         device: int = 0, rng_scheme: str = "lemire", profile: Optional[dict] = None):
    """One MSA through FastCherries: returns (cherries [(name_a, name_b)], lengths [n_cherries] -- total
    length of each cherry --, site_rates [L]), lengths multiplied and rates divided by the mean rate and
    both rounded through the reference's text files (fast_cherries.cpp:262-304)."""
    Q = np.ascontiguousarray(rate_matrix, dtype=np.float64)
    st = time.time()
    seqs, pairs = _encode_and_pair(sequences, alphabet, seed, rng_scheme)
    t_pair = time.time() - st
    grid = quantization_points_ble(quantization_grid_center, quantization_grid_step, quantization_grid_num_steps)
    rates = rate_categories_ble(num_rate_categories)
    weights = get_weights_for_initial_site_rates(rates)
    st = time.time()
    logP = compute_log_transition_matrices(Q, grid, rates, device=device)
    cx = seqs[[a for a, _ in pairs]]
    cy = seqs[[b for _, b in pairs]]
    lengths, site_rates = estimate_branch_lengths_and_site_rates(
        cx, cy, seqs, logP, grid, rates, weights, max_iters, device=device, profile=profile)
    lengths, site_rates = _normalise(lengths, site_rates)
    if profile is not None:
        profile["pairing_time"], profile["ble_time"] = t_pair, time.time() - st
    return [(names[a], names[b]) for a, b in pairs], lengths, site_rates


def fast_cherries_families(msas: Sequence[Tuple[Sequence[str], Sequence[str]]], rate_matrix: np.ndarray,
                           alphabet: Sequence[str], num_rate_categories: int = 20, max_iters: int = 50,
                           seed: int = 1234, quantization_grid_center: float = 0.03,
                           quantization_grid_step: float = 1.1, quantization_grid_num_steps: int = 64,
                           device: int = 0, rng_scheme: str = "lemire", profile: Optional[dict] = None):
    """MANY MSAs through FastCherries in one device call (the reference maps families over a process pool,
    _fast_cherries.py:185-281 / utils.py:59-67): `msas` = [(names, sequences), ...].  The pairing of every
    family runs on the host first, the log-transition bank is computed ONCE (one rate matrix for all
    families) and `cb_ble_batch` runs all coordinate ascents in lockstep.  Returns, family by family, exactly
    what `fast_cherries_family` returns."""
    Q = np.ascontiguousarray(rate_matrix, dtype=np.float64)
    st = time.time()
    enc = [_encode_and_pair(sequences, alphabet, seed, rng_scheme) for _, sequences in msas]
    t_pair = time.time() - st
    grid = quantization_points_ble(quantization_grid_center, quantization_grid_step, quantization_grid_num_steps)
    rates = rate_categories_ble(num_rate_categories)
    weights = get_weights_for_initial_site_rates(rates)
    st = time.time()
    logP = compute_log_transition_matrices(Q, grid, rates, device=device)
    fams = [(seqs[[a for a, _ in pairs]], seqs[[b for _, b in pairs]], seqs) for seqs, pairs in enc]
    from ._ble import estimate_branch_lengths_and_site_rates_batch
    res = estimate_branch_lengths_and_site_rates_batch(fams, logP, grid, rates, weights, max_iters, device=device,
                                                       profile=profile)
    out = []
    for (names, _), (_, pairs), (lengths, site_rates) in zip(msas, enc, res):
        lengths, site_rates = _normalise(lengths, site_rates)
        out.append(([(names[a], names[b]) for a, b in pairs], lengths, site_rates))
    if profile is not None:
        profile["pairing_time"], profile["ble_time"] = t_pair, time.time() - st
    return out


def cherries_to_tree(names: Sequence[str], cherries: Sequence[Tuple[str, str]], lengths: Sequence[float]) -> Tree:
    """_fast_cherries.py:116-131: a root with one child `internal-<i>` per cherry (branch 1.0, ete3's
    default), the two leaves at half the cherry's length each; an odd sequence out hangs off the root."""
    tree = Tree()
    tree.add_node("root")
    paired = set()
    for i, ((a, b), d) in enumerate(zip(cherries, lengths)):
        inner = f"internal-{i}"
        tree.add_node(inner)
        tree.add_edge("root", inner, 1.0)
        for leaf in (a, b):
            tree.add_node(leaf)
            tree.add_edge(inner, leaf, float(d) / 2.0)
            paired.add(leaf)
    missing = [n for n in names if n not in paired]
    if len(missing) & 1:
        tree.add_node(missing[-1])
        tree.add_edge("root", missing[-1], 1.0)
    return tree


# ----------------------------------------------------------------------------- the stage (files in, files out)
def _read_msa_file(path: str) -> Tuple[List[str], List[str]]:
    """io_helpers.cpp:30-73: `>name` lines, each followed by its sequence line."""
    names, seqs = [], []
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith(">"):
            if i + 1 >= len(lines):
                break
            names.append(lines[i][1:])
            seqs.append(lines[i + 1])
            i += 2
        else:
            i += 1
    return names, seqs


from ..caching import cached_computation  # noqa: E402


@cached_computation(output_dirs=["output_tree_dir", "output_site_rates_dir", "output_likelihood_dir"],
                    exclude_args=["num_processes", "verbose", "remake"])
def fast_cherries(msa_dir: str, families: List[str], rate_matrix_path: str, num_rate_categories: int,
                  max_iters: int, num_processes: int = 1, _version="2", output_tree_dir: Optional[str] = None,
                  output_site_rates_dir: Optional[str] = None, output_likelihood_dir: Optional[str] = None,
                  remake=False, quantization_grid_center=0.03, quantization_grid_step=1.1,
                  quantization_grid_num_steps=64, verbose=True, seed=1234) -> None:
    """The reference's stage function (same keywords, same files): per family `<tree_dir>/<family>.txt`
    (CherryML tree format), `<site_rates_dir>/<family>.txt`, `<likelihood_dir>/<family>.txt` ("0.0", as
    the reference) and `<tree_dir>/<family>.profiling`.  `num_processes` / `remake` are accepted and
    ignored: the reference's pool over families is ONE batched device call here (`cb_ble_batch`).  Keyword arguments only, and with a cache
    directory set the three output directories default into the cache and the call returns
    {"output_tree_dir": ..., ...} -- the reference's caching convention, which is how the end-to-end
    pipelines use it as their `tree_estimator`."""
    from ..io import read_rate_matrix
    if output_tree_dir is None or output_site_rates_dir is None or output_likelihood_dir is None:
        raise ValueError("output_tree_dir, output_site_rates_dir and output_likelihood_dir are required")
    for d in (output_tree_dir, output_site_rates_dir, output_likelihood_dir):
        os.makedirs(d, exist_ok=True)
    rm = read_rate_matrix(rate_matrix_path)
    alphabet = [str(c) for c in rm.columns]
    Q = rm.to_numpy()
    import torch
    dev = torch.cuda.current_device()
    st = time.time()
    msas = [_read_msa_file(os.path.join(msa_dir, family + ".txt")) for family in families]
    prof: Dict[str, float] = {}
    results = fast_cherries_families(
        msas, Q, alphabet, num_rate_categories=num_rate_categories, max_iters=max_iters, seed=seed,
        quantization_grid_center=quantization_grid_center, quantization_grid_step=quantization_grid_step,
        quantization_grid_num_steps=quantization_grid_num_steps, device=dev, profile=prof)
    per_family = (time.time() - st) / max(len(families), 1)
    for family, (names, _), (cherries, lengths, site_rates) in zip(families, msas, results):
        write_tree(cherries_to_tree(names, cherries, lengths), os.path.join(output_tree_dir, family + ".txt"))
        with open(os.path.join(output_site_rates_dir, family + ".txt"), "w") as f:
            f.write(f"{len(site_rates)} sites\n" + " ".join(repr(float(r)) for r in site_rates))
        with open(os.path.join(output_likelihood_dir, family + ".txt"), "w") as f:
            f.write(str(0.0))
        with open(os.path.join(output_tree_dir, family + ".profiling"), "w") as f:   # (all families run as ONE batch: shares)
            f.write(f"pairing_time: {prof.get('pairing_time', 0.0) / max(len(families), 1)}\n"
                    f"ble_time: {prof.get('ble_time', 0.0) / max(len(families), 1)}\ntotal_time: {per_family}")
