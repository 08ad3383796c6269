import sys, time
sys.path.insert(0, "tests")
import numpy as np
from test_gpu_likelihood import _random_tree, _random_msa, _random_pair_model
from test_oracle_golden import load_golden
from cherryml_amd.evaluation import dp_likelihood_computation, dp_likelihood_computation_batch
z = load_golden("likelihood.npz")
aa = [str(a) for a in z["amino_acids"]]
rng = np.random.default_rng(1)
pi2, Q2 = _random_pair_model(rng)
trees, msas, cms, rates = [], [], [], []
for f in range(32):
    tree, names = _random_tree(rng, 64)
    L = 100
    cm = np.zeros((L, L), dtype=int)
    perm = rng.permutation(L)
    for k in range(20):
        i, j = perm[2 * k], perm[2 * k + 1]
        cm[i, j] = cm[j, i] = 1
    trees.append(tree); msas.append(_random_msa(rng, names, L, aa)); cms.append(cm)
    rates.append(list(rng.choice([0.25, 0.5, 1.0, 2.0], size=L)))
for rep in range(2):
    t0 = time.time(); p = {}
    a = dp_likelihood_computation_batch(trees, msas, cms, rates, aa, z["pi_wag"], z["wag"], True, pi2, Q2, True, profile=p)
    t1 = time.time(); q = {}
    b = [dp_likelihood_computation(trees[f], msas[f], cms[f], rates[f], aa, z["pi_wag"], z["wag"], pi_2=pi2, Q_2=Q2, profile=q) for f in range(32)]
    t2 = time.time()
    print(f"32 families x (64 leaves, 60 sites + 20 pairs): batch {1e3*(t1-t0):.1f} ms wall ({p['kernel_ms']:.1f} ms GPU), "
          f"family by family {1e3*(t2-t1):.1f} ms wall ({q['kernel_ms']:.1f} ms GPU); equal: {all(np.array_equal(x[1], y[1], equal_nan=True) for x, y in zip(a, b))}")
