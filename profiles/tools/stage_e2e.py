#!/usr/bin/env python3
"""Wall time of the stage function quantized_transitions_mle on a co-evolution-sized problem (400 states,
129 buckets, sparse counts like the demo bank: an 84 MB count file), with the host-side parts itemised by
cProfile.  Run on the GPU box:  python profiles/tools/stage_e2e.py [epochs]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import cherryml_amd  # noqa: E402
from cherryml_amd.io import write_count_matrices, write_rate_matrix  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 500
AA = list("ARNDCQEGHILKMFPSTWYV")
states = [a + b for a in AA for b in AA]
rng = np.random.default_rng(0)
Q, pi, mask = bench.coevolution_truth(rng)
t, C = bench.reversible_bank(Q, pi, 1057194.0, rng)
C = np.round(C * (rng.random(C.shape) < 0.035) * 30.0) * 0.25          # sparse quarter-integer counts
C = 0.5 * (C + np.transpose(C, (0, 2, 1)))
with tempfile.TemporaryDirectory() as d:
    cpath, mpath, ipath, out = (os.path.join(d, n) for n in ("counts.txt", "mask.txt", "init.txt", "out"))
    write_count_matrices([(float("%.8f" % q), pd.DataFrame(C[b], index=states, columns=states)) for b, q in enumerate(t)], cpath)
    pd.DataFrame(mask.astype(int), index=states, columns=states).to_csv(mpath, sep=" ")
    write_rate_matrix(cherryml_amd.estimation.jtt_ipw_from_arrays(t, C, mask), states, ipath)
    print("count file: %.1f MB" % (os.path.getsize(cpath) / 1e6))
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    cherryml_amd.quantized_transitions_mle(count_matrices_path=cpath, initialization_path=ipath, mask_path=mpath,
                                           output_rate_matrix_dir=out, device="cuda", num_epochs=epochs)
    pr.disable()
    print("stage wall time: %.2f s for %d epochs" % (time.time() - t0, epochs))
    print(open(os.path.join(out, "profiling.txt")).read())
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
