#!/bin/bash
# per-solve records of the planned eigensolver over the driver's window and a 200-epoch run (CB_DEBUG=1 prints one line per
# solve: M = masked / L = all pairs, polynomial order, d = damped, squarings, cosine, generator row sums): which launches of a
# plan ran, which returned at once.   bash profiles/tools/r5_eigh_records.sh  (ON THE GPU BOX)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
CB_DEBUG=1 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "planned eigh" > gpurun_out/r5_eigh_records_window.txt
CB_DEBUG=1 python3 bench.py --shard-of 8 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "planned eigh" > gpurun_out/r5_eigh_records_shard200.txt
wc -l gpurun_out/r5_eigh_records_*.txt; tail -25 gpurun_out/r5_eigh_records_window.txt
