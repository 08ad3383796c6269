#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root.  Writes raw rocprofv3 output
# under gpurun_out/ (scratch); profiles/parse_profiles.py turns it into the
# committed summaries.  PMC passes are separate from --kernel-trace/--stats runs
# and FETCH_SIZE / WRITE_SIZE are collected in separate passes (TCC slot budget),
# as MI355X_MICROARCH.md prescribes.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-final}
# plain (unprofiled) default bench line first, on the fresh box, as the driver runs it
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_default.log 2>&1
for w in coevo400 lg20 siterm counting ble assembly likelihood; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$w -- \
    python3 $R/bench.py --workload $w --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_bench_$w.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_${w}_$c -- \
      python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  done
done
