#!/bin/bash
# usage: r5_co_exp.sh "<flags1>" ... : the co-counting pass with each experiment build ("" = shipped)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in "$@"; do
  export CB_EXTRA_HIPCC_FLAGS="$f"
  python3 -c "from cherryml_amd import _build; _build.build()" > gpurun_out/exp_build.log 2>&1 || { tail -5 gpurun_out/exp_build.log; continue; }
  python3 bench.py --workload co_counting --no-cpu-baseline > gpurun_out/exp_co.json 2> gpurun_out/exp_co.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/exp_co.json').read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step', round(d['ms_per_step'],4), d['phase_ms'], d['counted_events'])" "[$f]" || tail -3 gpurun_out/exp_co.err
done
unset CB_EXTRA_HIPCC_FLAGS
python3 -c "from cherryml_amd import _build; _build.build()" >> gpurun_out/exp_build.log 2>&1
