#!/bin/bash
# usage: r4_exp2.sh "<hipcc flags>|<env assignments>" ... : bench the co-evolution epoch with each experiment build / environment
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for spec in "$@"; do
  f="${spec%%|*}"; e="${spec#*|}"
  export CB_EXTRA_HIPCC_FLAGS="$f"
  python -c "from cherryml_amd import _build; _build.build()" > gpurun_out/exp_build.log 2>&1 || { tail -5 gpurun_out/exp_build.log; continue; }
  env $e python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/exp.json 2> gpurun_out/exp.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/exp.json').read().strip().splitlines()[-1]); print(sys.argv[1], round(d['ms_per_step'],4), d['phase_ms'])" "$spec" || tail -3 gpurun_out/exp.err
done
