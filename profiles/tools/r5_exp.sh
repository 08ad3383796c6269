#!/bin/bash
# usage: r5_exp.sh "<flags1>" "<flags2>" ... : the three bank shapes (full 129 buckets, the reference's real 43, rank 0's share
# of an 8-rank job = 17) with each experiment build; "" = the shipped build.  Run ON THE GPU BOX from the repo root.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export CB_TEST_HOOKS=1
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phase_ms']; print(sys.argv[2], 'ms/step', round(d['ms_per_step'],4), 'eigh', round(p['eigh'],4), 'bank', round(p['k1']+p['k2']+p['k3'],4), 'k4', round(p['k4'],4), 'loss', d['final_loss'], d.get('eigh'))" "$1" "$2" 2>/dev/null || { echo "$2: FAILED"; tail -3 ${1%.json}.err; }; }
i=0
for f in "$@"; do
  i=$((i+1))
  export CB_EXTRA_HIPCC_FLAGS="$f"
  python3 -c "from cherryml_amd import _build; _build.build()" > gpurun_out/exp_build.log 2>&1 || { tail -5 gpurun_out/exp_build.log; continue; }
  python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/exp${i}_full.json 2> gpurun_out/exp${i}_full.err
  line gpurun_out/exp${i}_full.json "[$f] full129 window"
  python3 bench.py --no-secondary --no-cpu-baseline > gpurun_out/exp${i}_full200.json 2> gpurun_out/exp${i}_full200.err
  line gpurun_out/exp${i}_full200.json "[$f] full129 200 epochs"
  python3 bench.py --workload coevo400_demo --no-secondary --no-cpu-baseline > gpurun_out/exp${i}_demo.json 2> gpurun_out/exp${i}_demo.err
  line gpurun_out/exp${i}_demo.json "[$f] demo43 200 epochs"
  python3 bench.py --shard-of 8 --no-secondary --no-cpu-baseline > gpurun_out/exp${i}_shard.json 2> gpurun_out/exp${i}_shard.err
  line gpurun_out/exp${i}_shard.json "[$f] shard8 200 epochs"
  python3 bench.py --shard-of 8 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/exp${i}_shardw.json 2> gpurun_out/exp${i}_shardw.err
  line gpurun_out/exp${i}_shardw.json "[$f] shard8 window"
done
unset CB_EXTRA_HIPCC_FLAGS
python3 -c "from cherryml_amd import _build; _build.build()" >> gpurun_out/exp_build.log 2>&1
