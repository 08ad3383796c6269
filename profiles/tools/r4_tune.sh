run() { # name, env...
  name=$1; shift
  env "$@" CB_DEBUG=1 timeout 300 python bench.py --steps 20 --warmup 5 --workload coevo400 --no-cpu-baseline --no-secondary > gpurun_out/tune_$name.json 2> gpurun_out/tune_$name.log
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/tune_$name.json').read().strip().splitlines()[-1])
import re
log=open('gpurun_out/tune_$name.log').read()
print('$name', round(d['ms_per_step'],4), d['phase_ms']['eigh'], 'stalls', log.count('STALL'))
PY
}
run base
run base2
run w1_1 CB_TUNE_W1=1
run w_11 CB_TUNE_W0=1 CB_TUNE_W1=1
for i in 1 2; do env CB_DEBUG=1 python bench.py --steps 200 --warmup 5 --workload coevo400 --no-cpu-baseline --no-secondary > gpurun_out/tune_long$i.json 2> gpurun_out/tune_long$i.log; done
python3 - <<PY
import json
for i in (1,2):
  d=json.loads(open('gpurun_out/tune_long%d.json'%i).read().strip().splitlines()[-1])
  log=open('gpurun_out/tune_long%d.log'%i).read()
  print('long', round(d['ms_per_step'],4), d['phase_ms']['eigh'], 'stalls', log.count('STALL'))
PY
timeout 600 python -m pytest tests/test_gpu_s400_full.py -x -q 2>&1 | tail -3
