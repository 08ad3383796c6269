run() { # name, env...
  name=$1; shift
  env "$@" CB_DEBUG=1 timeout 300 python bench.py --steps 20 --warmup 5 --workload coevo400 --no-cpu-baseline --no-secondary > gpurun_out/tune_$name.json 2> gpurun_out/tune_$name.log
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/tune_$name.json').read().strip().splitlines()[-1])
log=open('gpurun_out/tune_$name.log').read()
print('$name', round(d['ms_per_step'],4), d['phase_ms']['eigh'], 'stalls', log.count('STALL'))
PY
}
run so1
run noso1 CB_TUNE_NOSO=1
run so2
run noso2 CB_TUNE_NOSO=1
run so3
run noso3 CB_TUNE_NOSO=1
for v in so noso; do
  if [ $v = noso ]; then export CB_TUNE_NOSO=1; else unset CB_TUNE_NOSO; fi
  python bench.py --steps 200 --warmup 5 --workload coevo400 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long $v', d['ms_per_step'], d['phase_ms']['eigh'])"
done
