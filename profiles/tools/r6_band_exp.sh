# Round 6: build-time variants of the planned eigensolve's band passes, driver window + 200 epochs, records of the window
# solves (CB_DEBUG) for the variants that stall.  Run on the GPU box from the repo root: bash profiles/tools/r6_band_exp.sh
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/r6c
one() {
  export CB_EXTRA_HIPCC_FLAGS="$1"
  python3 -c "from cherryml_amd import _build; _build.build()" > gpurun_out/r6c/build.log 2>&1 || { echo BUILD FAILED; tail gpurun_out/r6c/build.log; return; }
  for rep in 1 2; do
    python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('FLAGS[$1] window ms', round(d['ms_per_step'],4), 'eigh', d['phase_ms']['eigh'], d['eigh'])"
  done
  python3 bench.py --steps 200 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('FLAGS[$1] 200ep ms', round(d['ms_per_step'],4), 'eigh', d['phase_ms']['eigh'], d['eigh'])"
  if [ -n "${2:-}" ]; then
    CB_DEBUG=1 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "planned eigh" | tail -n 25 > gpurun_out/r6c/records_$2.txt
  fi
}
for v in "$@"; do one "$v" "$(echo $v | tr -c 'A-Za-z0-9=\n' '_')"; done
unset CB_EXTRA_HIPCC_FLAGS
python3 -c "from cherryml_amd import _build; _build.build()" > /dev/null 2>&1
