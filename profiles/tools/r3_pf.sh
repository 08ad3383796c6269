R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PF   ', d['ms_per_step'], d['phase_ms'])"; done
CB_EXTRA_HIPCC_FLAGS=-DCB_NO_PF python -c "from cherryml_amd import _build; _build.build(force=True)" > /dev/null 2>&1
for i in 1 2; do python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_PF', d['ms_per_step'], d['phase_ms'])"; done
