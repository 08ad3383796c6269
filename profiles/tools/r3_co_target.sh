R=$GRAFT_REPO_ROOT
for T in 512 1024 2048 4096; do
  cd $R && CB_EXTRA_HIPCC_FLAGS=-DCO_TARGET=$T python -c "from cherryml_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  echo "target $T: $(python3 bench.py --workload co_counting --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['phase_ms'])")"
done
cd $R && python -c "from cherryml_amd import _build; _build.build(force=True)" > /dev/null 2>&1   # leave the UNFLAGGED library behind
