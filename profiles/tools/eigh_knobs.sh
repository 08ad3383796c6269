# sweep of the hybrid eigensolver's schedule knobs (environment switches) on the bench bank:
# eigh ms per epoch over 200 epochs and over the driver's 20
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  a=$(env "$@" python bench.py --steps 200 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['phase_ms']['eigh'],4), round(d['ms_per_step'],4), d['final_loss'])")
  b=$(env "$@" python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['phase_ms']['eigh'],4), round(d['ms_per_step'],4))")
  echo "$label | 200 epochs: $a | 20 epochs: $b"
}
run default X=1
run band2 CB_HYBRID_BAND=2
run band4 CB_HYBRID_BAND=4
run within1 CB_HYBRID_WITHIN=1
run within3 CB_HYBRID_WITHIN=3
run trig1e-4 CB_LIGHT_TRIGGER=1e-4
run trig1e-3 CB_LIGHT_TRIGGER=1e-3
run nsfrom3 CB_HYBRID_NS_FROM=3
run reps2 CB_HYBRID_REPS=2
run band2within1 CB_HYBRID_BAND=2 CB_HYBRID_WITHIN=1
