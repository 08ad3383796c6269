run() { # label, env...
  lbl=$1; shift
  a=$(env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))")
  b=$(env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))")
  c=$(env "$@" python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))")
  echo "$lbl: driver $a $b ; 200 epochs $c"
}
run base CB_X=1
run band2 CB_HYBRID_BAND=2
run band4 CB_HYBRID_BAND=4
run band5 CB_HYBRID_BAND=5
run reps2 CB_HYBRID_REPS=2
run band4reps2 CB_HYBRID_BAND=4 CB_HYBRID_REPS=2
run within3 CB_HYBRID_WITHIN=3
run trig1e-3 CB_LIGHT_TRIGGER=1e-3
run trig1e-4 CB_LIGHT_TRIGGER=1e-4
