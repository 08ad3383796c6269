for lim in 0.075 0.3 0.5 1.0; do
  for rep in 1 2; do
    CB_POLY_LIM_MASKED=$lim python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('lim', '$lim', 'ms_per_step', round(d['ms_per_step'],4))"
  done
  CB_POLY_LIM_MASKED=$lim python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('lim', '$lim', '200 epochs ms_per_step', round(d['ms_per_step'],4))"
done
