#!/bin/bash
for f in "" "-DTB_NO_BWD" "-DTB_NO_FWD" "-DTB_NO_LOG" "-DTB_NO_LOAD" "-DTB_NO_BWD -DTB_NO_FWD" "-DTB_NO_BWD -DTB_NO_FWD -DTB_NO_LOG"; do
  export CB_EXTRA_HIPCC_FLAGS="$f"
  python3 -c "from cherryml_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', '| ms', round(d['ms_per_step'],4), 'k2 phase (EW + K2)', d['phase_ms']['k2'])"
done
unset CB_EXTRA_HIPCC_FLAGS
