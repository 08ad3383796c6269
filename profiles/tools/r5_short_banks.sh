#!/bin/bash
# Round 5, item 1: the bank launch on SHORT banks -- the reference's real 43-bucket bank and rank 0's 17-bucket share of an
# 8-rank job -- fused (k123_bank) against the three separate launches, bench lines + in-kernel tile timelines.
# Run ON THE GPU BOX from the repo root:  bash profiles/tools/r5_short_banks.sh [tag]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=${1:-r5a}
export CB_TEST_HOOKS=1   # CB_BANK_UNFUSED / CB_BANK_KG are test hooks (csrc/cb_internal.hip.h)
O=gpurun_out
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'ms/step', round(d['ms_per_step'],4), d['phase_ms'], d.get('eigh'))" "$1" "$2" 2>/dev/null || { echo "$2: FAILED"; tail -3 ${1%.json}.err; }; }
for mode in fused unfused; do
  if [ $mode = unfused ]; then export CB_BANK_UNFUSED=1; else unset CB_BANK_UNFUSED; fi
  python3 bench.py --workload coevo400_demo --no-cpu-baseline --no-secondary > $O/${TAG}_demo_$mode.json 2> $O/${TAG}_demo_$mode.err
  line $O/${TAG}_demo_$mode.json "demo43 $mode 200 epochs"
  python3 bench.py --workload coevo400_demo --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/${TAG}_demo_w_$mode.json 2> $O/${TAG}_demo_w_$mode.err
  line $O/${TAG}_demo_w_$mode.json "demo43 $mode window"
  python3 bench.py --shard-of 8 --no-cpu-baseline --no-secondary > $O/${TAG}_shard8_$mode.json 2> $O/${TAG}_shard8_$mode.err
  line $O/${TAG}_shard8_$mode.json "shard8 $mode 200 epochs"
  python3 bench.py --shard-of 8 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/${TAG}_shard8_w_$mode.json 2> $O/${TAG}_shard8_w_$mode.err
  line $O/${TAG}_shard8_w_$mode.json "shard8 $mode window"
  python3 bench.py --workload coevo400_demo --shard-of 8 --no-cpu-baseline --no-secondary > $O/${TAG}_demoshard8_$mode.json 2> $O/${TAG}_demoshard8_$mode.err
  line $O/${TAG}_demoshard8_$mode.json "demo shard8 (6 buckets) $mode 200 epochs"
done
unset CB_BANK_UNFUSED
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/${TAG}_full_w.json 2> $O/${TAG}_full_w.err
line $O/${TAG}_full_w.json "full129 fused window"
# tile timelines (diagnostic build)
export CB_EXTRA_HIPCC_FLAGS=-DCB_CLOCK_STAMP
python3 -c "from cherryml_amd import _build; _build.build()" > $O/${TAG}_stamp_build.log 2>&1
for cfg in "coevo400 8" "coevo400_demo 0"; do
  set -- $cfg
  for mode in fused unfused; do
    if [ $mode = unfused ]; then export CB_BANK_UNFUSED=1; else unset CB_BANK_UNFUSED; fi
    python3 profiles/tools/clock_probe.py 60 $1 $2 > $O/${TAG}_clock_$1_$2_$mode.json 2>> $O/${TAG}_stamp_build.log
    python3 profiles/tools/stamp_timeline.py $O/clock_stamps_60_epochs.npy 10 > $O/${TAG}_timeline_$1_$2_$mode.txt
  done
done
unset CB_BANK_UNFUSED
unset CB_EXTRA_HIPCC_FLAGS
python3 -c "from cherryml_amd import _build; _build.build()" >> $O/${TAG}_stamp_build.log 2>&1
head -4 $O/${TAG}_timeline_*.txt
