#!/bin/bash
# Round 5: eight-wave tiles (two K-groups, CB_BANK_KG=2) against four-wave tiles (CB_BANK_KG=1), fused and separate launches,
# on the full bench bank (129 buckets), the reference's real bank (43) and rank 0's share of an 8-rank job (17).
# Run ON THE GPU BOX from the repo root:  bash profiles/tools/r5_kg.sh [tag]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=${1:-r5b}
export CB_TEST_HOOKS=1
O=gpurun_out
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phase_ms']; print(sys.argv[2], 'ms/step', round(d['ms_per_step'],4), 'eigh', round(p['eigh'],4), 'bank', round(p['k1']+p['k2']+p['k3'],4), 'k4', round(p['k4'],4), 'loss', d['final_loss'])" "$1" "$2" 2>/dev/null || { echo "$2: FAILED"; tail -3 ${1%.json}.err; }; }
for kg in 2 1; do
  export CB_BANK_KG=$kg
  for mode in fused unfused; do
    if [ $mode = unfused ]; then export CB_BANK_UNFUSED=1; else unset CB_BANK_UNFUSED; fi
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/${TAG}_full_kg${kg}_$mode.json 2> $O/${TAG}_full_kg${kg}_$mode.err
    line $O/${TAG}_full_kg${kg}_$mode.json "full129 kg$kg $mode window"
    python3 bench.py --workload coevo400_demo --no-cpu-baseline --no-secondary > $O/${TAG}_demo_kg${kg}_$mode.json 2> $O/${TAG}_demo_kg${kg}_$mode.err
    line $O/${TAG}_demo_kg${kg}_$mode.json "demo43 kg$kg $mode 200 epochs"
    python3 bench.py --shard-of 8 --no-cpu-baseline --no-secondary > $O/${TAG}_shard8_kg${kg}_$mode.json 2> $O/${TAG}_shard8_kg${kg}_$mode.err
    line $O/${TAG}_shard8_kg${kg}_$mode.json "shard8 kg$kg $mode 200 epochs"
  done
done
unset CB_BANK_UNFUSED
# tile timelines of the eight-wave form (diagnostic build)
export CB_EXTRA_HIPCC_FLAGS=-DCB_CLOCK_STAMP
python3 -c "from cherryml_amd import _build; _build.build()" > $O/${TAG}_stamp_build.log 2>&1
export CB_BANK_KG=2
for cfg in "coevo400 8" "coevo400_demo 0" "coevo400 0"; do
  set -- $cfg
  for mode in fused unfused; do
    if [ $mode = unfused ]; then export CB_BANK_UNFUSED=1; else unset CB_BANK_UNFUSED; fi
    python3 profiles/tools/clock_probe.py 60 $1 $2 > $O/${TAG}_clock_$1_$2_$mode.json 2>> $O/${TAG}_stamp_build.log
    python3 profiles/tools/stamp_timeline.py $O/clock_stamps_60_epochs.npy 10 > $O/${TAG}_timeline_$1_$2_$mode.txt
  done
done
unset CB_BANK_UNFUSED CB_BANK_KG
unset CB_EXTRA_HIPCC_FLAGS
python3 -c "from cherryml_amd import _build; _build.build()" >> $O/${TAG}_stamp_build.log 2>&1
head -4 $O/${TAG}_timeline_*.txt
