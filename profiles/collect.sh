#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/collect.sh [tag]
# Writes raw rocprofv3 output under gpurun_out/ (scratch); profiles/parse_profiles.py turns it into the
# committed summaries (profiles/<round>_<tag>_*).  PMC passes are separate from --kernel-trace/--stats
# runs; FETCH_SIZE / WRITE_SIZE are collected in separate passes (TCC slot budget) and the SQ counters
# (MFMA busy cycles, wave cycles, waits) in one pass of 8, as MI355X_MICROARCH.md prescribes.
set -u
export CB_TEST_HOOKS=1   # CB_BANK_UNFUSED / CB_BANK_KG are test hooks (csrc/cb_internal.hip.h)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-final}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
# plain (unprofiled) lines first, on the fresh box, as the driver runs them
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench_driver.log 2>&1
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_default.log 2>&1
for v in "coevo400 f64" "coevo400 f64 perbucket" "coevo400 mixed" "coevo400 f32" "coevo400_demo f64" "coevo400 f64 shard8" "lg20 f64" "siterm f64" "counting f64" "co_counting f64" "ble f64" "assembly f64" "likelihood f64"; do
  set -- $v; w=$1; dt=$2; name=$w; [ "$dt" != "f64" ] && name=${w}_$dt
  extra=""; if [ "${3:-}" = "shard8" ]; then name=${w}_shard8; extra="--shard-of 8"; fi   # rank 0's share of an 8-rank job on this GPU
  # the same bank with three products per bucket (test hook CB_BANK_TB=0): the form the time basis replaced
  unset CB_BANK_TB; if [ "${3:-}" = "perbucket" ]; then name=${w}_perbucket; export CB_BANK_TB=0; fi
  # the bench line of this workload UNPROFILED (kernel tracing adds a few percent to launch-bound epochs) ...
  python3 $R/bench.py --workload $w --dtype $dt $extra --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_bench_$name.log 2>&1
  # ... and the same command under the kernel trace for the per-kernel statistics
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$name -- \
    python3 $R/bench.py --workload $w --dtype $dt $extra --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_benchprof_$name.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_${name}_$c -- \
      python3 $R/bench.py --workload $w --dtype $dt $extra --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  done
done
unset CB_BANK_TB
for dt in f64 mixed f32; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA \
    --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq_$dt -- \
    python3 $R/bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
done
# the three bank products as separate launches (CB_BANK_UNFUSED=1): the per-kernel reference of the fused launch
export CB_BANK_UNFUSED=1
python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench_coevo400_unfused.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_coevo400_unfused -- \
  python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_benchprof_coevo400_unfused.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA \
  --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq_f64_unfused -- \
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
unset CB_BANK_UNFUSED
# in-kernel stamps of the bank's tiles (diagnostic build, -DCB_CLOCK_STAMP): tile timelines of the fused launch and of the three
# separate ones, and the clock held inside the K loops; the shipped build is restored afterwards
cd $R
export CB_EXTRA_HIPCC_FLAGS=-DCB_CLOCK_STAMP
python3 -c "from cherryml_amd import _build; _build.build()" > gpurun_out/${TAG}_stamp_build.log 2>&1
python3 profiles/tools/clock_probe.py 60 > gpurun_out/${TAG}_clock_probe_fused.json 2>> gpurun_out/${TAG}_stamp_build.log
python3 profiles/tools/stamp_timeline.py gpurun_out/clock_stamps_60_epochs.npy 25 > gpurun_out/${TAG}_tile_timeline_fused.txt
export CB_BANK_UNFUSED=1
python3 profiles/tools/clock_probe.py 60 > gpurun_out/${TAG}_clock_probe_separate.json 2>> gpurun_out/${TAG}_stamp_build.log
python3 profiles/tools/stamp_timeline.py gpurun_out/clock_stamps_60_epochs.npy 25 > gpurun_out/${TAG}_tile_timeline_separate.txt
unset CB_BANK_UNFUSED
unset CB_EXTRA_HIPCC_FLAGS
python3 -c "from cherryml_amd import _build; _build.build()" >> gpurun_out/${TAG}_stamp_build.log 2>&1
# round 6: SiteRM SQ counters (three passes) and the eigensolver's launch-by-launch timeline without a tracer (diagnostic build)
cd $R && bash profiles/tools/r6_sq_counters_sp_bank.sh > gpurun_out/${TAG}_r6sq.log 2>&1
export CB_EXTRA_HIPCC_FLAGS=-DCB_EIGH_STAMPS
python3 -c "from cherryml_amd import _build; _build.build()" > gpurun_out/${TAG}_eigh_stamp_build.log 2>&1
CB_DEBUG=1 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2> gpurun_out/${TAG}_eigh_stamps_window.txt > /dev/null < /dev/null
unset CB_EXTRA_HIPCC_FLAGS
python3 -c "from cherryml_amd import _build; _build.build()" >> gpurun_out/${TAG}_eigh_stamp_build.log 2>&1
python3 profiles/tools/r6_phase_event_cost.py > gpurun_out/${TAG}_phase_event_cost.json 2> /dev/null < /dev/null
