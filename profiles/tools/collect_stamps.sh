#!/bin/bash
# the tile-timeline part of profiles/collect.sh alone (diagnostic build -DCB_CLOCK_STAMP; the shipped build is restored afterwards)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
TAG=${1:-final}
export CB_TEST_HOOKS=1
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
cat gpurun_out/${TAG}_tile_timeline_fused.txt
