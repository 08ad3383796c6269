#!/bin/bash
# round 4: the clock held inside K1..K3 (in-kernel stamps, diagnostic build) and GRBM_GUI_ACTIVE per dispatch (shipped build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export CB_EXTRA_HIPCC_FLAGS=-DCB_CLOCK_STAMP
python -c "from cherryml_amd import _build; _build.build()" > gpurun_out/r4_clock_build.log 2>&1
python profiles/tools/clock_probe.py 200 > gpurun_out/r4_clock_probe.json 2> gpurun_out/r4_clock_probe.err
unset CB_EXTRA_HIPCC_FLAGS
python -c "from cherryml_amd import _build; _build.build()" >> gpurun_out/r4_clock_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4_grbm
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r4_grbm -- \
  python3 $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/gpurun_out/r4_grbm_bench.json 2> $R/gpurun_out/r4_grbm.err
cd $R
find gpurun_out/r4_grbm -name "*.csv" | head
