# kernel timeline of the driver's default run (20 timed epochs after 5 warm-up epochs): where does the
# warm-started eigensolver spend its time (kernels and gaps)?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2_trace20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/r2_trace20.log 2>&1
ls $R/gpurun_out/r2_trace20/*/
