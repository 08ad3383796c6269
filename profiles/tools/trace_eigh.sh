# kernel timeline of a bench run: where does the warm-started eigensolver spend its time (kernels and gaps)?
# usage: bash profiles/tools/trace_eigh.sh <steps> <tag>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$2 -- python3 $R/bench.py --steps $1 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/$2.log 2>&1
ls $R/gpurun_out/$2/*/
