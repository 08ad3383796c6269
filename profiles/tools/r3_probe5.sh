R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s64 -- python3 $R/bench.py --workload siterm --no-cpu-baseline > $O/s64.log 2>&1
export CB_SPP256=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s256 -- python3 $R/bench.py --workload siterm --no-cpu-baseline > $O/s256.log 2>&1
for f in s64 s256; do tail -1 $O/$f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('phase_ms'))"; head -4 $O/trace_$f/runc/*_kernel_stats.csv | cut -c1-120; done
