R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R && python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python3 $R/bench.py --workload lg20 --no-cpu-baseline > $O/lg20.log 2>&1
python3 $R/bench.py --workload siterm --no-cpu-baseline > $O/siterm.log 2>&1
python3 $R/bench.py --workload co_counting --no-cpu-baseline > $O/co.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_lg -- python3 $R/bench.py --workload lg20 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_siterm -- python3 $R/bench.py --workload siterm --no-cpu-baseline > /dev/null 2>&1
tail -4 $O/pytest.log; for f in lg20 siterm co; do tail -1 $O/$f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('phase_ms'))"; done
