R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R && python -m pytest tests/test_gpu_counting.py -x -q 2>&1 | tail -5 > $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_co -- python3 $R/bench.py --workload co_counting --no-cpu-baseline > $O/co_bench.log 2>&1
cd $R && CB_EXTRA_HIPCC_FLAGS=-DCB_CO_PLAIN python -c "from cherryml_amd import _build; _build.build(force=True)" > $O/build.log 2>&1
python3 $R/bench.py --workload co_counting --no-cpu-baseline > $O/co_bench_plain.log 2>&1
tail -3 $O/pytest.log; tail -1 $O/co_bench.log | cut -c1-400; tail -1 $O/co_bench_plain.log | cut -c1-400
cd $R && python -c "from cherryml_amd import _build; _build.build(force=True)" > /dev/null 2>&1   # leave the UNFLAGGED library behind
