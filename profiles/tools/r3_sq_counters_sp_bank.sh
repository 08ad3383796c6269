# round 3: cfg-4 / mixed parity tests, kernel trace of co_counting, SQ counters of sp_bank (siterm)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R && python -m pytest tests/test_gpu_siterm_cfg4.py tests/test_gpu_s400_full.py tests/test_gpu_likelihood.py -x -q -s 2>&1 | tail -40 > $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_co -- python3 $R/bench.py --workload co_counting --no-cpu-baseline > $O/co_bench.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $O/pmc_sq3 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find $O -name "*.csv" | head -30
tail -5 $O/pytest.log
