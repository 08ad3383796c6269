# round 4: SiteRM bench line, cfg-4 parity tests, SQ counters of sp_bank after the LDS re-striding (three passes of 8 counters)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4sq
mkdir -p $O
cd $R && python -m pytest tests/test_gpu_siterm_cfg4.py tests/test_gpu_bank.py -x -q 2>&1 | tail -5 > $O/pytest.log
python3 $R/bench.py --workload siterm --no-cpu-baseline > $O/siterm_bench.json 2> $O/siterm_bench.err
python3 $R/bench.py --workload lg20 --no-cpu-baseline > $O/lg20_bench.json 2> $O/lg20_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq3 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R && python3 profiles/tools/sq_counters_to_json.py $O/r04_sp_bank_sq_counters.json "rocprofv3 --pmc SQ_* (three passes), profiles/tools/r4_sq_counters_sp_bank.sh, bench.py --workload siterm --steps 3 --warmup 1; round 4, after the LDS re-striding of sp_bank (frames 36, tables 100 doubles)" $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cat $O/pytest.log; tail -c 600 $O/siterm_bench.json; tail -c 400 $O/lg20_bench.json
