# round 6: SQ counters of the SiteRM kernels (three passes of <= 8 counters), run ON THE GPU BOX from the repo root
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6sq
mkdir -p $O
cd $R && python3 $R/bench.py --workload siterm --no-cpu-baseline > $O/siterm_bench.json 2> $O/siterm_bench.err < /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --workload siterm --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq3 -- python3 $R/bench.py --workload siterm --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
cd $R && python3 profiles/tools/sq_counters_to_json.py $O/r06_sp_bank_sq_counters.json "rocprofv3 --pmc SQ_* (three passes), profiles/tools/r6_sq_counters_sp_bank.sh, bench.py --workload siterm --steps 3 --warmup 1; round 6" $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3
f=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r06_siterm_kernel_stats.csv && head -6 "$f" | cut -c1-150
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
tail -c 700 $O/siterm_bench.json
