R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r2_counters.txt 2>&1
for dt in f64 f32; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/r2_pmc_sq_$dt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/r2_pmc_sq2_$dt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt > /dev/null 2>&1
done
ls $R/gpurun_out/r2_pmc_sq_f64/* | head
