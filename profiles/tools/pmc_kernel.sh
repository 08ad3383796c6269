# SQ counters of one bench workload, summarised per kernel (run on the GPU box from the repo root):
#   bash profiles/tools/pmc_kernel.sh <workload> <kernel substring>
R=${GRAFT_REPO_ROOT:-$PWD}
W=${1:-likelihood}; K=${2:-tl_mfma}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmck_a $R/gpurun_out/pmck_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmck_a -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmck_b -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 - "$K" $R/gpurun_out/pmck_a $R/gpurun_out/pmck_b <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
for d in sys.argv[2:]:
    tot = collections.Counter(); n = collections.Counter()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c in sorted(tot):
        print(f"{k} {c}: total {tot[c]:.4g} over {n[c]} launches")
PY
