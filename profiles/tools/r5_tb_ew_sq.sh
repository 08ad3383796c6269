# SQ counters of the time-basis kernels (two passes of 8), run ON THE GPU BOX from the repo root: bash profiles/tools/r5_tb_ew_sq.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/tb_sq1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/tb_sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("tb_sq1", "tb_sq2"):
    fs = sorted(glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        for key in ("tb_ew", "k1_pt_loss_gt", "k2_t_eq_g_u", "k3_w_phi"):
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
