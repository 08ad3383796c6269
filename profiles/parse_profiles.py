#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>_* (written by profiles/collect.sh) into profiles/:
  r01_<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  r01_<tag>_<workload>_bench.json         the bench line of that profiled run
  pmc_traffic.json                        HBM-side bytes per launch per kernel:
        2 x FETCH_SIZE + WRITE_SIZE  [KB -> bytes]
     (gfx950: FETCH_SIZE reports half of the bytes of a streaming read -- calibrated here on
      lg_transpose_pad, which reads exactly the 165.12 MB count tensor once; WRITE_SIZE is exact)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "final"
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
out = {"bytes_per_launch": {}, "detail": {}, "calibration": {},
       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), gpurun_out/{tag}_pmc_*"}
names = {"k1_pt_loss_gt": "k1_pt_loss_gt", "k2_t_eq_g_u": "k2_t_eq_g_u", "k3_w_phi": "k3_w_phi",
         "lgj_round": "lgj_round", "lg_transpose_pad": "lg_transpose_pad",
         "small_train_kernel": "small_train_kernel", "small_bank_kernel": "small_bank_kernel",
         "lg_prepare": "lg_prepare", "lg_bank": "lg_bank", "lg_finish": "lg_finish",
         "count_transitions_lds_kernel": "count_transitions_lds_kernel",
         "count_reduce_slabs": "count_reduce_slabs", "k3_reduce": "k3_reduce", "k4_gemm": "k4_gemm",
         "sp_prepare": "sp_prepare", "sp_bank": "sp_bank", "sp_finish": "sp_finish", "sg_gemm": "sg_gemm", "lgx_build": "lgx_build", "ble_branch_lengths_kernel": "ble_branch_lengths_kernel",
         "ble_site_rates_kernel": "ble_site_rates_kernel", "siterm_raw_counts_kernel": "siterm_raw_counts_kernel",
         "siterm_mix_kernel": "siterm_mix_kernel"}
for w in ["coevo400", "lg20", "siterm", "counting", "ble", "assembly", "likelihood"]:
    st = glob.glob(f"{root}/gpurun_out/{tag}_trace_{w}/*/*_kernel_stats.csv")
    if st:
        shutil.copy(st[0], f"{here}/r01_{tag}_{w}_kernel_stats.csv")
    log = f"{root}/gpurun_out/{tag}_bench_{w}.log"
    if os.path.exists(log):
        line = [l for l in open(log) if l.startswith("{")]
        if line:
            open(f"{here}/r01_{tag}_{w}_bench.json", "w").write(line[-1])
    vals = collections.defaultdict(dict)
    for c in ["FETCH_SIZE", "WRITE_SIZE"]:
        fs = glob.glob(f"{root}/gpurun_out/{tag}_pmc_{w}_{c}/*/*counter_collection.csv")
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for short in names:
                if short in k:
                    vals[short][c] = sum(v) / len(v)
                    vals[short][c + "_launches"] = len(v)
    for k, d in vals.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            key = k if w == "coevo400" else f"{k}:{w}"
            b = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
            out["detail"][key] = {"fetch_KB_raw": d["FETCH_SIZE"], "write_KB": d["WRITE_SIZE"],
                                  "bytes": b, "launches_seen": d["FETCH_SIZE_launches"]}
            out["bytes_per_launch"][key] = b
if "lg_transpose_pad" in out["detail"]:
    d = out["detail"]["lg_transpose_pad"]
    out["calibration"] = {"kernel": "lg_transpose_pad", "true_read_bytes": 129 * 400 * 400 * 8,
                          "FETCH_SIZE_x1024": d["fetch_KB_raw"] * 1024,
                          "ratio": 129 * 400 * 400 * 8 / (d["fetch_KB_raw"] * 1024)}
# per-epoch / per-pass totals of the workloads whose step is more than one launch
bpl = out["bytes_per_launch"]
for w in ("siterm", "lg20"):   # S <= 20: three launches per epoch (sp_prepare / sp_bank / sp_finish)
    if all(f"{k}:{w}" in bpl for k in ("sp_prepare", "sp_bank", "sp_finish")):
        bpl[f"epoch:{w}"] = sum(bpl[f"{k}:{w}"] for k in ("sp_prepare", "sp_bank", "sp_finish"))
if all(f"{k}:counting" in bpl for k in ("count_transitions_lds_kernel", "count_reduce_slabs")):
    bpl["pass:counting"] = bpl["count_transitions_lds_kernel:counting"] + bpl["count_reduce_slabs:counting"]
json.dump(out, open(f"{here}/pmc_traffic.json", "w"), indent=1)
d = f"{root}/gpurun_out/{tag}_bench_default.log"
if os.path.exists(d):
    line = [l for l in open(d) if l.startswith("{")]
    if line:
        open(f"{here}/r01_{tag}_bench_default.json", "w").write(line[-1])
print(json.dumps(out["bytes_per_launch"], indent=1))
