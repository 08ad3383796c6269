#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>_* (written by profiles/collect.sh) into profiles/, named per round:
  <round>_<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  <round>_<tag>_<workload>_bench.json         the bench line of that profiled run
  <round>_<tag>_bench_default.json / _bench_driver.json   unprofiled lines (default flags; the driver's --steps 20 --warmup 5)
  pmc_traffic.json       HBM-side bytes per launch per kernel: 2 x FETCH_SIZE + WRITE_SIZE  [KB -> bytes]
                         (gfx950: FETCH_SIZE reports half of the bytes of a streaming read -- calibrated here on
                         lg_transpose_pad, which reads exactly the 165.12 MB count tensor once; WRITE_SIZE is exact)
  mfma_util.json         MFMA utilisation of the bank kernels: SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA pipe is busy,
                         summed over the 1024 SIMDs) / (1024 x kernel duration x 2.4 GHz), with the wave-cycle split
                         (SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES)
usage: python profiles/parse_profiles.py [tag] [round]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

def newest(pattern):
    """the most recent match (gpurun merges every call's files into gpurun_out/: an older run of the same tag may linger)"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


tag = sys.argv[1] if len(sys.argv) > 1 else "final"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
import datetime
STAMP = {"round": rnd, "collected": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d")}
out = {**STAMP, "bytes_per_launch": {}, "detail": {}, "calibration": {},
       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), gpurun_out/{tag}_pmc_*"}
NAMES = ["k123_bank", "k1_pt_loss_gt", "k2_t_eq_g_u", "k3_w_phi", "lgj_round", "lg_transpose_pad", "small_train_kernel", "small_bank_kernel",
         "lg_prepare", "lg_bank", "lg_finish", "count_transitions_lds_kernel", "count_reduce_slabs", "k3_reduce", "sp_prepare",
         "sp_bank", "sp_finish", "sp_step", "sg_gemm", "co_bucket_kernel", "co_plan_kernel", "co_expand_kernel", "co_count_lds_kernel", "lgx_build", "ble_branch_lengths_kernel", "ble_site_rates_kernel",
         "siterm_raw_counts_kernel", "siterm_mix_kernel", "tl_mfma_kernel", "tl_leaf_kernel", "tl_group_kernel", "lg_cast_f32", "lge_gram", "lge_gemm", "lge_so", "lge_p34", "lge_plain", "tl_leaf_mfma_kernel",
         "jtt_stats_partial", "ky_reduce_loss", "kphi_combine", "k3_reduce_loss", "tb_ew", "tb_tables"]
WORKLOADS = ["coevo400", "coevo400_perbucket", "coevo400_mixed", "coevo400_f32", "coevo400_demo", "coevo400_shard8", "lg20", "siterm", "counting", "co_counting",
             "ble", "assembly", "likelihood"]


def kshort(full):
    """kernel family + element type of the templated bank kernels (k2_t_eq_g_u<float> -> k2_t_eq_g_u_f32)"""
    for short in NAMES:
        if short in full:
            if short in ("k123_bank", "k1_pt_loss_gt", "k2_t_eq_g_u", "k3_w_phi", "k3_reduce"):
                if "<float" in full:
                    return short + "_f32"
                if short in ("k1_pt_loss_gt", "k123_bank") and "<double, float" in full:
                    return short + "_mixed"
            return short
    return None


for w in WORKLOADS:
    st = newest(f"{root}/gpurun_out/{tag}_trace_{w}/*/*_kernel_stats.csv")
    if st:
        shutil.copy(st[0], f"{here}/{rnd}_{tag}_{w}_kernel_stats.csv")
    log = f"{root}/gpurun_out/{tag}_bench_{w}.log"
    if os.path.exists(log):
        line = [l for l in open(log) if l.startswith("{")]
        if line:
            open(f"{here}/{rnd}_{tag}_{w}_bench.json", "w").write(line[-1])
    vals = collections.defaultdict(dict)
    for c in ["FETCH_SIZE", "WRITE_SIZE"]:
        fs = newest(f"{root}/gpurun_out/{tag}_pmc_{w}_{c}/*/*counter_collection.csv")
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            short = kshort(k)
            if short:
                vals[short][c] = sum(v) / len(v)
                vals[short][c + "_launches"] = len(v)
    for k, d in vals.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            key = k if w in ("coevo400", "coevo400_mixed", "coevo400_f32") else f"{k}:{w}"
            b = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
            out["detail"][key] = {"fetch_KB_raw": d["FETCH_SIZE"], "write_KB": d["WRITE_SIZE"], "bytes": b,
                                  "launches_seen": d["FETCH_SIZE_launches"], "workload": w}
            out["bytes_per_launch"][key] = b
if "lg_transpose_pad" in out["detail"]:
    d = out["detail"]["lg_transpose_pad"]
    out["calibration"] = {"kernel": "lg_transpose_pad", "true_read_bytes": 129 * 400 * 400 * 8,
                          "FETCH_SIZE_x1024": d["fetch_KB_raw"] * 1024, "ratio": 129 * 400 * 400 * 8 / (d["fetch_KB_raw"] * 1024)}
bpl = out["bytes_per_launch"]
for w in ("siterm", "lg20"):   # S <= 20: three launches per epoch (sp_prepare / sp_bank / sp_finish), or two for a few sites
    if all(f"{k}:{w}" in bpl for k in ("sp_step", "sp_bank")):
        bpl[f"epoch:{w}"] = bpl[f"sp_step:{w}"] + bpl[f"sp_bank:{w}"]
    elif all(f"{k}:{w}" in bpl for k in ("sp_prepare", "sp_bank", "sp_finish")):
        bpl[f"epoch:{w}"] = sum(bpl[f"{k}:{w}"] for k in ("sp_prepare", "sp_bank", "sp_finish"))
if all(f"{k}:counting" in bpl for k in ("count_transitions_lds_kernel", "count_reduce_slabs")):
    bpl["pass:counting"] = bpl["count_transitions_lds_kernel:counting"] + bpl["count_reduce_slabs:counting"]
co = ("co_bucket_kernel", "co_plan_kernel", "co_expand_kernel", "co_count_lds_kernel")
if all(f"{k}:co_counting" in bpl for k in co):
    bpl["pass:co_counting"] = sum(bpl[f"{k}:co_counting"] for k in co)
if "tb_ew" in bpl:   # the time-basis bank (default float64 form of the headline): its three phases as bench.py names them
    bpl["tb:k1"] = bpl.get("k1_pt_loss_gt")
    bpl["tb:k2"] = bpl["tb_ew"]
    if "k2_t_eq_g_u" in bpl and "k3_w_phi" in bpl:
        bpl["tb:k3"] = bpl["k2_t_eq_g_u"] + bpl["k3_w_phi"]
json.dump(out, open(f"{here}/pmc_traffic.json", "w"), indent=1)
for src, dst in ((f"{tag}_tile_timeline_fused.txt", f"{rnd}_bank_tile_timeline_fused.txt"),
                 (f"{tag}_tile_timeline_separate.txt", f"{rnd}_bank_tile_timeline_separate.txt"),
                 (f"{tag}_clock_probe_fused.json", f"{rnd}_bank_clock_probe_fused.json"),
                 (f"{tag}_clock_probe_separate.json", f"{rnd}_bank_clock_probe_separate.json")):
    if os.path.exists(f"{root}/gpurun_out/{src}"):
        shutil.copy(f"{root}/gpurun_out/{src}", f"{here}/{dst}")
st = newest(f"{root}/gpurun_out/{tag}_trace_coevo400_unfused/*/*_kernel_stats.csv")
if st:
    shutil.copy(st[0], f"{here}/{rnd}_{tag}_coevo400_unfused_kernel_stats.csv")
for name in ("bench_default", "bench_driver", "bench_coevo400_unfused"):
    d = f"{root}/gpurun_out/{tag}_{name}.log"
    if os.path.exists(d):
        line = [l for l in open(d) if l.startswith("{")]
        if line:
            open(f"{here}/{rnd}_{tag}_{name}.json", "w").write(line[-1])

# ---- MFMA utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation vs gfx950 peak")
util = {**STAMP, "source": f"rocprofv3 --pmc SQ_* (one pass of 8 counters), gpurun_out/{tag}_pmc_sq_*; durations from the "
                  "--kernel-trace --stats summaries of the same workloads", "simds": 1024, "clock_GHz": 2.4, "kernels": {}}
# keys: kernel : arithmetic : bank form.  Round 5 keyed by kernel : arithmetic only, so the CB_BANK_UNFUSED=1 pass (three launches
# over all 129 buckets: 222 / 278 us) overwrote the headline's time-basis launches of the same kernels (73 / 69 us) -- VERDICT r5.
FORMS = {"f64": "time_basis", "mixed": "time_basis", "f32": "per_bucket_fused", "f64_unfused": "per_bucket_unfused"}
for dt, w in (("f64", "coevo400"), ("mixed", "coevo400_mixed"), ("f32", "coevo400_f32"), ("f64_unfused", "coevo400_unfused")):
    fs = newest(f"{root}/gpurun_out/{tag}_pmc_sq_{dt}/*/*counter_collection.csv")
    st = newest(f"{root}/gpurun_out/{tag}_trace_{w}/*/*_kernel_stats.csv")
    form = FORMS[dt]
    stats_file = f"{rnd}_{tag}_{w}_kernel_stats.csv"
    dt = dt.replace("_unfused", "")   # (CB_BANK_UNFUSED=1: the three launches of the float64 bank over all 129 buckets)
    if not fs or not st:
        continue
    dur = {}
    for r in csv.DictReader(open(st[0])):
        short = kshort(r["Name"])
        if short:
            dur[short] = float(r["AverageNs"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        short = kshort(r["Kernel_Name"])
        if short and short.startswith(("k123_", "k1_", "k2_", "k3_w", "tb_ew")):
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        if k not in dur or "SQ_VALU_MFMA_BUSY_CYCLES" not in m:
            continue
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        util["kernels"][f"{k}:{dt}:{form}"] = {
            "workload": w, "bank_form": form, "durations_from": stats_file,
            "avg_duration_us": dur[k] / 1e3, "mfma_busy_cycles": m["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_instructions": m.get("SQ_INSTS_MFMA"),
            "mfma_util": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * dur[k] * 2.4),
            "wave_cycles_split": {"waiting_waitcnt_or_barrier": m.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stalled": m.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                                  "issuing": m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "lds_issue_stalled": m.get("SQ_WAIT_INST_LDS", 0.0) / wc},
            "mean_resident_waves": 4.0 * m.get("SQ_WAVE_CYCLES", 0.0) / (dur[k] * 2.4)}
json.dump(util, open(f"{here}/mfma_util.json", "w"), indent=1)
print(json.dumps(out["bytes_per_launch"], indent=1))
print(json.dumps({k: round(v["mfma_util"], 3) for k, v in util["kernels"].items()}, indent=1))
