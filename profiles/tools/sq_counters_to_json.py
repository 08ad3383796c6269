"""rocprofv3 --pmc csv output (one directory per pass) -> {kernel: {counter: mean value per launch}} as JSON.
python sq_counters_to_json.py OUT.json "note" DIR [DIR ...]   (kernels whose name contains one of KEEP only)"""
import collections, csv, glob, json, os, sys
KEEP = ("sp_bank", "sp_prepare", "sp_finish", "sp_step")
out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)   # (dispatch, kernel, counter) summed over the XCD / SE rows
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(k in name for k in KEEP):
                continue
            per[(r["Dispatch_Id"], name, r["Counter_Name"])] += float(r["Counter_Value"])
        for (disp, name, ctr), v in per.items():
            acc[name][ctr].append(v)
res = {"source": note, "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}}
for k, cs in res["kernels"].items():
    if cs.get("SQ_INSTS_LDS"):
        cs["conflict_cycles_per_lds_instruction"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_INSTS_LDS"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v.get("conflict_cycles_per_lds_instruction") for k, v in res["kernels"].items()}))
