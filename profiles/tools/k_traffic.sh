R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/r2_k1pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
v={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{R}/gpurun_out/r2_k1pmc_{c}/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in ("k1_pt","k2_t","k3_w"):
            if k in r["Kernel_Name"]: agg[k].append(float(r["Counter_Value"]))
    for k,x in agg.items(): v.setdefault(k,{})[c]=sum(x)/len(x)
for k,d in v.items(): print(k, "bytes", (2*d["FETCH_SIZE"]+d["WRITE_SIZE"])*1024/1e6, "MB", d)
PY
