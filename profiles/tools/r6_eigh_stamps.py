"""Round 6: where a planned eigensolve's time goes with NO tracer attached.  Input: the stderr of a CB_DEBUG=1 run of a library
built with -DCB_EIGH_STAMPS (every kernel of the solve leaves s_memrealtime at its entry, eigh_planned.hip.h).  Output (JSON on
stdout): per kernel id the mean microseconds from its entry to the next kernel's entry over the last N solves -- i.e. what the
launch COSTS in the dependent chain, launch overhead and kernel boundary included -- split into launches that ran and launches
that returned at once, plus the solve's segments (begin -> each decision -> lge_norms).
python profiles/tools/r6_eigh_stamps.py LOG [N=20]"""
import collections, json, re, sys
NAMES = {3: "lgj_round (band stage)", 4: "lge_gram", 5: "lge_decide", 6: "lge_so", 10: "lge_gemm<P2>", 12: "lge_gemm<T1>", 13: "lge_gemm<RP>",
         14: "lge_gemm<SQ>", 15: "lge_gemm<GR>", 16: "lge_gemm<R4>", 17: "lge_p34", 20: "lge_norms"}
lines = open(sys.argv[1]).read().splitlines()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
solves = []
for i, l in enumerate(lines):
    if "planned eigh" in l and "| us:" in l and i + 1 < len(lines) and "entries" in lines[i + 1]:
        seg = [float(x) for x in l.split("| us:")[1].replace("=", " ").split()]
        ent = [(int(a), float(b)) for a, b in re.findall(r" (\d+):([0-9.]+)", lines[i + 1].split("norms):")[1])]
        solves.append((l.split("|")[0].strip(), seg, ent))
solves = solves[-N:]
ran, empty = collections.defaultdict(list), collections.defaultdict(list)
first = []
for _, seg, ent in solves:
    first.append(ent[0][1])                       # lge_begin + the warm-start product (entry of the first band stage)
    for (kid, _), (_, dt) in zip(ent[:-1], ent[1:]):
        (empty if dt < 3.5 else ran)[kid].append(dt)
tot = [s[1][-1] for s in solves]
out = {"solves": len(solves), "solve_us_mean": round(sum(tot) / len(tot), 1), "solve_us_min_max": [min(tot), max(tot)],
       "begin_plus_warm_start_us": round(sum(first) / len(first), 1),
       "launches_that_ran": {NAMES[k]: {"per_solve": round(len(v) / len(solves), 2), "us_each": round(sum(v) / len(v), 2),
                                        "us_per_solve": round(sum(v) / len(solves), 1)} for k, v in sorted(ran.items())},
       "launches_that_returned_at_once": {NAMES[k]: {"per_solve": round(len(v) / len(solves), 2), "us_each": round(sum(v) / len(v), 2),
                                                     "us_per_solve": round(sum(v) / len(solves), 1)} for k, v in sorted(empty.items())},
       "example": {"record": solves[-1][0][:200], "segments_us": solves[-1][1]}}
print(json.dumps(out, indent=1))
