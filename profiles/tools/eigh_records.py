"""Sweeps per planned eigensolve of the bench optimisation (CB_DEBUG prints one record per solve): how often does a solve need
more sweeps than the one before (= use the plan's spare slot)?   python profiles/tools/eigh_records.py [epochs]"""
import os, subprocess, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = int(sys.argv[1]) if len(sys.argv) > 1 else 60
code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import numpy as np, torch, bench, cherryml_amd
from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_arrays
wl = bench.make_workload('coevo400', 0, np.random.default_rng(0))
bank = cherryml_amd.CherryBank(wl['t'], wl['C'])
init = jtt_ipw_from_arrays(wl['t'], wl['C'], wl['mask'])
mod = cherryml_amd.RateMatrix(num_states=400, mode='pande_reversible', mask=torch.tensor(wl['mask']), pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
bank.train_pande_reversible(mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy(), mask=wl['mask'], num_epochs={E}, lr=0.1)
print(bank.eigh_counters())
"""
env = dict(os.environ, CB_DEBUG="1")
p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
n = []
for line in p.stderr.splitlines():
    m = re.match(r"\[cherrybank\] planned eigh (\d+):( STALL)?(.*)", line)
    if m:
        sweeps = re.findall(r"([ML])(\d+)(d?)/(\d+) c=([0-9.e+-]+)", m.group(3))
        n.append((int(m.group(1)), bool(m.group(2)), len(sweeps), [s[0] + s[1] + s[2] for s in sweeps], [float(s[4]) for s in sweeps]))
print(p.stdout.strip()[-200:])
prev = None
more = 0
for seq, stall, k, kinds, cs in n:
    flag = ""
    if prev is not None and k > prev:
        more += 1
        flag = "  <-- one more than the solve before"
    print(f"solve {seq:3d}{' STALL' if stall else '      '} sweeps {k}: " + " ".join(f"{a}@{c:.0e}" for a, c in zip(kinds, cs)) + flag)
    if not stall:
        prev = k
print(f"{len(n)} records, {more} solves needed more sweeps than their predecessor")
