# Round 6: the headline's driver window twice + 200 epochs, eigensolver phase and counters only (no secondaries, no CPU baseline).
# Run on the GPU box from the repo root: bash profiles/tools/r6_quick.sh [label]
cd ${GRAFT_REPO_ROOT:-$PWD}
for a in "20 5" "20 5" "200 5"; do set -- $a
python3 bench.py --steps $1 --warmup $2 --no-secondary --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $1: ms', round(d['ms_per_step'],4), 'eigh', d['phase_ms']['eigh'], 'k1', d['phase_ms']['k1'], 'k2', d['phase_ms']['k2'], 'k3', d['phase_ms']['k3'], 'k4', d['phase_ms']['k4'], d['eigh'])"
done
