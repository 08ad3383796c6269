"""`python bench.py --gpus N` started bare must start N rank processes itself (the reference starts
its own ranks from inside the stage: counting/_count_transitions.py:295-316).  The launcher is driven here with a stub
child (this container has no GPU; a real N > 1 run is the driver's): per-rank environment, relay of rank 0's JSON line,
non-zero exit when any rank fails (and the surviving ranks are taken down), refusal of a --gpus / WORLD_SIZE mismatch."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STUB = r'''
import json, os, sys, time
rank = int(os.environ["RANK"])
with open(os.path.join(os.environ["STUB_OUT"], f"env{rank}.json"), "w") as f:
    json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                               "HSA_ENABLE_IPC_MODE_LEGACY", "CB_BENCH_LAUNCHER")} | {"argv": sys.argv[1:]}, f)
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    sys.exit(7)
if os.environ.get("STUB_HANG_RANK") == str(rank):
    time.sleep(600)
if rank == 0:
    print("RCCL banner line")
    print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}))
else:
    print("noise from rank", rank)   # must not reach the launcher's stdout
'''


def _run_launcher(tmp_path, n, extra_env=None, grace=1.0):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    out = tmp_path / "out"
    out.mkdir(exist_ok=True)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.launch_ranks({n}, ['--gpus', '{n}', '--steps', '3'], child=[sys.executable, {str(stub)!r}], grace={grace}))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(STUB_OUT=str(out), **(extra_env or {}))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    return r, out, time.time() - t0


def test_launcher_starts_n_ranks_and_relays_rank0_line(tmp_path):
    r, out, _ = _run_launcher(tmp_path, 4)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert json.loads(lines[-1]) == {"metric": "stub", "n_gpus": 4}     # the JSON line is the LAST line of stdout
    assert "noise from rank" not in r.stdout and "noise from rank" in r.stderr
    envs = [json.load(open(out / f"env{k}.json")) for k in range(4)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert {e["WORLD_SIZE"] for e in envs} == {"4"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and envs[0]["MASTER_PORT"].isdigit()
    assert {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}
    assert all(e["argv"] == ["--gpus", "4", "--steps", "3"] for e in envs)


def test_launcher_exits_nonzero_when_a_rank_fails_and_stops_the_others(tmp_path):
    # rank 2 fails at once, rank 1 would wait for it for ever (as in a collective): the launcher ends the job
    r, out, took = _run_launcher(tmp_path, 3, {"STUB_FAIL_RANK": "2", "STUB_HANG_RANK": "1"}, grace=1.0)
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert took < 60
    assert "rank 2 exited with code 7" in r.stderr and "rank 1 killed" in r.stderr
    assert r.stdout.strip() == ""           # no result line from a failed job


def test_bench_refuses_gpus_world_size_mismatch():
    """Started under a launcher whose WORLD_SIZE differs from --gpus, bench.py is a hard error (exit 2) before any GPU
    or process-group work -- it used to print n_gpus = WORLD_SIZE with at most a warning."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 4" in r.stderr and "2 rank(s)" in r.stderr, (r.returncode, r.stderr)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr)
