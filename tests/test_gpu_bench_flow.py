"""The N > 1 control flow of bench.py on ONE GPU: `CB_BENCH_BACKEND=gloo` (a test hook; bench.py marks such a line in its
`data` field) lets two ranks share the device, which RCCL refuses.  Everything but the RCCL transport runs for real: the
launcher starts two rank processes, the bank is sharded over them (family counts -> collective over the non-empty buckets ->
per-epoch all-reduce of (loss, dL/dA)), the raw RCCL communicator cannot be made on a duplicated GPU and EVERY rank falls back
to torch's collective together (the failure protocol of DESIGN section 5), the weak-scaling secondaries run with their
collectives on device tensors, and rank 0's JSON line is relayed.  A real N > 1 run over RCCL / xGMI is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_the_gpu_through_the_gloo_hook():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CB_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "TEST HOOK" in d["data"]
    assert "buckets x2" in d["config"]["sharding"] and d["amdahl"]["measured_on_ranks"] == 2
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]
    # two ranks on one device: no raw RCCL communicator -> both ranks chose torch's collective together
    assert "torch glue" in r.stderr and "Adam in torch" in d["config"]["epoch"]
    sr, cc = d["secondary_siterm"], d["secondary_co_counting"]
    assert sr["n_gpus"] == 2 and sr["scaling"] == "weak" and "sites x2" in sr["config"]["sharding"]
    assert cc["n_gpus"] == 2 and cc["scaling"] == "weak" and cc["phase_ms"]["allreduce"] > 0
    # weak scaling: both ranks' events are in the counted total (10,000 families x 64 cherries x ~65 contacts each)
    assert cc["value"] * cc["ms_per_step"] * 1e-3 > 6.0e7
    assert "secondary" not in d      # the LG replica line belongs to N = 1
