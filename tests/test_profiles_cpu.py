"""The tracked evidence files must describe the kernels they name ("weak 4": profiles/mfma_util.json keyed the
headline's time-basis launches and the 129-bucket per-bucket launches of the SAME kernels under one name, and the per-bucket
figures won).  Checked here, without a GPU: every entry of mfma_util.json names the kernel-stats summary its duration came from,
that summary holds the kernel at that duration (10 %), and the headline form's entries are the time-basis launches."""
import csv
import json
import os

PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def _stats(path):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            out[r["Name"]] = float(r["AverageNs"])
    return out


def _match(kernel_key, stats):
    """average duration of the kernel family + element type the key names (k2_t_eq_g_u_f32 -> k2_t_eq_g_u<float ...>)"""
    fam = kernel_key.replace("_f32", "").replace("_mixed", "")
    hits = []
    for name, ns in stats.items():
        if fam not in name:
            continue
        is_f32 = "<float" in name
        is_mixed = "<double, float" in name
        if kernel_key.endswith("_f32") != is_f32 or kernel_key.endswith("_mixed") != is_mixed:
            continue
        hits.append(ns)
    return hits


def test_mfma_util_entries_are_the_launches_they_name():
    with open(os.path.join(PROFILES, "mfma_util.json")) as f:
        d = json.load(f)
    ks = d["kernels"]
    assert ks, "mfma_util.json holds no kernels"
    for key, v in ks.items():
        kernel, dtype, form = key.split(":")          # kernel : arithmetic : bank form
        assert form == v["bank_form"] and form in ("time_basis", "per_bucket_fused", "per_bucket_unfused"), key
        path = os.path.join(PROFILES, v["durations_from"])
        assert os.path.exists(path), f"{key}: {v['durations_from']} is not tracked under profiles/"
        hits = _match(kernel, _stats(path))
        assert len(hits) == 1, (key, hits)
        assert abs(hits[0] * 1e-3 - v["avg_duration_us"]) <= 0.1 * v["avg_duration_us"], (key, hits[0] * 1e-3, v["avg_duration_us"])
        # utilisation = busy cycles / (SIMDs x duration x clock), recomputed
        want = v["mfma_busy_cycles"] / (d["simds"] * v["avg_duration_us"] * 1e3 * d["clock_GHz"])
        assert abs(want - v["mfma_util"]) < 1e-9, key
    # the headline (float64) runs in the time basis: its product launches are the short ones (~30-47 virtual buckets, < 0.12 ms),
    # not the 129-bucket launches of the same kernels (0.22 / 0.28 ms), which live under per_bucket_unfused
    for k in ("k1_pt_loss_gt", "k2_t_eq_g_u", "k3_w_phi", "tb_ew"):
        assert f"{k}:f64:time_basis" in ks, k
        assert ks[f"{k}:f64:time_basis"]["avg_duration_us"] < 120.0, k
    for k in ("k1_pt_loss_gt", "k2_t_eq_g_u"):
        if f"{k}:f64:per_bucket_unfused" in ks:
            assert ks[f"{k}:f64:per_bucket_unfused"]["avg_duration_us"] > 150.0, k
