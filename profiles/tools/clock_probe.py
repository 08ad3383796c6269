"""The clock the chip holds INSIDE the K loops of K1..K3 of the co-evolution epoch (guide: 'DVFS give-back' item 6).

Diagnostic build only:

    CB_EXTRA_HIPCC_FLAGS=-DCB_CLOCK_STAMP python profiles/tools/clock_probe.py [epochs [workload [shard_of]]] > gpurun_out/clock_probe.json

(workload: coevo400 (default) or coevo400_demo; shard_of N: only rank 0's buckets of an N-way deal, as `bench.py --shard-of N`)

(the stamps: csrc/large_bank.hip.h, CB_STAMP_BEGIN / CB_STAMP_END; the shipped library has none).  Runs the bench's
co-evolution workload (400 x 400, B = 129) for `epochs` epochs on the device-driven loop and reads the LAST epoch's stamps:
per workgroup (delta s_memtime, delta s_memrealtime) around the tile's K loop; clock = cycles / ticks * 100 MHz.  Also prints
the K-loop's MFMA issue density per workgroup: 64 cycles per v_mfma_f64_16x16x4 x MFMAs per wave / stamped cycles.
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    assert "-DCB_CLOCK_STAMP" in os.environ.get("CB_EXTRA_HIPCC_FLAGS", ""), "diagnostic build: set CB_EXTRA_HIPCC_FLAGS"
    import torch
    import bench
    import cherryml_amd
    from cherryml_amd import _lib
    from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_arrays

    rng = np.random.default_rng(0)
    wl = bench.make_workload(sys.argv[2] if len(sys.argv) > 2 else "coevo400", 0, rng)
    S = 400
    shard_of = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if shard_of:
        live = np.flatnonzero(np.any(wl["C"].reshape(wl["C"].shape[0], -1) != 0.0, axis=1))
        mine = live[np.arange(0, live.size, shard_of)]
        wl["t"], wl["C"] = wl["t"][mine], wl["C"][mine]
    bank = cherryml_amd.CherryBank(wl["t"], wl["C"], device=0, dtype="f64")
    nB = int(bank.live_buckets[0])
    init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
    mod = cherryml_amd.RateMatrix(num_states=S, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                                  pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True, initialization=init)
    u0 = mod.upper_diag.detach().numpy().copy()
    p0 = mod._pi.detach().numpy().copy()
    out = {"epochs": epochs, "workload": wl["desc"], "live_buckets": nB, "shard_of": shard_of, "kernels": {}}
    lib = _lib.load()
    # the fused bank launch (k123_bank, the default) lives in a translation unit of its own with its own stamp buffer
    # (large_eval picks the form by the bank's shape; the probe pins it: CB_BANK_UNFUSED=1 or, else, CB_BANK_FUSED=1)
    os.environ["CB_TEST_HOOKS"] = "1"
    if not os.environ.get("CB_BANK_UNFUSED"):
        os.environ["CB_BANK_FUSED"] = "1"
    fn = lib.cb_debug_clock_stamps if os.environ.get("CB_BANK_UNFUSED") else lib.cb_debug_clock_stamps_fused
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    # two windows: the driver's (epochs 5..24: the eigensolver's low-power stretches are long) and a long run
    for label, E in (("driver_window_25_epochs", 25), (f"{epochs}_epochs", epochs)):
        bank.profile(True)
        bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=E, lr=0.1)
        tm = bank.timing_means()
        bank.profile(False)
        buf = np.zeros((3, 4096, 6), dtype=np.uint64)
        assert fn(buf.ctypes.data) == 0
        np.save(os.path.join(ROOT, "gpurun_out", f"clock_stamps_{label}.npy"), buf)
        res = {}
        # MFMAs per wave in one tile's K loop: 25 K-steps x 4 sub-steps x (5 + 1 + 1/4) = 625
        for kid, (name, nwg) in enumerate((("k1_pt_loss_gt", nB * 15), ("k2_t_eq_g_u", nB * 25), ("k3_w_phi", nB * 15))):
            n = min(nwg, 4096)
            cyc = buf[kid, :n, 0].astype(np.float64)
            tick = buf[kid, :n, 1].astype(np.float64)
            ok = tick > 0
            if not ok.any():   # (no tile of this stage ran: K3 is gone when the buckets are summed before the last product)
                res[name] = {"workgroups_stamped": 0}
                continue
            clk = cyc[ok] / tick[ok] * 0.1   # GHz
            res[name] = {
                "workgroups_stamped": int(ok.sum()),
                "clock_GHz_median": float(np.median(clk)),
                "clock_GHz_p10": float(np.percentile(clk, 10)),
                "clock_GHz_p90": float(np.percentile(clk, 90)),
                "kloop_cycles_median": float(np.median(cyc[ok])),
                "kloop_us_median": float(np.median(tick[ok]) * 0.01),
                "kloop_mfma_density_median": float(np.median(625.0 * 64.0 / cyc[ok])),
                "phase_ms": tm.get({0: "k1", 1: "k2", 2: "k3"}[kid]),
            }
        out["kernels"][label] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
