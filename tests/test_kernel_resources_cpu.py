"""Register budgets of the kernels whose speed rests on an occupancy: read from the code objects the build
produced (no GPU needed), so that a compiler release that pushes one of them over its budget fails HERE and not as a silent
halving of the occupancy on the GPU box.

The bank tiles (`k123_bank`, `k1_pt_loss_gt`, `k2_t_eq_g_u`, `k3_w_phi`) keep sixteen waves per CU -- four per SIMD -- and that
needs <= 128 VGPRs (512-entry file per SIMD lane, allocation granule 8: MI355X_MICROARCH.md, "Register files"); the fused
kernel lives in a translation unit of its own with `-mllvm -disable-machine-licm` precisely to stay there (csrc/cb_bank_fused.hip)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))


@pytest.fixture(scope="module")
def meta():
    from cherryml_amd import _build
    from kernel_meta import kernel_meta
    _build.build()   # (a no-op when the library is newer than its sources)
    return kernel_meta()


def _find(meta, prefix):
    hits = {k: v for k, v in meta.items() if k.startswith(prefix)}
    assert hits, f"no kernel named {prefix}* in the built objects: {sorted(meta)[:5]} ..."
    return hits


def test_fused_bank_kernel_stays_at_four_waves_per_simd(meta):
    for name, c in _find(meta, "void k123_bank<").items():
        assert c["vgpr"] <= 128, (name, c)
        assert c["vgpr_spill"] <= 8, (name, c)          # a handful of spilled registers outside the K loop is what it has today
        kg = 2 if ", 2>" in name else 1
        assert c["lds"] <= 40960 * kg, (name, c)        # 16 waves per CU: four workgroups of 40 KB or two of 80 KB
    # both tile forms of the float64 bank exist
    assert any("k123_bank<double, double, 1>" in n for n in meta) and any("k123_bank<double, double, 2>" in n for n in meta)


@pytest.mark.parametrize("prefix", ["void k1_pt_loss_gt<", "void k2_t_eq_g_u<", "void k3_w_phi<"])
def test_separate_bank_kernels_do_not_spill(meta, prefix):
    for name, c in _find(meta, prefix).items():
        assert c["vgpr"] <= 128 and c["vgpr_spill"] == 0 and c["scratch"] == 0, (name, c)


def test_site_bank_kernel_keeps_three_workgroups_per_cu(meta):
    # sp_bank<5, symmetric, three workgroups per CU> (SiteRM, cfg 4): 256 threads x 3 workgroups = 3 waves per SIMD -> <= 168 VGPRs
    hits = {k: v for k, v in meta.items() if k.startswith("void sp_bank<5, true, true>")}
    assert hits, "sp_bank<5, true, true> not found"
    for name, c in hits.items():
        assert c["vgpr"] <= 168 and c["vgpr_spill"] <= 12, (name, c)   # (9 today: outside the tile loops)


def test_time_basis_kernels(meta):
    # tb_ew: 16 waves of one workgroup per CU = four per SIMD -> <= 128 VGPRs; nothing spilled inside its loop (a couple of
    # dwords around it is what it has today); the skeleton products are k1_pt_loss_gt's RAW form (checked above with its siblings)
    hits = _find(meta, "void tb_ew<")
    assert len(hits) == 4, sorted(hits)
    for name, c in hits.items():
        # (the forms with 24 forward / 48 gradient skeleton slots -- wider spectral ranges than any bank so far -- hold more
        # accumulators and spill a few of them around the loop)
        # (round 6: 4 / 6 spilled with 32 gradient slots, 19 / 27 with 48 -- tb_ew<16, 48> timed at 400 states:
        # profiles/r06_tb_ew48_timing.json)
        assert c["vgpr"] <= 128 and c["vgpr_spill"] <= (8 if ", 32, 8" in name else 32), (name, c)
    assert any("k1_pt_loss_gt<double, double, false, 1, true>" in n for n in meta)
