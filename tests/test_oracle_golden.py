"""Pins the CPU oracle (oracle/genpc_oracle.c).

The reference has no tests and no golden vectors.  The pins
available are the values measured at survey time by executing the reference's
kernel bodies on the CPU (arithmetic without
contraction = oracle mode 0), and the committed fixtures generated from the
oracle (regression pins for both arithmetic modes).
"""
import numpy as np
import pytest

from conftest import gen_pair


def test_survey_chamfer_seed1(oracle):
    a, b = gen_pair(1, (1, 2048, 3), (1, 2048, 3))
    d1, d2, _, _ = oracle.chamfer_forward(a, b, 0)
    assert oracle.cd_l1(d1, d2) == np.float32(0.04512813)
    assert oracle.cd_l2(d1, d2) == np.float32(0.0046723103)


def test_survey_chamfer_seed0_ragged(oracle):
    a, b = gen_pair(0, (2, 1000, 3), (2, 777, 3))
    d1, d2, _, _ = oracle.chamfer_forward(a, b, 0)
    assert oracle.cd_l1(d1, d2) == np.float32(0.06116741)
    assert abs(float(oracle.cd_l2(d1, d2)) - 0.00856258) < 5e-9


def test_survey_emd_seed0(oracle):
    a, b = gen_pair(0, (2, 1024, 3), (2, 1024, 3), 0.0)
    d, ass = oracle.emd_forward(a, b, 0.005, 50, 0)
    per = np.sqrt(d).mean(axis=1, dtype=np.float32)
    assert per[0] == np.float32(0.07579704) and per[1] == np.float32(0.07364403)
    assert [len(np.unique(x)) for x in ass] == [979, 975]


def test_survey_emd_seed1(oracle):
    a, b = gen_pair(1, (1, 2048, 3), (1, 2048, 3), 0.0)
    d, _ = oracle.emd_forward(a, b, 0.005, 50, 0)
    assert abs(float(oracle.emd_loss(d)) - 0.059634) < 5e-7


def test_survey_scan01184(oracle, golden):
    g = golden("scan01184_fps2048.npz")
    P, G = g["partial"], g["gt"]
    d1, d2, _, _ = oracle.chamfer_forward(P, G, 0)
    assert oracle.cd_l1(d1, d2) == np.float32(0.040335327)
    assert oracle.cd_l2(d1, d2) == np.float32(0.0099346815)
    d, _ = oracle.emd_forward(P, G, 0.005, 50, 0)
    assert oracle.emd_loss(d) == np.float32(0.065699235)


@pytest.mark.parametrize("name", ["chamfer_seed1_b1_2048.npz", "chamfer_seed0_b2_1000x777.npz",
                                  "chamfer_seed7_b3_5x3.npz", "chamfer_seed11_dups.npz"])
@pytest.mark.parametrize("mode", [0, 1])
def test_chamfer_fixture_regression(oracle, golden, name, mode):
    g = golden(name)
    d1, d2, i1, i2 = oracle.chamfer_forward(g["xyz1"], g["xyz2"], mode)
    np.testing.assert_array_equal(d1, g[f"dist1_m{mode}"])
    np.testing.assert_array_equal(d2, g[f"dist2_m{mode}"])
    np.testing.assert_array_equal(i1, g[f"idx1_m{mode}"])
    np.testing.assert_array_equal(i2, g[f"idx2_m{mode}"])


@pytest.mark.parametrize("name", ["emd_seed0_b2_1024.npz", "emd_seed5_dups.npz", "emd_seed9_2304_dups.npz"])
@pytest.mark.parametrize("mode", [0, 1])
def test_emd_fixture_regression(oracle, golden, name, mode):
    g = golden(name)
    d, ass = oracle.emd_forward(g["xyz1"], g["xyz2"], float(g["eps"]), int(g["iters"]), mode)
    np.testing.assert_array_equal(ass, g[f"assignment_m{mode}"])
    np.testing.assert_array_equal(d, g[f"dist_m{mode}"])


def test_emd_input_checks(oracle):
    a = np.zeros((1, 256, 3), np.float32)
    with pytest.raises(ValueError):
        oracle.emd_forward(a, np.zeros((1, 512, 3), np.float32), 0.005, 2)      # n != m
    with pytest.raises(ValueError):
        oracle.emd_forward(np.zeros((1, 100, 3), np.float32), np.zeros((1, 100, 3), np.float32), 0.005, 2)
    with pytest.raises(ValueError):
        oracle.emd_forward(np.zeros((513, 256, 3), np.float32), np.zeros((513, 256, 3), np.float32), 0.005, 1)


def test_config3_fixture_is_consistent(oracle, golden):
    """The 13-scan fixture: shapes, the survey's full-resolution CD-L1 ordering
    (06830 mis-framed, 06145 / 09868 the two best) and one scan re-derived."""
    g = golden("scans13_fps16384.npz")
    assert g["partial"].shape == (13, 16384, 3) and g["gt"].shape == (13, 16384, 3)
    ids = list(g["ids"])
    cd = g["cd_l1_m0"]
    assert cd[ids.index("06830")] > 2.5
    assert set(np.argsort(cd)[:2]) == {ids.index("06145"), ids.index("09868")}
    assert np.abs(g["cd_l1_m0"] - g["cd_l1_m1"]).max() < 1e-7
    i = ids.index("06145")
    d1, d2, i1, i2 = oracle.chamfer_forward(g["partial"][i:i + 1], g["gt"][i:i + 1], 1)
    assert oracle.cd_l1(d1, d2) == g["cd_l1_m1"][i]
    assert int(i1.astype(np.int64).sum()) == int(g["idx1_sum_m1"][i])
