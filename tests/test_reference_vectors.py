"""The CPU oracle against vectors computed by the REFERENCE'S OWN Python code.

tests/golden/ref_py_*.npz hold inputs and the outputs the reference's functions produced for them
in the build container (tests/golden/make_reference_vectors.py: a selective import -- the named
definitions are parsed out of /root/reference with ``ast`` and executed unmodified on real torch /
numpy; nothing of the reference's text is stored).  These tests pin the oracle's restatements of those
functions to the reference itself; tests/test_gpu_reference_vectors.py does the same for the HIP library.
"""
import numpy as np
import pytest


def test_mask_loss_matches_reference_code(oracle, golden):
    """compute_loss_function (diff_obj_pose.py:286-336: per-channel statistical normalisation,
    luminance soft masks, 30 MSE + BCE + 10 Dice) and its gradient with respect to the rendered
    image, as torch autograd gives it through the reference's code in float32."""
    g = golden("ref_py_mask_loss.npz")
    for name in g["cases"]:
        ref, res = g[name + "_ref"], g[name + "_result"]
        loss, grad = oracle.mask_loss(res, ref, with_grad=True)
        want, wgrad = float(g[name + "_mask"]), g[name + "_grad"]
        assert float(g[name + "_total"]) == want              # total = mask_loss * 1 with no clouds given (:329-333)
        assert abs(loss - want) <= 1e-4 * abs(want), (name, loss, want)
        scale = np.abs(wgrad).max()
        # float32 autograd through ~10 elementwise ops and three global reductions against an fp64 evaluation
        assert np.abs(grad - wgrad).max() <= 2e-3 * scale, (name, np.abs(grad - wgrad).max(), scale)
        assert np.abs(grad - wgrad).mean() <= 1e-4 * scale, name


def test_mask_loss_colour_matters(oracle, golden):
    """Darkening half of the posed cloud's points changes the reference's loss (luminance mask): the
    vector `darkhalf48` was rendered with 50 % of the points at 8 % brightness."""
    g = golden("ref_py_mask_loss.npz")
    res = g["darkhalf48_result"]
    lum = 0.299 * res[..., 0] + 0.587 * res[..., 1] + 0.114 * res[..., 2]
    occ = res.max(-1) > 0
    assert (lum[occ] < 0.1).mean() > 0.05                      # visibly dark covered pixels exist
    white = np.repeat(res.max(-1, keepdims=True), 3, -1)      # the same coverage drawn bright
    assert abs(oracle.mask_loss(res, g["darkhalf48_ref"]) - oracle.mask_loss(white, g["darkhalf48_ref"])) > 0.05


def test_paint_and_raw_depth_match_reference_code(oracle, golden):
    """DepthPrompting.paintPixels / getRawDepth (:292-391) on CPU tensors, duplicates included
    (sequential index_put: the last writer wins), stamps 1 / 3 / 5 pixels wide, border clipping."""
    g = golden("ref_py_paint.npz")
    for name in g["cases"]:
        res, point_size, rate = (int(x) for x in g[name + "_params"])
        s_img, s_dep, h1, h2 = oracle.get_raw_depth(g[name + "_pix"], g[name + "_depth"], g[name + "_colors"], res,
                                                    point_size, rate)
        np.testing.assert_array_equal(s_img, g[name + "_sparse_img"], err_msg=name)
        np.testing.assert_array_equal(s_dep, g[name + "_sparse_depth"], err_msg=name)
        np.testing.assert_array_equal(h1, g[name + "_hole_mask1"], err_msg=name)
        np.testing.assert_array_equal(h2, g[name + "_hole_mask2"], err_msg=name)


def test_get_uvs_rescale_matches_reference_code(oracle, golden):
    """DepthPrompting.getUvs (:239-271) on given camera-space points: bit-exact uv and depth."""
    g = golden("ref_py_uvs.npz")
    for name in g["cases"]:
        rescale, padding = g[name + "_params"]
        uv, dp = oracle.rescale_uvs(g[name + "_transformed"], bool(rescale), float(padding))
        np.testing.assert_array_equal(uv, g[name + "_uv"], err_msg=name)
        np.testing.assert_array_equal(dp, g[name + "_depth"], err_msg=name)


def test_loss_reductions_match_reference_code(oracle, golden):
    """utils/loss_util.py:25-49, the five reductions, on given distance arrays."""
    g = golden("ref_py_loss_util.npz")
    for name in g["cases"]:
        d1, d2, de = g[name + "_d1"], g[name + "_d2"], g[name + "_demd"]
        for fn, got in (("chamfer_l1", oracle.cd_l1(d1, d2)), ("chamfer_l2", oracle.cd_l2(d1, d2)),
                        ("chamfer_partial_l1", oracle.cd_partial_l1(d1)), ("chamfer_partial_l2", oracle.cd_partial_l2(d1)),
                        ("emd_loss", oracle.emd_loss(de))):
            want = float(g[name + "_" + fn])
            # torch.mean's float32 summation order (vectorised, pairwise) is not restated: 2 ulp of slack per
            # reduction (emd_loss chains two: mean(1).mean())
            assert abs(float(got) - want) <= (4.8e-7 if fn == "emd_loss" else 2.4e-7) * abs(want), (name, fn, float(got), want)


def test_camera_and_frame_helpers_match_reference_code(oracle, golden):
    """fibonacci_sphere, calculate_up_vector (utils/camera_utils.py:86-113), get_rotate_matrix,
    normalize_numpy (utils/dataUtils.py:455-472,561-581), build_transform (diff_obj_pose.py:464-468)."""
    g = golden("ref_py_utils.npz")
    np.testing.assert_array_equal(oracle.fibonacci_sphere(1024, 1.6), g["fib_1024_1p6"])
    np.testing.assert_array_equal(oracle.fibonacci_sphere(7, 2.0), g["fib_7_2"])
    for eye, up in zip(g["up_eyes"], g["up_vectors"]):
        np.testing.assert_allclose(oracle.calculate_up_vector(eye, np.zeros(3)), up, rtol=0, atol=1e-15)
    # the host-side mirrors in the package (numpy; no GPU needed)
    from genpc_amd import DepthPrompting as DP
    from genpc_amd.utils import dataUtils as DU
    np.testing.assert_array_equal(DP.fibonacci_sphere(1024, 1.6), g["fib_1024_1p6"])
    for eye, up in zip(g["up_eyes"], g["up_vectors"]):
        np.testing.assert_allclose(DP.calculate_up_vector(eye, np.zeros(3)), up, rtol=0, atol=1e-15)
    for key in g.files:
        if key.startswith("rot_"):
            _, ax, ang = key.split("_")
            ang = float(ang.replace("p", ".").replace("m", "-"))
            np.testing.assert_array_equal(DU.get_rotate_matrix(ax, ang), g[key])
    for r, key in ((0.5, "norm_out_0p5"), (1.0, "norm_out_1p0")):
        nx, c, s = DU.normalize_numpy(g["norm_in"].copy(), range=r)
        np.testing.assert_array_equal(nx, g[key])
        np.testing.assert_array_equal(c, g["norm_center"])
        assert s == float(g["norm_scale"])
    T = golden("ref_py_mask_loss.npz")["build_transform"]
    R = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]], np.float32)
    want = np.eye(4, dtype=np.float32)
    want[:3, :3] = R * np.float32(0.8)
    want[:3, 3] = [0.1, -0.2, 0.3]
    np.testing.assert_array_equal(T, want)
