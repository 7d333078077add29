"""The HIP library against vectors computed by the REFERENCE'S OWN Python code
(tests/golden/ref_py_*.npz, produced by tests/golden/make_reference_vectors.py -- see
tests/test_reference_vectors.py for the oracle's half)."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import DepthPrompting as DP
    from genpc_amd.optim_registration import diff_obj_pose as POSE
    return dict(torch=torch, DP=DP, POSE=POSE)


def test_mask_loss_kernels_match_reference_code(env, golden):
    """genpc_mask_loss (the alignment loop's own device code for normalisation, luminance soft masks,
    30 MSE + BCE + 10 Dice and their backward) on the reference's vectors: loss 1e-4, gradient 2e-3
    of its largest entry (float32 here and in the reference, different reduction orders)."""
    torch = env["torch"]
    g = golden("ref_py_mask_loss.npz")
    for name in g["cases"]:
        ref, res = torch.from_numpy(g[name + "_ref"]).cuda(), torch.from_numpy(g[name + "_result"]).cuda()
        loss, grad = env["POSE"].mask_loss(res, ref, with_grad=True)
        want, wgrad = float(g[name + "_mask"]), g[name + "_grad"]
        assert abs(float(loss) - want) <= 1e-4 * abs(want), (name, float(loss), want)
        scale = np.abs(wgrad).max()
        err = np.abs(grad.cpu().numpy() - wgrad)
        assert err.max() <= 2e-3 * scale and err.mean() <= 1e-4 * scale, (name, err.max(), err.mean(), scale)
        assert float(env["POSE"].mask_loss(res, ref)) == float(loss)
        hard = env["POSE"].compute_mask_from_rendering(ref).cpu().numpy()
        np.testing.assert_array_equal(hard, g[name + "_hardmask"])


def test_paint_and_raw_depth_match_reference_code(env, golden):
    """DepthPrompting.paintPixels / getRawDepth on the GPU against the reference's CPU result
    (sequential index_put: the highest point index wins a pixel): bit-exact."""
    torch = env["torch"]
    g = golden("ref_py_paint.npz")
    for name in g["cases"]:
        res, point_size, rate = (int(x) for x in g[name + "_params"])
        cfg = SimpleNamespace(device="cuda", fovy=49.1, res=res, padding=0.15, rescale=True, point_size=point_size,
                              mask_pixel_rate=rate, view_num=6, distance=1.6)
        dp = env["DP"].DepthPrompting(cfg)
        pix = torch.from_numpy(g[name + "_pix"]).cuda()
        out = dp.getRawDepth(pix, torch.from_numpy(g[name + "_depth"]).cuda(), "redwood",
                             colors=torch.from_numpy(g[name + "_colors"]).cuda(), res=res, point_size=point_size,
                             mask_pixel_rate=rate)
        for got, key in zip(out, ("_sparse_img", "_sparse_depth", "_hole_mask1", "_hole_mask2")):
            np.testing.assert_array_equal(got.cpu().numpy(), g[name + key], err_msg=name + key)
