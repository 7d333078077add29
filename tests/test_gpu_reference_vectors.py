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


def test_get_uvs_rescale_matches_reference_code_on_the_gpu(env, golden):
    """DepthPrompting.getUvs' own arithmetic (:246-271: per-camera box, centre, max extent, padding) through
    genpc_get_uvs.  The vectors hold camera-space points `transformed` (kaolin's transform is absent and unpinned) and
    the reference's uv for them; a camera [I | 0] with focal 1 and every point at z = -1 makes the library's projection
    the identity on (x, y) -- 1 * x / 1 -- so the box, its fast-division rescale and the padding run on exactly the
    reference's operands: uv bit-exact, one call per camera row."""
    torch = env["torch"]
    g = golden("ref_py_uvs.npz")
    eye = torch.tensor([[1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]], device="cuda")
    for name in g["cases"]:
        rescale, padding = g[name + "_params"]
        cfg = SimpleNamespace(device="cuda", fovy=90.0, res=256, padding=float(padding), rescale=bool(rescale), point_size=1,
                              mask_pixel_rate=3, view_num=6, distance=1.6)
        dp = env["DP"].DepthPrompting(cfg)
        dp.focal = 1.0
        tr = g[name + "_transformed"]
        for c in range(tr.shape[0]):
            pts = tr[c].copy()
            pts[:, 2] = -1.0
            uv, _, back = dp.getUvs(eye, torch.from_numpy(pts).cuda(), rescale=bool(rescale), padding=float(padding))
            np.testing.assert_array_equal(back[0, :, :2].cpu().numpy(), tr[c][:, :2], err_msg=name)      # the projection was the identity
            np.testing.assert_array_equal(uv[0].cpu().numpy(), g[name + "_uv"][c], err_msg="%s camera %d" % (name, c))


def test_loss_reductions_match_reference_code_on_the_gpu(env, golden, monkeypatch):
    """Completionloss' five reductions (utils/loss_util.py:25-49) as the package computes them -- torch reductions on
    GPU tensors -- on the reference's distance arrays: the distance kernels are replaced by stubs that hand the vectors'
    arrays over, everything after them is the shipped code.  2 ulp per float32 reduction (torch's GPU summation order
    is not the CPU's; emd_loss chains two)."""
    torch = env["torch"]
    from genpc_amd.utils.loss_util import Completionloss
    g = golden("ref_py_loss_util.npz")
    for name in g["cases"]:
        d1, d2, de = (torch.from_numpy(g[name + k]).cuda() for k in ("_d1", "_d2", "_demd"))
        dummy = torch.zeros(d1.shape[0], 4, 3, device="cuda")
        for fn in ("chamfer_l1", "chamfer_l2", "chamfer_partial_l1", "chamfer_partial_l2", "emd_loss"):
            cl = Completionloss("emd" if fn == "emd_loss" else ("cd_l2" if fn.endswith("l2") else "cd_l1"))
            cl.chamfer_dist = lambda a, b: (d1, d2, None, None)
            cl.EMD = lambda a, b, eps, iters: (de, None)
            got = float(getattr(cl, fn)(dummy, dummy))
            want = float(g[name + "_" + fn])
            assert abs(got - want) <= (4.8e-7 if fn == "emd_loss" else 2.4e-7) * abs(want), (name, fn, got, want)
