"""The SE(3)+scale alignment loop of the reference's
optim_registration/diff_obj_pose.py on the gfx950 library.

``object_pose_optimization`` keeps the reference's name, hyper-parameters and
return value (4x4 numpy ``[[sR, t],[0,1]]``, :464-468,:496-594) but takes the two
clouds as tensors (the reference loads a GLB and a PLY through trimesh/open3d,
which is I/O outside the hot path) and optimises the reference's objective
``mask_loss + 3 * (partial_l1(pts, partial) + 0.5 partial_l1(partial, pts)) + 1e-3 |RR^T - I|_F``
(:329-333,543-546).  ``mask_loss`` (30 MSE + BCE + 10 Dice on sigmoid soft masks of the
statistically normalised images, :204-217,261-311) is the reference's own torch code
restated; the IMAGES are not the reference's: it draws them with pytorch3d's CUDA-only
Pulsar renderer (absent, unpinned), this build with its own differentiable occupancy
splat, same camera and radii (include/genpc_hip.h).  The whole multi-start loop
runs on the device without host synchronisation (7 launches per Adam step; 4 with
``cd_only=True``).
"""
import torch

from .. import _lib

_L = _lib.lib
_p = _lib.ptr


def pose_transform(vert_pos, center, params):
    """ObjectPoseOptim.forward's point map (:419-423).  params = rot_6d|trans|log_scale."""
    vert_pos = vert_pos.contiguous().float()
    center = center.contiguous().float()
    params = params.contiguous().float()
    _lib.check_tensors((("vert_pos", vert_pos), ("center", center), ("params", params)))
    pts = torch.empty_like(vert_pos)
    rc = _lib.on_device_of(vert_pos, _L.genpc_pose_transform, vert_pos.shape[0], _p(vert_pos), _p(center),
                           _p(params), _p(pts))
    if rc != 1:
        raise RuntimeError("genpc_pose_transform failed: " + _lib.last_error())
    return pts


def pose_cd_loss_grad(vert_pos, center, params, partial, cd_weight=3.0, reg_weight=0.001):
    """-> (loss[3] = total, cd, |RR^T-I|_F ; grad[10]) for the current parameters."""
    from .. import chamfer_3D
    vert_pos = vert_pos.contiguous().float()
    partial = partial.contiguous().float()
    pts = pose_transform(vert_pos, center, params)
    nc, np_ = vert_pos.shape[0], partial.shape[0]
    dev = vert_pos.device
    d1 = torch.empty(1, nc, device=dev)
    d2 = torch.empty(1, np_, device=dev)
    i1 = torch.empty(1, nc, device=dev, dtype=torch.int32)
    i2 = torch.empty(1, np_, device=dev, dtype=torch.int32)
    if chamfer_3D.forward(pts[None], partial[None], d1, d2, i1, i2) != 1:
        raise RuntimeError("chamfer forward failed: " + _lib.last_error())
    loss = torch.empty(3, device=dev)
    grad = torch.empty(10, device=dev)
    rc = _lib.on_device_of(vert_pos, _L.genpc_pose_cd_grad, nc, _p(vert_pos), _p(center.contiguous().float()),
                           _p(params.contiguous().float()), np_, _p(partial), _p(d1), _p(i1), _p(d2), _p(i2),
                           float(cd_weight), float(reg_weight), _p(loss), _p(grad))
    if rc != 1:
        raise RuntimeError("genpc_pose_cd_grad failed: " + _lib.last_error())
    return loss, grad


def splat_image(points, radius, render_size=224):
    """The library's occupancy splat of a cloud [N,3] -> [render_size, render_size] in [0,1]
    (what stands in for render_reference_image, diff_obj_pose.py:108-134)."""
    pts = points.contiguous().float()
    _lib.check_tensors((("points", pts),))
    img = torch.empty(render_size, render_size, device=pts.device)
    rc = _lib.on_device_of(pts, _L.genpc_splat_image, pts.shape[0], _p(pts), float(radius), int(render_size), _p(img))
    if rc != 1:
        raise RuntimeError("genpc_splat_image failed (rc=%d): %s" % (rc, _lib.last_error()))
    return img


def pose_loss_grad(vert_pos, center, params, partial, radius, render_size=224, cd_weight=3.0, reg_weight=0.001,
                   mask_weight=1.0):
    """compute_loss_function + rot_reg for the current parameters
    -> (loss[4] = total, cd, |RR^T-I|_F, mask_loss ; grad[10])."""
    from .. import chamfer_3D
    vert_pos = vert_pos.contiguous().float()
    partial = partial.contiguous().float()
    pts = pose_transform(vert_pos, center, params)
    nc, np_ = vert_pos.shape[0], partial.shape[0]
    dev = vert_pos.device
    d1 = torch.empty(1, nc, device=dev)
    d2 = torch.empty(1, np_, device=dev)
    i1 = torch.empty(1, nc, device=dev, dtype=torch.int32)
    i2 = torch.empty(1, np_, device=dev, dtype=torch.int32)
    if chamfer_3D.forward(pts[None], partial[None], d1, d2, i1, i2) != 1:
        raise RuntimeError("chamfer forward failed: " + _lib.last_error())
    loss = torch.empty(4, device=dev)
    grad = torch.empty(10, device=dev)
    rc = _lib.on_device_of(vert_pos, _L.genpc_pose_loss_grad, nc, _p(vert_pos), _p(center.contiguous().float()),
                           _p(params.contiguous().float()), np_, _p(partial), _p(d1), _p(i1), _p(d2), _p(i2),
                           float(cd_weight), float(reg_weight), float(mask_weight), float(radius), int(render_size),
                           _p(loss), _p(grad))
    if rc != 1:
        raise RuntimeError("genpc_pose_loss_grad failed (rc=%d): %s" % (rc, _lib.last_error()))
    return loss, grad


def object_pose_optimization(complete_xyz, partial_xyz, radius=0.005, lr=0.005, iters=300, render_size=224,
                             vis=False, save_path=None, device=None, cam_bias_num=4, return_history=False,
                             cd_only=False):
    """diff_obj_pose.py:496-594.  complete_xyz [Nc,3], partial_xyz [Np,3] GPU tensors -- or
    [B,Nc,3] / [B,Np,3]: B scans optimised in lock-step, one batched NN launch per Adam step
    (returns [B,4,4]).  radius / render_size parameterise the silhouette term (splat radius in
    world units, image side); cd_only=True drops that term (Chamfer + orthogonality only).
    vis / save_path (the reference's debug GIF) are accepted and unused."""
    batched = complete_xyz.dim() == 3
    complete_xyz = (complete_xyz if batched else complete_xyz[None]).contiguous().float()
    partial_xyz = (partial_xyz if batched else partial_xyz[None]).contiguous().float()
    _lib.check_tensors((("complete_xyz", complete_xyz), ("partial_xyz", partial_xyz)))
    if complete_xyz.shape[0] != partial_xyz.shape[0]:
        raise ValueError("object_pose_optimization: batch sizes differ")
    dev = complete_xyz.device
    b = complete_xyz.shape[0]
    T = torch.empty(b, 16, device=dev)
    hist = torch.empty(b, cam_bias_num * (iters + 1), device=dev)
    bp = torch.empty(b, 10, device=dev)
    rc = _lib.on_device_of(complete_xyz, _L.genpc_pose_optimize_batch, b, complete_xyz.shape[1], _p(complete_xyz),
                           partial_xyz.shape[1], _p(partial_xyz), float(lr), int(iters), int(cam_bias_num),
                           float(radius), int(render_size), 0.0 if cd_only else 1.0, _p(T), _p(hist), _p(bp))
    if rc != 1:
        raise RuntimeError("genpc_pose_optimize_batch failed (rc=%d): %s" % (rc, _lib.last_error()))
    final_transform = T.reshape(b, 4, 4).cpu().numpy()
    h = hist.reshape(b, cam_bias_num, iters + 1).cpu().numpy()
    if not batched:
        final_transform, h, bpn = final_transform[0], h[0], bp[0].cpu().numpy()
    else:
        bpn = bp.cpu().numpy()
    if return_history:
        return final_transform, h, bpn
    return final_transform
