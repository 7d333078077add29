"""Geometric half of the reference's ``ScaleAdapter`` stage (ScaleAdapter.py):
``colorPoint``'s image->point colour gather (:46-68, a Python loop over N points on
the CPU in the reference) as one HIP gather, and ``scaleReg`` -> the alignment loop
of genpc_amd.optim_registration.  Background removal and the image-to-3D generator
stay with the reference's stock torch modules (out of scope)."""
import torch

from . import _lib

_L = _lib.lib
_p = _lib.ptr


class ScaleAdapter:
    def __init__(self, cfg):
        self.cfg = cfg
        self.device = cfg.device

    def colorPoint(self, point_uv, img, scale=1024):
        """ScaleAdapter.py:57-66 on tensors: point_uv [N,2] (what DepthPrompting saved as
        point_uv.npy), img [3,H,W] float (ToTensor of img.png, NOT yet flipped -- the
        flip of :57 is folded into the gather).  Returns colours [N,3]."""
        point_uv = point_uv.contiguous().float()
        img = img.contiguous().float()
        _lib.check_tensors((("point_uv", point_uv), ("img", img)))
        ch, h, w = img.shape
        if h < scale or w < scale:
            raise ValueError("colorPoint indexes a %dx%d pixel grid (ScaleAdapter.py:59-62); image is %dx%d"
                             % (scale, scale, h, w))
        n = point_uv.shape[0]
        pix = torch.empty(n, 2, device=img.device, dtype=torch.int32)
        out = torch.empty(n, ch, device=img.device)
        rc = _lib.on_device_of(img, _L.genpc_uv_to_pixels, n, _p(point_uv), float(scale), int(scale) - 1, _p(pix))
        if rc == 1:
            rc = _lib.on_device_of(img, _L.genpc_gather_colors, n, _p(pix), _p(img), ch, h, w, _p(out))
        if rc != 1:
            raise RuntimeError("colorPoint failed: " + _lib.last_error())
        return out

    def scaleReg(self, partial_xyz, complete_xyz):
        """ScaleAdapter.py:74-75: reg(cfg, flag, cd_inv_weight=0.5, diff_init=True,
        reg_fine_xyz=True) on tensors (partial = color_point.ply, complete = points
        sampled from the generated mesh).  Returns reg()'s dict (aligned clouds and
        transforms); the fusion tail (de-duplication, FPS, outlier removal, PLY output)
        is the "next" row of SURVEY 8f."""
        from .reg_xyz import reg
        return reg(partial_xyz, complete_xyz, generative_model=getattr(self.cfg, "generative_model", "trellis"),
                   dataset=getattr(self.cfg, "dataset", "redwood"), cd_inv_weight=0.5, diff_init=True,
                   reg_fine_xyz=True)
