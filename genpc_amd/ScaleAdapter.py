"""Geometric half of the reference's ``ScaleAdapter`` stage (ScaleAdapter.py), with the
reference's method signatures so that ``main.py`` calls it unchanged:

    sa = ScaleAdapter(cfg)
    sa.scaleAdapter(xyz, flag, rgb=None)      # remove_bg -> colorPoint -> img2shape   (:78-86)
    sa.scaleReg(flag)                         # reg(cfg, flag, 0.5, True, True)         (:74-75)

``colorPoint``'s image->point colour gather (:46-68, a Python loop over N points on the CPU in
the reference) is one HIP gather; ``scaleReg`` runs the alignment loop / ICP / scale search /
fusion of genpc_amd.reg_xyz on the files of ``{cfg.output_path}/{flag}/``.  Background removal
and the image-to-3D generator are stock torch modules outside this library (north_star): pass them
as ``rembg=`` / ``generative=`` callables with the reference's call shapes
(``rembg(in_png, out_png)``, ``generative(cfg, flag, PIL image)``, :44,:72).

Every method also has a tensor form (no files), used by genpc_amd.pipeline and the tests."""
import os

import numpy as np
import torch

from . import _lib

_L = _lib.lib
_p = _lib.ptr


class ScaleAdapter:
    def __init__(self, cfg, rembg=None, generative=None):
        self.cfg = cfg
        self.device = cfg.device
        self.rembg = rembg
        self.generative = generative

    # ------------------------------------------------------------------ generator stages (stock modules)
    def remove_bg(self, flag, img_resource):
        """ScaleAdapter.py:38-44."""
        if self.rembg is None:
            raise RuntimeError("ScaleAdapter.remove_bg: background removal is a stock module outside this library; "
                               "construct ScaleAdapter(cfg, rembg=callable(in_png, out_png))")
        return self.rembg(f"{self.cfg.output_path}/{flag}/img.png", f"{self.cfg.output_path}/{flag}/img_sam.png")

    def img2shape(self, flag):
        """ScaleAdapter.py:70-72."""
        if self.generative is None:
            raise RuntimeError("ScaleAdapter.img2shape: the image-to-3D generator is a stock module outside this "
                               "library; construct ScaleAdapter(cfg, generative=callable(cfg, flag, image))")
        from PIL import Image
        img = Image.open(f"{self.cfg.output_path}/{flag}/img_sam.png")
        return self.generative(self.cfg, flag, img)

    # ------------------------------------------------------------------ colorPoint
    def colorPoint(self, flag, xyz=None, gt=None, rgb=None, img_resource="depth", scale=1024):
        """ScaleAdapter.py:46-68, same arguments: reads ``{output_path}/{flag}/point_uv.npy`` (and
        ``img.png`` when img_resource == 'depth'), colours every point of ``xyz`` with the pixel it
        projects to (uv * 1024, (row, col) swapped, clipped, image flipped top-bottom), and writes
        ``color_point.ply``; with img_resource == 'obj' the cloud's own ``rgb`` is written (:49-51).
        Returns the colours [N,3] (the reference returns None).

        Tensor form: ``colorPoint(point_uv [N,2], img [3,H,W])`` -> colours [N,3] (no files)."""
        if not isinstance(flag, str):
            return self._color_gather(flag, xyz, scale)
        from .utils.dataUtils import save_ply_xyzrgb
        base = f"{self.cfg.output_path}/{flag}"
        if img_resource == "obj":                                                           # :49-51
            save_ply_xyzrgb(xyz.detach().cpu().numpy(), rgb.detach().cpu().numpy(), f"{base}/color_point.ply")
            return rgb
        if img_resource != "depth":
            raise ValueError("img_resource must be 'obj' or 'depth'")
        from PIL import Image
        point_uv = np.load(f"{base}/point_uv.npy")                                            # :48
        img = np.asarray(Image.open(f"{base}/img.png").convert("RGB"), np.float32) / 255.0    # ToTensor (:58)
        dev = torch.device(self.device)
        img_t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(dev)
        if img_t.shape[1] < scale or img_t.shape[2] < scale:
            raise ValueError("colorPoint indexes a %dx%d pixel grid (ScaleAdapter.py:59-62); img.png is %dx%d"
                             % (scale, scale, img_t.shape[1], img_t.shape[2]))
        colors = self._color_gather(torch.as_tensor(point_uv, dtype=torch.float32, device=dev).reshape(-1, 2), img_t, scale)
        save_ply_xyzrgb(xyz.detach().cpu().numpy(), colors.cpu().numpy(), f"{base}/color_point.ply")   # :68
        return colors

    def _color_gather(self, point_uv, img, scale=1024):
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

    # ------------------------------------------------------------------ scaleReg / scaleAdapter
    def scaleReg(self, flag, complete_xyz=None, **kwargs):
        """ScaleAdapter.py:74-75: ``reg(cfg, flag, cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True)``
        on the files of ``{output_path}/{flag}/`` (writes ``{flag}_fused.ply`` with colours).
        Tensor form: ``scaleReg(partial_xyz, complete_xyz)`` (partial = color_point.ply, complete =
        points sampled from the generated mesh).  Returns reg()'s dict."""
        from .reg_xyz import reg, reg_tensors
        if isinstance(flag, str):
            return reg(self.cfg, flag, cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True, **kwargs)
        return reg_tensors(flag, complete_xyz, generative_model=getattr(self.cfg, "generative_model", "trellis"),
                           dataset=getattr(self.cfg, "dataset", "redwood"), cd_inv_weight=0.5, diff_init=True,
                           reg_fine_xyz=True, **kwargs)

    def scaleAdapter(self, xyz, flag, rgb=None):
        """ScaleAdapter.py:78-86."""
        print("Stage 2 : .....")
        img_resource = "obj" if rgb is not None else "depth"
        self.remove_bg(flag, img_resource=img_resource)
        self.colorPoint(flag, xyz, xyz, rgb, img_resource=img_resource)
        self.img2shape(flag)
