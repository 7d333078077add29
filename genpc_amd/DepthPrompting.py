"""Geometric half of the reference's ``DepthPrompting`` stage (DepthPrompting.py)
on the gfx950 library: camera set, ``getUvs`` (:239-271), ``paintPixels``
(:292-339), ``getRawDepth`` (:341-391).  Same method names, argument meaning and
return shapes; the image-generation half (inpainting, depth-conditioned diffusion)
stays with the reference's stock torch modules and is out of scope here.

Cameras are [C,12] tensors of 3x4 world->camera matrices instead of kaolin
``Camera`` objects (kaolin is not a dependency): ``create_cameras`` restates
utils/camera_utils.py:84-160 (fibonacci sphere, ``calculate_up_vector``, look-at).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

_L = _lib.lib
_p = _lib.ptr


def fibonacci_sphere(samples, radius):
    """utils/camera_utils.py:84-101."""
    pts = []
    phi = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(samples):
        y = 1 - (i / float(samples - 1)) * 2
        radius_y = math.sqrt(1 - y * y)
        theta = phi * i
        pts.append((math.cos(theta) * radius_y * radius, y * radius, math.sin(theta) * radius_y * radius))
    return np.array(pts)


def calculate_up_vector(eye_position, target_position):
    """utils/camera_utils.py:104-113."""
    gaze = np.asarray(target_position, np.float64) - np.asarray(eye_position, np.float64)
    world_up = np.array([0.0, 1.0, 0.0])
    side = np.cross(gaze, world_up)
    if np.allclose(side, 0):
        return np.array([0.0, 0.0, 1.0])
    up = np.cross(side, gaze)
    return up / np.linalg.norm(up)


def look_at(eye, at, up):
    """3x4 row-major world->camera matrix of a right-handed camera looking down -Z
    (kaolin Camera.from_args extrinsics), fp32 like the oracle."""
    eye, at, up = (np.asarray(x, np.float32) for x in (eye, at, up))
    back = eye - at
    back = back / np.sqrt((back * back).sum(dtype=np.float32))
    right = np.cross(up, back).astype(np.float32)
    right = right / np.sqrt((right * right).sum(dtype=np.float32))
    upv = np.cross(back, right).astype(np.float32)
    rows = np.stack([right, upv, back]).astype(np.float32)
    t = -(rows @ eye).astype(np.float32)
    return np.concatenate([rows, t[:, None]], axis=1).astype(np.float32).reshape(12)


def create_cameras(num_views=1024, distance=1.6, fovy=49.1, device="cuda"):
    """utils/camera_utils.py:115-160 for the fibonacci distribution -> (views[C,12]
    tensor, eye_positions[C,3] numpy, focal)."""
    eyes = fibonacci_sphere(num_views, distance)
    views = np.stack([look_at(e, np.zeros(3), calculate_up_vector(e, np.zeros(3))) for e in eyes])
    focal = 1.0 / math.tan(math.pi * fovy / 180 / 2)
    return torch.from_numpy(views).to(device), eyes, focal


import threading as _threading
_EYES_LOCK = _threading.Lock()

class DepthPrompting:
    """cfg needs: device, fovy, res, padding, rescale, point_size, mask_pixel_rate
    (configs/config.yaml keys, unchanged)."""

    def __init__(self, cfg, cameras=None, focal=None, inpainter=None, depth2image=None):
        """inpainter / depth2image: the two generator stages of stage 1 (depth inpainting: Flux fill / DDNM / cv2,
        DepthPrompting.py:196-227; depth-conditioned image generation: ControlNet / Flux / Qwen, :42-67,77-82) are
        stock torch modules outside this library (north_star) -- pass them as callables:
        ``inpainter(raw_depth [3,R,R], mask [3,R,R]) -> depth [3,R,R] tensor (or a PIL image)`` and
        ``depth2image(depth PIL image, category, size) -> PIL image`` (the reference's ``.generate``)."""
        self.cfg = cfg
        self.device = torch.device(cfg.device)
        self.inpainter = inpainter
        self.depth2image = depth2image
        if cameras is None:
            cameras, self.viewpoints, focal = create_cameras(
                num_views=cfg.view_num, distance=cfg.distance, fovy=cfg.fovy, device=self.device)
        else:
            # cameras were handed in: eye = -R^T t of each 3x4 world->camera matrix (viewpoint_select, getDepth and
            # hidden_point_removal all index self.viewpoints)
            m = torch.as_tensor(cameras).reshape(-1, 3, 4).double().cpu().numpy()
            self.viewpoints = -np.einsum("cji,cj->ci", m[:, :, :3], m[:, :, 3])
        self.cameras = cameras
        self.focal = focal if focal is not None else 1.0 / math.tan(math.pi * cfg.fovy / 180 / 2)

    # DepthPrompting.py:239-271
    def getUvs(self, cams, points, rescale=True, padding=0.15, want_transformed=True):
        cams = cams.reshape(-1, 12).contiguous().float()
        points = points.contiguous().float()
        _lib.check_tensors((("cams", cams), ("points", points)))
        c, n = cams.shape[0], points.shape[0]
        uv = torch.empty(c, n, 2, device=points.device)
        depth = torch.empty(c, n, device=points.device)
        tr = torch.empty(c, n, 3, device=points.device) if want_transformed else None
        rc = _lib.on_device_of(points, _L.genpc_get_uvs, c, n, _p(cams), float(self.focal), 1e-2, 1e2, _p(points),
                               _p(tr), _p(uv), _p(depth), int(bool(rescale)), float(np.float32(1 - 2 * padding)),
                               _p(None))
        if rc != 1:
            raise RuntimeError("genpc_get_uvs failed: " + _lib.last_error())
        return uv, depth, tr

    # DepthPrompting.py:273-290
    def getVisiblePoints(self, points, viewpoints=None, radius=None):
        """[C,N] bool: point n is visible from viewpoints[c] -- the reference's call
        (points, viewpoints, radius) and result.  The reference loops over the viewpoints calling
        open3d's hidden_point_removal (Katz: spherical flipping + qhull, CPU); here all viewpoints go
        through one exact evaluation of the same operator on the GPU (csrc/hpr.hip: the hull-vertex test
        as a normal-cone polygon per point, double arithmetic).  viewpoints: eye positions [C,3]
        (default: this object's), radius: cfg.removal_radius by default."""
        return self.hidden_point_removal(points, viewpoints, radius)[0]

    def hidden_point_removal(self, points, viewpoints=None, radius=None, best_only=False, want_second=False):
        """-> (visible [C,N] bool, counts [C] int32, points that needed the wave-per-point passes -- 0 unless
        want_second: that count is the one thing of this operator the host has to wait for; without it the call
        enqueues its kernels and returns).  An internal error of the library (none known) turns every count into -1.
        best_only=True (viewpoint_select): the library stops working on views that can no longer see the most
        points; counts are then exact for the views that could, lower bounds below the maximum for the rest
        (argmax unchanged), and `visible` is complete only for the former."""
        if viewpoints is None:
            viewpoints = self.viewpoints
        if radius is None:
            radius = self.cfg.removal_radius
        points = points.contiguous().float()
        _lib.check_tensors((("points", points),))
        if torch.is_tensor(viewpoints) and viewpoints.device == points.device:
            eyes = viewpoints.double().reshape(-1, 3).contiguous()
        else:
            # (an upload from pageable memory waits for the stream: the device copy of a viewpoint set is kept, keyed by content)
            host = np.ascontiguousarray(np.asarray(viewpoints.cpu() if torch.is_tensor(viewpoints) else viewpoints, np.float64).reshape(-1, 3))
            # (keyed by the bytes themselves -- a hash alone could hand a colliding set the wrong viewpoints -- and guarded by a lock:
            #  one DepthPrompting object may be shared by host threads)
            key = (str(points.device), host.shape[0], host.tobytes())
            with _EYES_LOCK:
                cache = self.__dict__.setdefault("_eyes_on_device", {})
                eyes = cache.get(key)
                if eyes is None:
                    if len(cache) >= 8:
                        cache.clear()
                    eyes = cache[key] = torch.from_numpy(host).to(points.device)
        c, n = eyes.shape[0], points.shape[0]
        vis = torch.zeros(c, n, device=points.device, dtype=torch.uint8)
        cnt = torch.empty(c, device=points.device, dtype=torch.int32)
        total_second = 0
        # the library takes at most 4096 viewpoints and 2^31 - 1 (viewpoint, point) pairs per call, and its
        # per-stream workspace holds ~65 bytes per pair plus a 256-byte polygon slot per pair (sized for the worst
        # case since the counts stay on the device, capped by GENPC_HPR_PARK_MB) and is never released: views are
        # chunked so that one call stays within cfg.hpr_workspace_bytes (default 4 GiB; 1024 x 10000 pairs is one call)
        budget = int(getattr(self.cfg, "hpr_workspace_bytes", 4 << 30))
        step = max(1, min(4096, (2 ** 31 - 1) // max(n, 1), budget // (321 * max(n, 1))))
        for v0 in range(0, c, step):
            v1 = min(c, v0 + step)
            second = ctypes.c_int(0)
            sp = ctypes.addressof(second) if want_second else None
            if best_only and v0 == 0 and v1 == c:      # (pruning needs all views in one call)
                rc = _lib.on_device_of(points, _L.genpc_hpr_best_view_counts, c, n, _p(points), _p(eyes), float(radius),
                                       _p(vis), _p(cnt), None, sp)
            else:
                rc = _lib.on_device_of(points, _L.genpc_hpr_visibility, v1 - v0, n, _p(points), _p(eyes[v0:v1]), float(radius),
                                       _p(vis[v0:v1]), _p(cnt[v0:v1]), sp)
            if rc != 1:
                raise RuntimeError("genpc_hpr_visibility failed (rc=%d): %s" % (rc, _lib.last_error()))
            total_second += int(second.value)
        return vis.bool(), cnt, total_second

    def getVisiblePointsZBuffer(self, points, cams=None, tol=1e-4, res=None, uvs=None, depths=None, point_size=2):
        """A cheaper visibility for ranking viewpoints (NOT the reference's operator, which is
        getVisiblePoints above): [C,N] bool and the per-camera counts [C] from a z-buffer test at
        `res` x `res` (cfg.cam_res by default): visible = no point whose (2*point_size-1)^2 stamp covers
        the pixel is nearer by more than `tol` (NDC depth).  `cams` [C,12] view matrices (default: this
        object's).  How differently the two operators rank views is measured in
        tests/test_gpu_scans.py::test_viewpoint_selection_against_katz_hpr."""
        if uvs is None:
            uvs, depths, _ = self.getUvs(self.cameras if cams is None else cams, points,
                                         rescale=self.cfg.rescale, padding=self.cfg.padding, want_transformed=False)
        res = int(res or getattr(self.cfg, "cam_res", 256))
        c, n = depths.shape
        vis = torch.empty(c, n, device=depths.device, dtype=torch.uint8)
        cnt = torch.empty(c, device=depths.device, dtype=torch.int32)
        rc = _lib.on_device_of(depths, _L.genpc_zbuffer_visibility, c, n, _p(uvs.contiguous()), _p(depths.contiguous()),
                               res, int(point_size), float(tol), _p(vis), _p(cnt))
        if rc != 1:
            raise RuntimeError("genpc_zbuffer_visibility failed: " + _lib.last_error())
        return vis.bool(), cnt

    # DepthPrompting.py:87-98
    def viewpoint_select(self, xyz, tol=1e-4, zbuffer=False):
        """FPS to cfg.downsample_num points, hidden-point removal from every viewpoint with
        cfg.removal_radius, the viewpoint that sees the most points (zbuffer=True: rank with the
        z-buffer test instead)."""
        from .fps import fps_sampling
        k = int(getattr(self.cfg, "downsample_num", 10000))
        # (a cloud that is not larger than downsample_num is taken whole: sampling all of its points would only
        # permute them -- fpsample starts from a random point, so the reference's own order is arbitrary -- and the
        # visible COUNTS this method ranks by do not depend on the order)
        xyz_fps = xyz[fps_sampling(xyz.contiguous().float(), k).long()] if xyz.shape[0] > k else xyz.contiguous().float()
        if zbuffer:
            _, counts = self.getVisiblePointsZBuffer(xyz_fps, cams=self.cameras, tol=tol)
        else:
            _, counts, _ = self.hidden_point_removal(xyz_fps, self.viewpoints, getattr(self.cfg, "removal_radius", 10000),
                                                     best_only=True)
        top, best = torch.max(counts, 0)
        top, best = torch.stack([top.long(), best]).tolist()          # (one host read)
        if top < 0:
            raise RuntimeError("genpc_hpr_best_view_counts: internal error (the library marked its counts invalid)")
        return int(best)

    # DepthPrompting.py:69-85
    def getImage(self, xyz, flag, rgb=None, depth_gen=True, img_gen=True):
        """Stage 1 with the reference's signature (main.py:54): the depth prompt of the scan -- getDepth writes
        ``raw_depth.png``, ``mask.png``, ``depth.png`` (through the injected inpainter), ``point_uv.npy``,
        ``viewpoint.npy`` and ``camera.pth`` under ``{cfg.output_path}/{flag}/`` -- then the depth-conditioned image
        (``img.png``, through the injected generator).  rgb None: random colours like the reference's getRandomColor
        (they only tint the sparse image).  Returns getDepth's dict (None with depth_gen=False)."""
        print("Stage 1 : Depth Prompting.....")
        import os
        import time
        from PIL import Image
        start = time.time()
        base = f"{self.cfg.output_path}/{flag}"
        if rgb is None:
            # getRandomColor (utils/dataUtils.py:157-160): SH2RGB(random / 255) = random / 255 * 0.2820948 + 0.5
            rgb = torch.from_numpy(np.random.random((xyz.shape[0], 3)) / 255.0 * 0.28209479177387814 + 0.5).float().to(xyz.device)
        out = self.getDepth(xyz, flag, rgb) if depth_gen else None
        if not os.path.exists(f"{base}/depth.png"):
            raise RuntimeError("DepthPrompting.getImage: %s/depth.png does not exist -- depth inpainting is a stock module "
                               "outside this library; construct DepthPrompting(cfg, inpainter=callable(raw_depth, mask))" % base)
        self.depth = Image.open(f"{base}/depth.png").convert("RGB")                               # load_image (:76)
        if img_gen:
            if self.depth2image is None:
                raise RuntimeError("DepthPrompting.getImage: the depth-conditioned image generator is a stock module outside "
                                   "this library; construct DepthPrompting(cfg, depth2image=callable(depth, category, size))")
            print(" Image Generation.....")
            from .utils.dataUtils import getCategory
            self.image = self.depth2image(self.depth, getCategory(flag), getattr(self.cfg, "generate_res", 512))
            self.image.save(f"{base}/img.png")
        print(f" Take {int(time.time() - start)} seconds")
        return out

    # DepthPrompting.py:100-237
    def getDepth(self, xyz, flag=None, rgb=None):
        """The reference's getDepth: viewpoint selection by hidden-point removal (view_num == 6: view 1), the
        opposite viewpoint's camera, hidden-point removal of the WHOLE cloud from both, the depth-sum heuristic that
        picks one of the two (:154-175), pixels, getRawDepth on the visible points, and -- when `flag` names a
        directory under cfg.output_path -- the stage's files (:196-237): ``raw_depth.png``, ``mask.png``, ``depth.png``
        (if an inpainter was injected), ``point_uv.npy``, ``viewpoint.npy``, ``camera.pth`` (a dict of plain tensors:
        the 3x4 world->camera matrix, eye, focal -- kaolin's Camera object is not a dependency).
        SELECT FIRST, PROJECT SECOND: the reference projects the cloud through all 1024 cameras ([1024,N,3], 0.9 GB
        at N = 71 k) and then consumes two rows; here only the chosen camera and its opposite are projected
         -- getUvs is per camera, so the two rows are the same bits.
        Sets self.point_uv / self.view / self.cam like the reference; returns a dict with sparse_img, raw_depth,
        hole_mask1, hole_mask2, view_index, used_opposite, visible (bool [N]), depth_sums, uv, depth, pixels."""
        xyz = xyz.contiguous().float()
        n = xyz.shape[0]
        rgb = torch.ones(n, 3, device=xyz.device) if rgb is None else rgb
        best = 1 if getattr(self.cfg, "view_num", 0) == 6 else self.viewpoint_select(xyz)
        original = np.asarray(self.viewpoints[best], np.float64)
        opposite = -original
        opp_cam = torch.from_numpy(look_at(opposite, np.zeros(3), calculate_up_vector(opposite, np.zeros(3)))).to(xyz.device)
        two = torch.stack([self.cameras.reshape(-1, 12)[best].to(xyz.device), opp_cam])
        uv2, depth2, _ = self.getUvs(two, xyz, rescale=self.cfg.rescale, padding=self.cfg.padding, want_transformed=False)
        radius = getattr(self.cfg, "removal_radius", 10000)
        vis, _, _ = self.hidden_point_removal(xyz, np.stack([original, opposite]), radius)      # both viewpoints, one call
        sum1 = float(depth2[0][vis[0]].sum())
        sum2 = float(depth2[1][vis[1]].sum())
        if sum1 >= sum2:
            used_opposite, visible, uvs, depths = False, vis[0], uv2[0], depth2[0]
            self.view, self.cam = original, two[0]
        else:
            used_opposite, visible, uvs, depths = True, vis[1], uv2[1], depth2[1]
            self.view, self.cam = opposite, opp_cam
        pix = self.uvToPixels(uvs, self.cfg.res).long()
        sparse_img, raw_depth, hole1, hole2 = self.getRawDepth(
            pix[visible], depths[visible], colors=rgb[visible].contiguous(), dataset=getattr(self.cfg, "dataset", None),
            res=self.cfg.res, point_size=self.cfg.point_size, mask_pixel_rate=self.cfg.mask_pixel_rate)
        self.point_uv = uvs
        out = dict(sparse_img=sparse_img, raw_depth=raw_depth, hole_mask1=hole1, hole_mask2=hole2, view_index=best,
                   used_opposite=used_opposite, visible=visible, depth_sums=(sum1, sum2), uv=uvs, depth=depths, pixels=pix)
        if isinstance(flag, str) and getattr(self.cfg, "output_path", None):
            self._write_stage1_files(flag, out)
        return out

    @staticmethod
    def _save_image(t, path):
        """torchvision.utils.save_image for one [3,H,W] image in [0,1]: mul(255).add(0.5).clamp(0,255) -> uint8 PNG."""
        from PIL import Image
        a = t.detach().float().mul(255).add(0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
        Image.fromarray(a).save(path)

    def _write_stage1_files(self, flag, out):
        """DepthPrompting.py:196-237."""
        import os
        base = f"{self.cfg.output_path}/{flag}"
        os.makedirs(base, exist_ok=True)
        self._save_image(out["raw_depth"], f"{base}/raw_depth.png")
        # the reference's mask choice: hole_mask1 for 'flux' / 'cv2', hole_mask2 for 'DDNM' (:200-227)
        mask = out["hole_mask2"] if str(getattr(self.cfg, "inpainter", "")).upper().startswith("DDNM") else out["hole_mask1"]
        self._save_image(mask, f"{base}/mask.png")
        if self.inpainter is not None:
            print(" Inpainting depth...")
            depth = self.inpainter(out["raw_depth"], mask)
            if torch.is_tensor(depth):
                self._save_image(depth, f"{base}/depth.png")
            else:
                depth.save(f"{base}/depth.png")
        np.save(f"{base}/point_uv.npy", self.point_uv.detach().cpu().numpy())
        np.save(f"{base}/viewpoint.npy", np.asarray(self.view))
        torch.save({"view": self.cam.detach().cpu().reshape(3, 4), "eye": torch.as_tensor(np.asarray(self.view, np.float64)),
                    "focal": float(self.focal), "fovy": float(self.cfg.fovy),
                    "cam_res": int(getattr(self.cfg, "cam_res", self.cfg.res))}, f"{base}/camera.pth")

    def uvToPixels(self, uvs, res):
        """DepthPrompting.py:179-184: (uv*res).long(), swap to (row, col), clip."""
        uvs = uvs.contiguous().float()
        pix = torch.empty(uvs.shape[0], 2, device=uvs.device, dtype=torch.int32)
        rc = _lib.on_device_of(uvs, _L.genpc_uv_to_pixels, uvs.shape[0], _p(uvs), float(res), int(res) - 1, _p(pix))
        if rc != 1:
            raise RuntimeError("genpc_uv_to_pixels failed: " + _lib.last_error())
        return pix

    # DepthPrompting.py:292-339
    def paintPixels(self, img, pixel_coords, pixel_colors, point_size):
        """img [C,res,res] is painted in place; returns the vertically flipped image."""
        n = pixel_coords.shape[0]
        ch = img.shape[0]
        if not torch.is_tensor(pixel_colors):
            pixel_colors = pixel_colors * torch.ones((n, ch), device=img.device)
        pixel_colors = pixel_colors.contiguous().float()
        pix = pixel_coords.to(torch.int32).contiguous()
        _lib.check_tensors((("img", img), ("pixel_colors", pixel_colors)), (("pixel_coords", pix),))
        res = img.shape[1]
        out = torch.empty_like(img)
        owner = torch.empty(res * res, device=img.device, dtype=torch.int32)
        rc = _lib.on_device_of(img, _L.genpc_paint_pixels, res, n, _p(pix), _p(pixel_colors), ch, int(point_size),
                               _p(img), _p(out), _p(owner))
        if rc != 1:
            raise RuntimeError("genpc_paint_pixels failed: " + _lib.last_error())
        return out

    # DepthPrompting.py:341-391
    def getRawDepth(self, point_pixels, point_depth, dataset=None, colors=None, res=512, point_size=1,
                    mask_pixel_rate=3):
        dev = point_depth.device
        R = self.cfg.res
        sparse_img, sparse_depth, all_temp = [torch.zeros((3, R, R), device=dev) for _ in range(3)]
        visible_point_depth = 0.1 + 0.8 * (
            1 - (point_depth - point_depth.min()) / (point_depth.max() - point_depth.min())
        ).unsqueeze(1).expand(-1, 3)
        sparse_img = self.paintPixels(sparse_img, point_pixels, colors, point_size=point_size)
        sparse_depth = self.paintPixels(sparse_depth, point_pixels, visible_point_depth.contiguous(),
                                        point_size=point_size)
        all_front_mask = (self.paintPixels(all_temp, point_pixels, colors,
                                           point_size=point_size * mask_pixel_rate) != 0).float()
        all_back_mask = 1 - all_front_mask
        front_mask = (sparse_img != 0).float()
        back_mask = 1 - front_mask
        hole_mask1 = ((all_back_mask * 255).int() ^ (back_mask * 255).int()).float() / 255
        hole_mask2 = ((all_front_mask * 255).int() ^ (back_mask * 255).int()).float() / 255
        return sparse_img, sparse_depth, hole_mask1, hole_mask2
