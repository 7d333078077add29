"""ICP + scale search of the reference's ``reg_xyz.py`` on the gfx950 library
(SURVEY.md 8a row a17): ``icp_with_scaling`` (:24-38), ``icp_with_scaling_xyz``
(:9-21), the coarse 11-scale sweep of ``reg`` (:146-173) and
``iterative_scale_search`` (:60-96).  Same names, argument meaning and returned
quantities; clouds are [N,3] GPU tensors (the reference passes open3d point clouds;
open3d is absent and unpinned -- see DESIGN.md).  Voxel down-sampling, noise
removal and the fusion tail of ``reg`` are the "next" rows of SURVEY 8f.

Batching: the 11 coarse candidates share their first ICP (it does not depend on
the scale) and run their second ICP as one K=11 batch; the 10x10x10 anisotropic
search scores all 1000 candidates with one batched NN launch and solves ICP only
for the winner, because the reference's score (:77-83) is computed on the scaled
source, before and independently of the candidate's ICP result.
"""
import numpy as np
import torch

from . import _lib
from .loss_functions import chamfer_3DDist

_L = _lib.lib
_p = _lib.ptr


def registration_icp(source, target, max_correspondence_distance, init=None, max_iteration=30,
                     relative_fitness=1e-6, relative_rmse=1e-6):
    """open3d.pipelines.registration.registration_icp(point-to-point) for K initial
    transforms at once.  init: [4,4] or [K,4,4] (numpy / tensor, float64).
    Returns (transformation [K,4,4] float64 numpy, fitness [K], inlier_rmse [K], iters [K]);
    a single [4,4] init returns unbatched values."""
    source = source.contiguous().float()
    target = target.contiguous().float()
    _lib.check_tensors((("source", source), ("target", target)))
    dev = source.device
    if init is None:
        init = np.eye(4)
    init_t = torch.as_tensor(np.asarray(init, np.float64)).reshape(-1, 4, 4).contiguous().to(dev)
    k = init_t.shape[0]
    buf = torch.empty(k * 19, dtype=torch.float64, device=dev)          # transforms and statistics: one copy back
    out_T, stats = buf[:k * 16], buf[k * 16:]
    rc = _lib.on_device_of(source, _L.genpc_icp_batch, k, source.shape[0], _p(source), target.shape[0], _p(target),
                           float(max_correspondence_distance), _p(init_t), int(max_iteration),
                           float(relative_fitness), float(relative_rmse), _p(out_T), _p(stats))
    if rc != 1:
        raise RuntimeError("genpc_icp_batch failed (rc=%d): %s" % (rc, _lib.last_error()))
    host = buf.cpu().numpy()
    T = host[:k * 16].reshape(k, 4, 4)
    st = host[k * 16:].reshape(k, 3)
    if np.asarray(init).ndim == 2:
        return T[0], float(st[0, 0]), float(st[0, 1]), int(st[0, 2])
    return T, st[:, 0], st[:, 1], st[:, 2].astype(int)


def icp_with_scaling_xyz(source, target, scales, max_correspondence_distance=0.05, init_transform=None):
    """reg_xyz.py:9-21: the source is scaled per axis first (the reference does it in
    place on the open3d cloud), then ICP.  Returns (transformation, scaled_source)."""
    s = torch.as_tensor(np.asarray(scales, np.float64), device=source.device)
    scaled = (source.double() * s).float()
    T, _, _, _ = registration_icp(scaled, target, max_correspondence_distance,
                                  np.eye(4) if init_transform is None else init_transform)
    return T, scaled


def icp_with_scaling(source, target, scale, max_correspondence_distance=0.05, init_transform=None):
    """reg_xyz.py:24-38: ICP, then ICP again from result @ diag(scale).  `scale` may be
    a sequence: the first ICP is shared and the second runs as one batch."""
    init = np.eye(4) if init_transform is None else np.asarray(init_transform, np.float64)
    T1, _, _, _ = registration_icp(source, target, max_correspondence_distance, init)
    scales = np.atleast_1d(np.asarray(scale, np.float64))
    inits = []
    for sc in scales:
        S = np.eye(4)
        S[:3, :3] *= sc
        inits.append(T1 @ S)
    T2, fit, rmse, its = registration_icp(source, target, max_correspondence_distance, np.stack(inits))
    if np.ndim(scale) == 0:
        return T2[0]
    return T2


def _partial_cd_scores(src_b, tgt_b, cd_inv_weight):
    """chamfer_partial_l1(src, tgt) + w * chamfer_partial_l1(tgt, src) per batch row."""
    d1, d2, _, _ = chamfer_3DDist()(src_b, tgt_b)
    return torch.sqrt(d1).mean(1) + torch.sqrt(d2).mean(1) * cd_inv_weight


def coarse_scale_sweep(source_down, target_down, cd_inv_weight=0.5, scales=None, max_correspondence_distance=0.075):
    """The 11-scale sweep of reg() (reg_xyz.py:146-173).  Returns (best_scale, best_loss,
    coarse_transformation) with the reference's strict '<' (first minimum wins)."""
    if scales is None:
        scales = np.linspace(1.5, 0.8, 11)
    T = icp_with_scaling(source_down, target_down, scales, max_correspondence_distance)      # [K,4,4]
    inv = torch.as_tensor(np.linalg.inv(T), device=source_down.device)                        # :161-162
    tgt = target_down.double()
    tgt_k = (tgt @ inv[:, :3, :3].transpose(1, 2) + inv[:, None, :3, 3]).float().contiguous()
    src_k = source_down.float()[None].expand(len(scales), -1, -1).contiguous()
    cd = _partial_cd_scores(src_k, tgt_k, cd_inv_weight).cpu().numpy()
    best = int(np.argmin(cd))             # np.argmin returns the first minimum == strict '<' update
    return float(scales[best]), float(cd[best]), T[best]


def iterative_scale_search(source_pcd, target_pcd, scale_ranges, scale_steps, init_transform=None, cd_inv_weight=0):
    """reg_xyz.py:60-96.  Returns (best_scales_transformation [4,4], best_loss,
    best_transformation [4,4]).  Candidate order z (outer), x, y (inner), strict '<'."""
    source = source_pcd.contiguous().float()
    target = target_pcd.contiguous().float()
    _lib.check_tensors((("source", source), ("target", target)))
    xs = np.linspace(scale_ranges[0][0], scale_ranges[0][1], scale_steps)
    ys = np.linspace(scale_ranges[1][0], scale_ranges[1][1], scale_steps)
    zs = np.linspace(scale_ranges[2][0], scale_ranges[2][1], scale_steps)
    # candidate order z (outer), x, y (inner)
    gz, gx, gy = np.meshgrid(zs, xs, ys, indexing="ij")
    cand = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1).astype(np.float64)
    scales_t = torch.from_numpy(cand.astype(np.float32)).to(source.device)
    scores = torch.empty(len(cand), device=source.device)
    rc = _lib.on_device_of(source, _L.genpc_scale_search_scores, len(cand), source.shape[0], _p(source),
                           target.shape[0], _p(target), _p(scales_t), float(cd_inv_weight), _p(scores))
    if rc != 1:
        raise RuntimeError("genpc_scale_search_scores failed (rc=%d): %s" % (rc, _lib.last_error()))
    sc = scores.cpu().numpy()
    best = int(np.argmin(sc))
    best_scales = cand[best]
    best_T, _ = icp_with_scaling_xyz(source, target, best_scales, max_correspondence_distance=0.075,
                                     init_transform=np.eye(4) if init_transform is None else init_transform)
    S = np.eye(4)
    S[0, 0], S[1, 1], S[2, 2] = best_scales
    return S, float(sc[best]), best_T


# ---------------------------------------------------------------------------
# reg(): the composition of the stages above (reg_xyz.py:99-205) on tensors.
# ---------------------------------------------------------------------------
def get_rotate_matrix(axis, angle):
    """utils/dataUtils.py:455-472 (degrees)."""
    a = angle * np.pi / 180
    c, s = np.cos(a), np.sin(a)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    if axis == "z":
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    raise ValueError("axis should be x,y,z")


def normalize_numpy(xyz, range=1.0):
    """utils/dataUtils.py:561-581 on a tensor: bbox-centre, divide by the largest
    extent, scale to [-range, range].  Returns (normalised, centre, scale_factor)."""
    vmin, vmax = xyz.min(0).values, xyz.max(0).values
    center = (vmax + vmin) / 2.0
    scale_factor = (vmax - vmin).max()
    return (xyz - center) / scale_factor * (range / 0.5), center, scale_factor


def voxel_down_sample(xyz, voxel_size, colors=None):
    """Counterpart of open3d's PointCloud.voxel_down_sample (reg_xyz.py:154-155; open3d
    absent and unpinned): grid anchored at min_bound - voxel/2, one output point per
    occupied voxel = mean of its points (double, point order); `colors` [N,3], when given, are
    averaged per voxel the same way and returned as a second tensor.  Output order: ascending
    voxel index (open3d's is its hash-map order).  One call into the HIP library
    (csrc/voxel.hip: keys, radix sort, segmented mean).  voxel_size goes down as a double."""
    pts = xyz.contiguous().float()
    _lib.check_tensors((("xyz", pts),))
    n = pts.shape[0]
    col = None
    if colors is not None:
        col = colors.contiguous().float()
        _lib.check_tensors((("colors", col),))
        if col.shape != pts.shape:
            raise ValueError("voxel_down_sample: colors must be [N,3] like xyz")
    out = torch.empty(n, 3, device=pts.device)
    outc = torch.empty(n, 3, device=pts.device) if col is not None else None
    cnt = torch.empty(1, device=pts.device, dtype=torch.int32)
    rc = _lib.on_device_of(pts, _L.genpc_voxel_down_sample, n, _p(pts), _p(col), float(voxel_size), _p(out), _p(outc), _p(cnt))
    if rc == -1:
        raise ValueError("voxel_down_sample: voxel_size must be positive")
    if rc != 1:
        raise RuntimeError("genpc_voxel_down_sample failed: " + _lib.last_error())
    k = int(cnt.item())
    if k < 0:
        raise ValueError("voxel_down_sample: non-finite coordinate, or more than 2^21 voxels along an axis")
    if col is None:
        return out[:k].to(xyz.dtype)
    return out[:k].to(xyz.dtype), outc[:k].to(colors.dtype)


def _apply(T, xyz):
    """open3d PointCloud.transform: p <- T[:3,:3] p + T[:3,3] (double, like open3d)."""
    Tt = torch.as_tensor(np.asarray(T, np.float64), device=xyz.device)
    return (xyz.double() @ Tt[:3, :3].T + Tt[:3, 3]).to(xyz.dtype)


def _is_cfg(x):
    return hasattr(x, "output_path") and not torch.is_tensor(x)


def reg(cfg, flag, cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=False, **kwargs):
    """reg_xyz.py:99-223, same signature.

    (Deviation, file form: the PLY's double coordinates are cast to float32 before the voxel grids hash them; open3d
    hashes the doubles, so a point within float32 rounding of a voxel face can land in the neighbouring voxel.)
    File form (the reference's): ``reg(cfg, flag, cd_inv_weight, diff_init, reg_fine_xyz)`` reads
    ``{cfg.output_path}/{flag}/color_point.ply`` (the partial cloud with the colours colorPoint gave
    it) and ``{flag}_{cfg.generative_model}.glb`` (the generated mesh; 163 840 surface samples with
    their colours), aligns them and writes ``{flag}_fused.ply`` WITH COLOURS (:207-219); returns the
    dict of `reg_tensors` plus ``fused`` / ``fused_col`` / ``fused_path``.  Raises FileNotFoundError
    like the reference when an input is missing.
    Tensor form: ``reg(partial_xyz, complete_xyz, ...)`` = `reg_tensors` (no files)."""
    if not _is_cfg(cfg):
        return reg_tensors(cfg, flag, cd_inv_weight=cd_inv_weight, diff_init=diff_init, reg_fine_xyz=reg_fine_xyz, **kwargs)
    import os
    from .optim_registration.diff_obj_pose import object_pose_optimization
    from .utils.dataUtils import read_ply, save_ply_xyzrgb
    from .utils.mesh_io import glb2point
    unknown = set(kwargs) - {"cd_only_pose", "rng"}
    if unknown:          # (the file form reads two options; anything else -- a misspelt one -- used to be dropped silently: ADVICE r3)
        raise TypeError("reg(cfg, flag, ...): unexpected keyword argument(s) %s (the file form takes cd_only_pose, rng)" % sorted(unknown))
    path = cfg.output_path
    ply = f"{path}/{flag}/color_point.ply"
    glb = f"{path}/{flag}/{flag}_{cfg.generative_model}.glb"
    for f in (ply, glb):                                                # :103-108
        if not os.path.exists(f):
            print(f"Path {f} does not exist.")
            raise FileNotFoundError(f"Path {f} does not exist.")
    dev = torch.device(getattr(cfg, "device", "cuda"))
    diff_transform = None
    if diff_init:                                                       # :109-122
        diff_transform = np.linalg.inv(object_pose_optimization(
            glb_path=glb, point_path=ply, radius=0.02, lr=0.01, iters=200, render_size=224, vis=True, device=dev,
            cd_only=kwargs.get("cd_only_pose", False)).astype(np.float64))
    sxyz, scol = read_ply(ply)                                          # :124 o3d.io.read_point_cloud
    txyz, tcol = glb2point(glb, num_points=163840, rng=kwargs.get("rng"))                      # :125
    if scol is None:
        scol = np.zeros_like(sxyz)                                      # an open3d cloud without colours: fused file stays valid
    out = reg_tensors(torch.as_tensor(sxyz, dtype=torch.float32, device=dev), torch.as_tensor(txyz, dtype=torch.float32, device=dev),
                      generative_model=cfg.generative_model, dataset=getattr(cfg, "dataset", "redwood"),
                      cd_inv_weight=cd_inv_weight, diff_init=diff_init, reg_fine_xyz=reg_fine_xyz,
                      partial_col=torch.as_tensor(scol, dtype=torch.float32, device=dev),
                      complete_col=torch.as_tensor(tcol, dtype=torch.float32, device=dev), diff_transform=diff_transform)
    fused, fused_col = fuse(out["source"], out["target"], num_points=20000, distance_threshold=0.0001, std_ratio=2.5,
                            source_col=out["source_col"], target_col=out["target_col"])       # :207-217
    out.update(fused=fused, fused_col=fused_col, fused_path=f"{path}/{flag}/{flag}_fused.ply")
    save_ply_xyzrgb(fused.double().cpu().numpy(), fused_col.double().cpu().numpy(), out["fused_path"])     # :221
    return out


def reg_tensors(partial_xyz, complete_xyz, generative_model="trellis", dataset="redwood", cd_inv_weight=0.5,
                diff_init=True, reg_fine_xyz=False, pose_voxel=0.02, cd_only_pose=False, partial_col=None,
                complete_col=None, diff_transform=None):
    """reg_xyz.py:99-205 on tensors, without file I/O and without the fusion tail.
    partial_xyz: the observed cloud (color_point.ply), complete_xyz: points sampled from
    the generated mesh (glb2point); partial_col / complete_col: their colours ([N,3] in [0,1], None =
    white in the pose loss) -- they feed the silhouette term of the pose initialisation and are carried
    to the outputs.  diff_transform: a precomputed inverse pose (the file form computes it from its own
    120 000-point sampling like the reference); None runs object_pose_optimization on both clouds
    voxel-down-sampled at `pose_voxel` (load_point_cloud's radius).  Returns a dict with the aligned
    clouds (`source`, `target`: both back in the partial cloud's original frame, as at :200-205;
    `source_col`, `target_col`) and every intermediate transform."""
    from .optim_registration.diff_obj_pose import object_pose_optimization
    source = partial_xyz.contiguous().float()
    target = complete_xyz.contiguous().float()
    scol = None if partial_col is None else partial_col.contiguous().float()
    tcol = None if complete_col is None else complete_col.contiguous().float()
    out = {}
    if not diff_init:
        diff_transform = np.eye(4)
    elif diff_transform is None:                                        # :109-122
        if tcol is not None:
            tv, tvc = voxel_down_sample(target, pose_voxel, colors=tcol)
        else:
            tv, tvc = voxel_down_sample(target, pose_voxel), None
        if scol is not None:
            sv, svc = voxel_down_sample(source, pose_voxel, colors=scol)
        else:
            sv, svc = voxel_down_sample(source, pose_voxel), None
        T = object_pose_optimization(tv, sv, radius=0.02, lr=0.01, iters=200, render_size=224, cd_only=cd_only_pose,
                                     complete_col=tvc, partial_col=svc)
        diff_transform = np.linalg.inv(T.astype(np.float64))
    out["diff_transform"] = diff_transform
    source = _apply(diff_transform, source)                             # :126
    target, _, _ = normalize_numpy(target, range=0.5)                   # :130
    if generative_model in ("instantmesh",):                            # :132-137
        source, keep = remove_noise_from_point_cloud(source)
        if scol is not None:
            scol = scol[keep]
        target = (target.double() @ torch.as_tensor(get_rotate_matrix("x", 90).T, device=target.device)
                  @ torch.as_tensor(get_rotate_matrix("y", 90).T, device=target.device)).float()
    best_scale, best_loss, coarse = coarse_scale_sweep(voxel_down_sample(source, 0.03), voxel_down_sample(target, 0.03),
                                                       cd_inv_weight=cd_inv_weight)            # :146-173
    out.update(best_scale=best_scale, coarse_loss=best_loss, coarse_transformation=coarse)
    if reg_fine_xyz:                                                    # :176-199
        source = _apply(coarse, source)
        tv = 0.04 if dataset in ("pcn", "kitti") else 0.03
        src_s = source if dataset in ("pcn", "kitti") else voxel_down_sample(source, 0.03)
        S, loss_xyz, T_xyz = iterative_scale_search(src_s, voxel_down_sample(target, tv), [(0.8, 1.2)] * 3, 10,
                                                    init_transform=np.eye(4), cd_inv_weight=cd_inv_weight)
        out.update(best_scales_transformation=S, xyz_loss=loss_xyz, best_transformation_xyz=T_xyz)
        target = _apply(np.linalg.inv(S), target)
        target = _apply(np.linalg.inv(T_xyz), target)
        source = _apply(np.linalg.inv(coarse), source)
    target = _apply(np.linalg.inv(coarse), target)                      # :201-205
    target = _apply(np.linalg.inv(diff_transform), target)
    source = _apply(np.linalg.inv(diff_transform), source)
    out.update(source=source, target=target, source_col=scol, target_col=tcol)
    return out


# ---------------------------------------------------------------------------
# Fusion tail of reg() (reg_xyz.py:207-219), SURVEY 8f row f2.
# ---------------------------------------------------------------------------
def remove_close_points(source_xyz, target_xyz, distance_threshold=0.0001):
    """reg_xyz.py:41-57: drop every target point whose nearest source point is closer
    than sqrt(distance_threshold) (open3d's KD-tree returns SQUARED distances, so the
    reference's 1e-4 is 0.01 units).  One NN launch instead of a Python loop of KD-tree
    queries (measured at 163840 x 16384: 147 us on the default MFMA filter; the radius-limited
    cell search, chamfer_3D.nm_distance_within, gives the same mask in 156 us -- no gain yet).
    Returns (filtered target [K,3], keep mask [M]) -- index colours with the mask (:56)."""
    from . import chamfer_3D
    src = source_xyz.contiguous().float()
    tgt = target_xyz.contiguous().float()
    d = torch.empty(1, tgt.shape[0], device=tgt.device)
    i = torch.empty(1, tgt.shape[0], device=tgt.device, dtype=torch.int32)
    if tgt.shape[0] == 0 or src.shape[0] == 0:
        return tgt, torch.ones(tgt.shape[0], dtype=torch.bool, device=tgt.device)
    if chamfer_3D.nm_distance(tgt[None], src[None], d, i) != 1:
        raise RuntimeError("nm_distance failed: " + _lib.last_error())
    keep = ~(d[0] < distance_threshold)
    return tgt[keep], keep


def fuse(source_xyz, target_xyz, num_points=20000, distance_threshold=0.0001, std_ratio=2.5, source_col=None,
         target_col=None, side_fps=None):
    """reg_xyz.py:207-217: partial + (complete minus what the partial already covers),
    farthest-point-sampled to `num_points`, then the statistical outlier filter
    (std_ratio 2.5 as at :217; None skips it).  With source_col / target_col the colours follow their
    points through all three steps (:56,:212-216) and (fused, fused_col) is returned.
    side_fps: [(cloud [N,3], k), ...] -- independent subsamplings the caller needs anyway (the metric's
    ground-truth cloud, main.py:21) ride along in the fused cloud's FPS launch (a pass costs its longest
    chain of sequential steps, not the sum); their index tensors are appended to the return value."""
    from .fps import fps_sampling, fps_sampling_multi
    with_col = source_col is not None and target_col is not None
    filtered, keep = remove_close_points(source_xyz, target_xyz, distance_threshold)
    fused = torch.cat([source_xyz.float(), filtered], dim=0).contiguous()
    col = torch.cat([source_col.float(), target_col.float()[keep]], dim=0) if with_col else None
    side = []
    if side_fps:
        need = fused.shape[0] > num_points
        clouds = ([fused] if need else []) + [c.contiguous().float() for c, _ in side_fps]
        ks = ([num_points] if need else []) + [k for _, k in side_fps]
        got = fps_sampling_multi(clouds, ks)
        if need:
            idx = got[0].long()
            fused = fused[idx]
            col = col[idx] if with_col else None
        side = got[1:] if need else got
    elif fused.shape[0] > num_points:
        idx = fps_sampling(fused, num_points).long()
        fused = fused[idx]
        col = col[idx] if with_col else None
    if std_ratio is not None:                                            # :217
        fused, ok = remove_noise_from_point_cloud(fused, std_ratio=std_ratio)
        col = col[ok] if with_col else None
    out = (fused, col) if with_col else fused
    if side_fps:
        return out, side
    return out


def knn_mean_distance(xyz, k=20):
    """Mean distance of every point to its k nearest points of the same cloud (the
    point itself included) -- the statistic of open3d's remove_statistical_outlier."""
    pts = xyz.contiguous().float()
    _lib.check_tensors((("xyz", pts),))
    out = torch.empty(pts.shape[0], device=pts.device)
    rc = _lib.on_device_of(pts, _L.genpc_knn_mean_distance, pts.shape[0], _p(pts), int(k), _p(out))
    if rc == -1:
        raise ValueError("knn_mean_distance: k must be 8, 16, 20 or 32")
    if rc != 1:
        raise RuntimeError("genpc_knn_mean_distance failed: " + _lib.last_error())
    return out


def remove_noise_from_point_cloud(xyz, nb_neighbors=20, std_ratio=1.5):
    """utils/dataUtils.py:648-662 (open3d remove_statistical_outlier) on a tensor: keeps
    points whose mean k-NN distance is below mean + std_ratio * std (sample std, double).
    Returns (filtered [K,3], keep mask [N])."""
    m = knn_mean_distance(xyz, nb_neighbors).double()
    mean = m.mean()
    std = torch.sqrt(((m - mean) ** 2).sum() / (m.numel() - 1))
    keep = m < mean + std_ratio * std
    return xyz[keep], keep
