"""ctypes front-end of the CPU oracle (oracle/genpc_oracle*.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, by bench.py's
``cpu_baseline`` leg and by ``__graft_entry__.smoke()`` -- never by genpc_amd/.
Parity status: the reference's own pure-Python functions pin the rows they cover
(tests/golden/ref_py_*.npz, see genpc_oracle_geom.c); for the CUDA kernels see the
header of genpc_oracle.c ("parity unpinned" by the
reference's own artefacts; pinned against BASELINE.md section 2 values).

All arrays are numpy, C-contiguous, float32 / int32, shaped like the reference's
tensors ([B,N,3] clouds, [B,N] distances and indices).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgenpc_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc, -ffp-contract=off)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(t):
    lib().oracle_set_num_threads(int(t))


# --------------------------------------------------------------------------
# Chamfer (chamfer3D.cu)
# --------------------------------------------------------------------------
def chamfer_forward(xyz1, xyz2, fma_mode=1):
    """-> dist1[B,N], dist2[B,M] (squared), idx1, idx2 (int32)."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), np.float32)
    d2 = np.zeros((b, m), np.float32)
    i1 = np.zeros((b, n), np.int32)
    i2 = np.zeros((b, m), np.int32)
    rc = lib().oracle_chamfer_forward(
        b, n, p1, m, p2, d1.ctypes.data_as(_f32p), i1.ctypes.data_as(_i32p),
        d2.ctypes.data_as(_f32p), i2.ctypes.data_as(_i32p), int(fma_mode))
    assert rc == 1
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    """-> gradxyz1[B,N,3], gradxyz2[B,M,3] (sequential accumulation order)."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    g1, pg1 = _f(graddist1)
    g2, pg2 = _f(graddist2)
    i1, pi1 = _i(idx1)
    i2, pi2 = _i(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gx1 = np.zeros((b, n, 3), np.float32)
    gx2 = np.zeros((b, m, 3), np.float32)
    rc = lib().oracle_chamfer_backward(
        b, n, p1, m, p2, pg1, pi1, pg2, pi2,
        gx1.ctypes.data_as(_f32p), gx2.ctypes.data_as(_f32p))
    assert rc == 1
    return gx1, gx2


# --------------------------------------------------------------------------
# EMD (emd_cuda.cu)
# --------------------------------------------------------------------------
def emd_forward(xyz1, xyz2, eps, iters, fma_mode=1, return_state=False):
    """-> dist[B,n] (squared), assignment[B,n] int32.  rc -1 -> ValueError."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    st = dict(
        dist=np.zeros((b, n), np.float32),
        assignment=np.full((b, n), -1, np.int32),
        price=np.zeros((b, m), np.float32),
        assignment_inv=np.full((b, m), -1, np.int32),
        bid=np.zeros((b, n), np.int32),
        bid_increments=np.zeros((b, n), np.float32),
        max_increments=np.zeros((b, m), np.float32),
        unass_idx=np.zeros(b * n, np.int32),
        unass_cnt=np.zeros(512, np.int32),
        unass_cnt_sum=np.zeros(512, np.int32),
        cnt_tmp=np.zeros(512, np.int32),
        max_idx=np.zeros(b * m, np.int32),
    )
    c = {k: (v.ctypes.data_as(_f32p) if v.dtype == np.float32 else v.ctypes.data_as(_i32p))
         for k, v in st.items()}
    rc = lib().oracle_emd_forward(
        b, n, m, p1, p2, c["dist"], c["assignment"], c["price"], c["assignment_inv"],
        c["bid"], c["bid_increments"], c["max_increments"], c["unass_idx"],
        c["unass_cnt"], c["unass_cnt_sum"], c["cnt_tmp"], c["max_idx"],
        ctypes.c_float(eps), int(iters), int(fma_mode))
    if rc != 1:
        raise ValueError("oracle_emd_forward rc=%d (n!=m, B>512 or n%%256!=0)" % rc)
    if return_state:
        return st["dist"], st["assignment"], st
    return st["dist"], st["assignment"]


def emd_backward(xyz1, xyz2, graddist, assignment):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    g, pg = _f(graddist)
    a, pa = _i(assignment)
    b, n, _ = xyz1.shape
    gx = np.zeros((b, n, 3), np.float32)
    rc = lib().oracle_emd_backward(b, n, p1, p2, gx.ctypes.data_as(_f32p), pg, pa)
    assert rc == 1
    return gx


# --------------------------------------------------------------------------
# Reductions of utils/loss_util.py:25-49 (Completionloss), fp32 means like
# torch.mean on float32 tensors (pairwise summation in both numpy and torch;
# values agree to the last ulp or two -- tests state the tolerance).
# --------------------------------------------------------------------------
def cd_l1(d1, d2):
    return (np.sqrt(d1).mean(dtype=np.float32) + np.sqrt(d2).mean(dtype=np.float32)) / np.float32(2)


def cd_l2(d1, d2):
    return d1.mean(dtype=np.float32) + d2.mean(dtype=np.float32)


def cd_partial_l1(d1):
    return np.sqrt(d1).mean(dtype=np.float32)


def cd_partial_l2(d1):
    return d1.mean(dtype=np.float32)


def emd_loss(dist):
    return np.sqrt(dist).mean(axis=1, dtype=np.float32).mean(dtype=np.float32)


# --------------------------------------------------------------------------
# Fixture helpers
# --------------------------------------------------------------------------
def fps(xyz, k, fma_mode=0):
    """Deterministic farthest point sampling (start 0, first arg-max) -> idx[k].
    fma_mode 0 is what the committed fixtures were subsampled with."""
    xyz, p = _f(xyz)
    out = np.zeros(k, np.int32)
    lib().oracle_fps_mode(xyz.shape[0], p, int(k), int(fma_mode), out.ctypes.data_as(_i32p))
    return out


def read_ply_xyz(path):
    """Binary-little-endian PLY with double/float x,y,z vertex properties (the
    layout Open3D wrote for the reference's data/*.ply) -> float64 [N,3]."""
    with open(path, "rb") as f:
        header = b""
        while not header.endswith(b"end_header\n"):
            line = f.readline()
            if not line:
                raise ValueError("bad PLY header")
            header += line
        lines = header.decode("ascii").split("\n")
        assert "format binary_little_endian 1.0" in lines[1], lines[1]
        nv = 0
        props = []
        in_vertex = False
        for ln in lines:
            tok = ln.split()
            if not tok:
                continue
            if tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    nv = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                props.append((tok[2], {"double": "<f8", "float": "<f4", "uchar": "u1",
                                       "int": "<i4", "uint": "<u4", "short": "<i2",
                                       "ushort": "<u2", "char": "i1"}[tok[1]]))
        dt = np.dtype(props)
        data = np.frombuffer(f.read(nv * dt.itemsize), dtype=dt, count=nv)
    return np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float64)


# --------------------------------------------------------------------------
# Projection / splat / colour gather / pose (genpc_oracle_geom.c)
# --------------------------------------------------------------------------
_f64p = ctypes.POINTER(ctypes.c_double)


def look_at(eye, at, up):
    """-> view[12]: 3x4 row-major world->camera matrix (camera looks down -Z)."""
    e, pe = _f(eye)
    a, pa = _f(at)
    u, pu = _f(up)
    v = np.zeros(12, np.float32)
    lib().oracle_look_at(pe, pa, pu, v.ctypes.data_as(_f32p))
    return v


def calculate_up_vector(eye, target):
    e = np.ascontiguousarray(eye, np.float64)
    t = np.ascontiguousarray(target, np.float64)
    u = np.zeros(3, np.float64)
    lib().oracle_calculate_up_vector(e.ctypes.data_as(_f64p), t.ctypes.data_as(_f64p), u.ctypes.data_as(_f64p))
    return u


def fibonacci_sphere(samples, radius):
    """utils/camera_utils.py:84-101 restated (float64, like the reference)."""
    import math
    pts = []
    phi = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(samples):
        y = 1 - (i / float(samples - 1)) * 2
        radius_y = math.sqrt(1 - y * y)
        theta = phi * i
        pts.append((math.cos(theta) * radius_y * radius, y * radius, math.sin(theta) * radius_y * radius))
    return np.array(pts)


def get_uvs(views, focal, xyz, rescale=True, padding=0.15, near=1e-2, far=1e2):
    """DepthPrompting.getUvs -> uv[C,N,2], depth[C,N], transformed[C,N,3], bbox[C,4]."""
    views, pv = _f(np.asarray(views, np.float32).reshape(-1, 12))
    xyz, px = _f(xyz)
    c, n = views.shape[0], xyz.shape[0]
    tr = np.zeros((c, n, 3), np.float32)
    uv = np.zeros((c, n, 2), np.float32)
    dp = np.zeros((c, n), np.float32)
    bb = np.zeros((c, 4), np.float32)
    lib().oracle_get_uvs(c, n, pv, ctypes.c_float(focal), ctypes.c_float(near), ctypes.c_float(far), px,
                         tr.ctypes.data_as(_f32p), uv.ctypes.data_as(_f32p), dp.ctypes.data_as(_f32p),
                         int(bool(rescale)), ctypes.c_float(np.float32(1 - 2 * padding)), bb.ctypes.data_as(_f32p))
    return uv, dp, tr, bb


def rescale_uvs(transformed, rescale=True, padding=0.15):
    """getUvs' own arithmetic (DepthPrompting.py:246-268) on given transformed points [C,N,3] -> uv, depth."""
    t, pt = _f(transformed)
    c, n = t.shape[0], t.shape[1]
    uv = np.zeros((c, n, 2), np.float32)
    dp = np.zeros((c, n), np.float32)
    lib().oracle_rescale_uvs(c, n, pt, int(bool(rescale)), ctypes.c_float(np.float32(1 - 2 * padding)),
                             uv.ctypes.data_as(_f32p), dp.ctypes.data_as(_f32p))
    return uv, dp


def uv_to_pixels(uv, res, clip_max=None):
    uv, pu = _f(uv)
    n = uv.shape[0]
    pix = np.zeros((n, 2), np.int32)
    lib().oracle_uv_to_pixels(n, pu, ctypes.c_float(res), int(res - 1 if clip_max is None else clip_max),
                              pix.ctypes.data_as(_i32p))
    return pix


def paint_pixels(res, pix, colors, point_size, img=None):
    """-> (flipped image [C,res,res], painted-in-place img)."""
    pix, pp = _i(pix)
    colors, pc = _f(colors)
    ch = colors.shape[1]
    if img is None:
        img = np.zeros((ch, res, res), np.float32)
    img, pi = _f(img)
    out = np.zeros_like(img)
    lib().oracle_paint_pixels(int(res), pix.shape[0], pp, pc, ch, int(point_size), pi, out.ctypes.data_as(_f32p))
    return out, img


def get_raw_depth(pix, depth, colors, res, point_size=1, mask_pixel_rate=3):
    """DepthPrompting.getRawDepth (:341-391) composed from paint_pixels: -> sparse_img, sparse_depth,
    hole_mask1, hole_mask2, each [3,res,res] float32.  Grey level (:362-366) in float32 like torch."""
    d = np.ascontiguousarray(depth, np.float32)
    one, p1, p8 = np.float32(1.0), np.float32(0.1), np.float32(0.8)
    grey = p1 + p8 * (one - (d - d.min()) / (d.max() - d.min()))
    sparse_img, _ = paint_pixels(res, pix, colors, point_size)
    sparse_depth, _ = paint_pixels(res, pix, np.repeat(grey[:, None], 3, 1), point_size)
    all_front, _ = paint_pixels(res, pix, colors, point_size * mask_pixel_rate)
    all_front = (all_front != 0).astype(np.float32)
    all_back = 1 - all_front
    back = 1 - (sparse_img != 0).astype(np.float32)
    h1 = ((all_back * 255).astype(np.int32) ^ (back * 255).astype(np.int32)).astype(np.float32) / 255
    h2 = ((all_front * 255).astype(np.int32) ^ (back * 255).astype(np.int32)).astype(np.float32) / 255
    return sparse_img, sparse_depth, h1, h2


def gather_colors(pix, img):
    pix, pp = _i(pix)
    img, pi = _f(img)
    ch, h, w = img.shape
    out = np.zeros((pix.shape[0], ch), np.float32)
    lib().oracle_gather_colors(pix.shape[0], pp, pi, ch, h, w, out.ctypes.data_as(_f32p))
    return out


def rot6d_to_matrix(d6):
    d, pd = _f(d6)
    R = np.zeros(9, np.float32)
    lib().oracle_rot6d_to_matrix(pd, R.ctypes.data_as(_f32p))
    return R.reshape(3, 3)


def pose_transform(v, center, params):
    v, pv = _f(v)
    c, pc = _f(center)
    p, pp = _f(params)
    out = np.zeros_like(v)
    lib().oracle_pose_transform(v.shape[0], pv, pc, pp, out.ctypes.data_as(_f32p))
    return out


def pose_loss_grad(v, center, params, partial, d1, i1, d2, i2, cd_weight=3.0, reg_weight=0.001):
    v, pv = _f(v)
    c, pc = _f(center)
    p, pp = _f(params)
    q, pq = _f(partial)
    d1, pd1 = _f(d1)
    d2, pd2 = _f(d2)
    i1, pi1 = _i(i1)
    i2, pi2 = _i(i2)
    lo = np.zeros(3, np.float32)
    g = np.zeros(10, np.float32)
    lib().oracle_pose_loss_grad(v.shape[0], pv, pc, pp, q.shape[0], pq, pd1, pi1, pd2, pi2,
                                ctypes.c_float(cd_weight), ctypes.c_float(reg_weight),
                                lo.ctypes.data_as(_f32p), g.ctypes.data_as(_f32p))
    return lo, g


def adam_step(params, grad, m, v, step, lr):
    """In place on params/m/v (float32[10])."""
    lib().oracle_adam_step(params.ctypes.data_as(_f32p), np.ascontiguousarray(grad, np.float32).ctypes.data_as(_f32p),
                           m.ctypes.data_as(_f32p), v.ctypes.data_as(_f32p), int(step), ctypes.c_float(lr))


def pose_optimize_cd(complete, partial, lr=0.01, iters=200, starts=4, fma_mode=1):
    """-> (T[4,4], history[starts, iters+1], best_params[10])."""
    c, pc = _f(complete)
    q, pq = _f(partial)
    T = np.zeros(16, np.float32)
    hist = np.zeros((starts, iters + 1), np.float32)
    bp = np.zeros(10, np.float32)
    lib().oracle_pose_optimize_cd(c.shape[0], pc, q.shape[0], pq, ctypes.c_float(lr), int(iters), int(starts),
                                  int(fma_mode), T.ctypes.data_as(_f32p), hist.ctypes.data_as(_f32p),
                                  bp.ctypes.data_as(_f32p))
    return T.reshape(4, 4), hist, bp


def _fopt(x):
    """optional float array -> (array or None, pointer or NULL)"""
    if x is None:
        return None, None
    return _f(x)


def set_blend(blend):
    """0: the build's coverage splat; 1: Pulsar's published blending function (softmax in depth, gamma 1e-2), restated in
    genpc_oracle_geom.c.  Applies to splat_image, pose_full_loss_grad and pose_optimize.  Returns the previous setting."""
    return int(lib().oracle_set_blend(int(blend)))


def set_render_variant(falloff_linear=0, depth_hit=0):
    """What of Pulsar's renderer is restated from memory, switchable in the oracle only (tools/renderer_sensitivity.py):
    falloff_linear 1: coverage 1 - r / rho instead of 1 - r^2 / rho^2; depth_hit 1: the ray-sphere hit instead of the sphere's
    centre enters the softmax's exponent.  (0, 0) is what the kernels implement."""
    lib().oracle_set_render_variant(int(falloff_linear), int(depth_hit))


def splat_image(pts, radius, size, colors=None):
    """Own differentiable colour splat (genpc_oracle_geom.c, PARITY UNPINNED against Pulsar)
    -> [size, size, 3]; colors None = white."""
    p, pp = _f(pts)
    c, pc = _fopt(colors)
    img = np.zeros((size, size, 3), np.float32)
    lib().oracle_splat_image(p.shape[0], pp, pc, ctypes.c_float(radius), int(size), img.ctypes.data_as(_f32p))
    return img


def mask_loss(img, ref, with_grad=False):
    """compute_loss_function's mask_loss on [S,S,3] images (diff_obj_pose.py:286-311); with_grad: also d/d img."""
    a, pa = _f(img)
    r, pr = _f(ref)
    assert a.shape == r.shape and a.ndim == 3 and a.shape[2] == 3 and a.shape[0] == a.shape[1]
    lib().oracle_mask_loss.restype = ctypes.c_float
    if not with_grad:
        return float(lib().oracle_mask_loss(int(a.shape[0]), pa, pr, None))
    g = np.zeros_like(a)
    l = float(lib().oracle_mask_loss(int(a.shape[0]), pa, pr, g.ctypes.data_as(_f32p)))
    return l, g


def pose_full_loss_grad(v, center, params, partial, d1, i1, d2, i2, radius, size, ref_img, cd_weight=3.0,
                        reg_weight=0.001, mask_weight=1.0, vert_col=None):
    """-> (loss[4] = total, cd, ortho, mask ; grad[10]).  ref_img [size,size,3]."""
    v, pv = _f(v)
    vc, pvc = _fopt(vert_col)
    c, pc = _f(center)
    p, pp = _f(params)
    q, pq = _f(partial)
    d1, pd1 = _f(d1)
    d2, pd2 = _f(d2)
    i1, pi1 = _i(i1)
    i2, pi2 = _i(i2)
    ref, pref = _f(ref_img)
    assert ref.shape == (size, size, 3)
    lo = np.zeros(4, np.float32)
    g = np.zeros(10, np.float32)
    lib().oracle_pose_full_loss_grad(v.shape[0], pv, pvc, pc, pp, q.shape[0], pq, pd1, pi1, pd2, pi2,
                                     ctypes.c_float(cd_weight), ctypes.c_float(reg_weight), ctypes.c_float(mask_weight),
                                     ctypes.c_float(radius), int(size), pref, lo.ctypes.data_as(_f32p),
                                     g.ctypes.data_as(_f32p))
    return lo, g


def pose_optimize(complete, partial, lr=0.01, iters=200, starts=4, radius=0.02, size=224, mask_weight=1.0, fma_mode=1,
                  complete_col=None, partial_col=None):
    """Full objective (mask + 3 cd + ortho) -> (T[4,4], history[starts, iters+1], best_params[10])."""
    c, pc = _f(complete)
    q, pq = _f(partial)
    cc, pcc = _fopt(complete_col)
    qc, pqc = _fopt(partial_col)
    T = np.zeros(16, np.float32)
    hist = np.zeros((starts, iters + 1), np.float32)
    bp = np.zeros(10, np.float32)
    lib().oracle_pose_optimize(c.shape[0], pc, pcc, q.shape[0], pq, pqc, ctypes.c_float(lr), int(iters), int(starts),
                               int(fma_mode), ctypes.c_float(radius), int(size), ctypes.c_float(mask_weight),
                               T.ctypes.data_as(_f32p), hist.ctypes.data_as(_f32p), bp.ctypes.data_as(_f32p))
    return T.reshape(4, 4), hist, bp


def kabsch_from_sums(sums):
    s = np.ascontiguousarray(sums, np.float64)
    u = np.zeros(16, np.float64)
    lib().oracle_kabsch_from_sums(s.ctypes.data_as(_f64p), u.ctypes.data_as(_f64p))
    return u.reshape(4, 4)


def icp(source, target, max_dist, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, fma_mode=1):
    """Point-to-point ICP (open3d registration_icp semantics) -> (T[4,4] float64,
    fitness, inlier_rmse, iterations)."""
    s, ps = _f(source)
    t, pt = _f(target)
    init = np.eye(4) if init is None else init
    i0 = np.ascontiguousarray(init, np.float64)
    T = np.zeros(16, np.float64)
    st = np.zeros(3, np.float64)
    lib().oracle_icp(s.shape[0], ps, t.shape[0], pt, ctypes.c_double(max_dist), i0.ctypes.data_as(_f64p),
                     int(max_iter), ctypes.c_double(rel_fitness), ctypes.c_double(rel_rmse), int(fma_mode),
                     T.ctypes.data_as(_f64p), st.ctypes.data_as(_f64p))
    return T.reshape(4, 4), float(st[0]), float(st[1]), int(st[2])


def knn_mean_distance(xyz, k=20, fma_mode=1):
    """Mean distance to the k nearest points of the same cloud (self included)."""
    x, px = _f(xyz)
    out = np.zeros(x.shape[0], np.float32)
    lib().oracle_knn_mean_distance(x.shape[0], px, int(k), int(fma_mode), out.ctypes.data_as(_f32p))
    return out


def statistical_outlier_mask(xyz, nb_neighbors=20, std_ratio=2.0, fma_mode=1):
    """open3d remove_statistical_outlier: keep mask."""
    m = knn_mean_distance(xyz, nb_neighbors, fma_mode).astype(np.float64)
    mean = m.mean()
    std = np.sqrt(((m - mean) ** 2).sum() / (len(m) - 1))
    return m < mean + std_ratio * std


def zbuffer_visibility(uv, depth, res, tol, point_size=2):
    uv, pu = _f(uv)
    depth, pd = _f(depth)
    c, n = depth.shape
    vis = np.zeros((c, n), np.uint8)
    cnt = np.zeros(c, np.int32)
    lib().oracle_zbuffer_visibility(c, n, pu, pd, int(res), int(point_size), ctypes.c_float(tol),
                                    vis.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), cnt.ctypes.data_as(_i32p))
    return vis.astype(bool), cnt


def voxel_down_sample(xyz, voxel_size, colors=None):
    """open3d-style voxel grid mean (published definition; unpinned) -> [K,3] (and the voxel-mean colours
    [K,3] when `colors` is given)."""
    p, pp = _f(xyz)
    c, pc = _fopt(colors)
    out = np.zeros_like(p)
    outc = np.zeros_like(p) if c is not None else None
    k = int(lib().oracle_voxel_down_sample(p.shape[0], pp, pc, ctypes.c_double(voxel_size), out.ctypes.data_as(_f32p),
                                           outc.ctypes.data_as(_f32p) if outc is not None else None))
    if k < 0:
        raise ValueError("voxel_down_sample: bad input")
    if colors is None:
        return out[:k].copy()
    return out[:k].copy(), outc[:k].copy()


def hpr_visibility(points, eye, radius, want_max_vertices=False):
    """Exact Katz visibility by normal-cone clipping (genpc_oracle_hpr.c) -> bool [N]."""
    p, pp = _f(points)
    e = np.ascontiguousarray(eye, dtype=np.float64).reshape(3)
    vis = np.zeros(p.shape[0], np.uint8)
    mv = ctypes.c_int(0)
    k = int(lib().oracle_hpr_visibility(p.shape[0], pp, e.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                        ctypes.c_double(radius), vis.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                        ctypes.byref(mv)))
    if k < 0:
        raise ValueError("hpr_visibility failed (%d)" % k)
    if want_max_vertices:
        return vis.astype(bool), int(mv.value)
    return vis.astype(bool)
