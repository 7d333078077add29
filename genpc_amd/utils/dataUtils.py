"""The few helpers of the reference's utils/dataUtils.py that the geometric path
needs, without open3d / trimesh: binary-little-endian PLY
point I/O in the layout Open3D writes for the reference's data/*.ply (double x,y,z,
optional uchar red,green,blue), ``normalize_numpy`` (:561-581) and
``get_rotate_matrix`` (:455-472).  Host-side format code, numpy only."""
import numpy as np

_PLY_TYPES = {"double": "<f8", "float": "<f4", "float32": "<f4", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "short": "<i2", "ushort": "<u2", "char": "i1"}


def read_ply(path, want_color=True):
    """Binary little-endian and ASCII PLY vertex reader (what open3d's read_point_cloud does for the
    reference's files): returns (xyz float64 [N,3], rgb float64 [N,3] in [0,1] or None)."""
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            header.append(line.decode("ascii", "replace").strip())
            if header[-1] == "end_header":
                break
        if header[0] != "ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt = [h for h in header if h.startswith("format")][0].split()[1]
        nv, props, in_vertex = 0, [], False
        for h in header:
            tok = h.split()
            if not tok:
                continue
            if tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    nv = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("%s: list property in the vertex element" % path)
                props.append((tok[2], _PLY_TYPES[tok[1]]))
        if fmt == "binary_little_endian":
            dt = np.dtype(props)
            data = np.frombuffer(f.read(nv * dt.itemsize), dtype=dt, count=nv)
            cols = {n: data[n] for n, _ in props}
        elif fmt == "ascii":
            arr = np.loadtxt(f, max_rows=nv, ndmin=2)
            cols = {n: arr[:, i] for i, (n, _) in enumerate(props)}
        else:
            raise ValueError("%s: unsupported PLY format %s" % (path, fmt))
    xyz = np.stack([cols["x"], cols["y"], cols["z"]], axis=1).astype(np.float64)
    rgb = None
    if want_color and all(c in cols for c in ("red", "green", "blue")):
        rgb = np.stack([cols["red"], cols["green"], cols["blue"]], axis=1).astype(np.float64)
        if rgb.max() > 1.0:
            rgb = rgb / 255.0
    return xyz, rgb


def voxel_down_sample_colored(xyz, colors, voxel_size, device=None):
    """open3d's PointCloud.voxel_down_sample of a coloured cloud (points AND colours are averaged per
    voxel) on the HIP library (csrc/voxel.hip).  numpy in, numpy float32 out.  GPU only: there is no
    CPU fallback."""
    import torch
    from .. import reg_xyz
    dev = torch.device("cuda" if device is None else device)
    x = torch.as_tensor(np.ascontiguousarray(xyz, np.float32), device=dev)
    if colors is None:
        return reg_xyz.voxel_down_sample(x, voxel_size).cpu().numpy(), None
    c = torch.as_tensor(np.ascontiguousarray(colors, np.float32), device=dev)
    ox, oc = reg_xyz.voxel_down_sample(x, voxel_size, colors=c)
    return ox.cpu().numpy(), oc.cpu().numpy()


def load_xyz(path, down_sample=None, device=None):
    """utils/dataUtils.py:174-189, same signature and returns: (points float32 [N,3], colours float32
    [N,3]).  down_sample: voxel size of an open3d-style voxel_down_sample applied to points and
    colours (runs on the GPU).  A PLY without colours -- or with all-zero ones -- gets colours derived
    from the position inside the bounding box (:185-187), so the colours are never None."""
    xyz, rgb = read_ply(path)
    points = xyz.astype(np.float32)
    colors = None if rgb is None else rgb.astype(np.float32)
    if down_sample:
        points, colors = voxel_down_sample_colored(xyz, rgb, down_sample, device)
    has_valid_color = colors is not None and not np.allclose(colors, 0)
    if not has_valid_color:
        colors = (points - points.min(axis=0)) / (points.max(axis=0) - points.min(axis=0) + 1e-8)
        colors = np.clip(colors, 0, 1)
    return points, colors


def save_ply_xyzrgb(xyz, rgb, path):
    """utils/dataUtils.py `save_ply_xyzrgb`: binary little-endian, double xyz + uchar rgb
    (rgb in [0,1] or [0,255]; None writes xyz only)."""
    xyz = np.asarray(xyz, np.float64)
    n = xyz.shape[0]
    props = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    lines = ["ply", "format binary_little_endian 1.0", "comment Created by genpc_amd", "element vertex %d" % n,
             "property double x", "property double y", "property double z"]
    if rgb is not None:
        rgb = np.asarray(rgb, np.float64)
        if rgb.max() <= 1.0:
            rgb = rgb * 255.0
        props += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        lines += ["property uchar red", "property uchar green", "property uchar blue"]
    lines.append("end_header")
    data = np.empty(n, dtype=np.dtype(props))
    data["x"], data["y"], data["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if rgb is not None:
        c = np.clip(np.rint(rgb), 0, 255).astype(np.uint8)
        data["red"], data["green"], data["blue"] = c[:, 0], c[:, 1], c[:, 2]
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        f.write(data.tobytes())


def normalize_numpy(xyz, range=1.0):
    """utils/dataUtils.py:561-581."""
    vmin, vmax = xyz.min(axis=0), xyz.max(axis=0)
    center = (vmax + vmin) / 2.0
    scale_factor = (vmax - vmin).max()
    return (xyz - center) / scale_factor * (range / 0.5), center, scale_factor


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


def getCategory(flag):
    """utils/dataUtils.py:601-614 -- the text prompt category of a bundled scan id; unknown ids (the reference raises
    KeyError) fall back to 'object'."""
    kv = {"01184": "Wheelie Bin", "05117": "chair", "05452": "armchair", "06127": "Plant vases", "06145": "table",
          "06188": "vespa", "06830": "Kid tricycle", "07136": "sofa", "07306": "trash can", "09639": "swivel chair"}
    return kv.get(str(flag), "object")
