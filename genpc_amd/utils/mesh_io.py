"""GLB mesh loading and surface sampling without trimesh: what
``glb2point`` (utils/dataUtils.py:217-250) needs to turn a generated ``*.glb`` into
the "complete" cloud that reg() aligns.  Host-side format code, numpy only.

``load_glb``   binary glTF 2.0: JSON chunk + BIN chunk, every triangle primitive of
               every mesh reachable from the default scene, node transforms applied
               (matrix or translation / rotation / scale), concatenated like
               ``trimesh.Scene.dump(concatenate=True)``.  Vertex colours: COLOR_0 when
               present; otherwise the material's base-colour TEXTURE baked to the vertices
               the way ``TextureVisuals.to_color()`` does it (utils/dataUtils.py:224-225;
               trimesh's published ``uv_to_color``: nearest texel at x = u (W-1),
               y = v_gltf (H-1), rounded, wrapped like GL_REPEAT -- trimesh is absent and
               unpinned); otherwise the material's baseColorFactor.  The image (PNG / JPEG
               in a bufferView or a data URI) is decoded with PIL.
``sample_surface``  area-weighted uniform sampling of a triangle mesh, the published
               algorithm of ``trimesh.sample.sample_surface``: faces drawn with
               probability proportional to area, points by the folded-parallelogram
               trick.  trimesh draws from numpy's unseeded global RNG, so the
               reference's samples are not reproducible; here a Generator is passed.
"""
import base64
import io
import json
import struct

import numpy as np

_COMPONENT = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _accessor(gltf, bin_chunk, index):
    acc = gltf["accessors"][index]
    view = gltf["bufferViews"][acc["bufferView"]]
    dtype, size = _COMPONENT[acc["componentType"]]
    ncomp = _NCOMP[acc["type"]]
    offset = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or size * ncomp
    count = acc["count"]
    if stride == size * ncomp:
        arr = np.frombuffer(bin_chunk, dtype=dtype, count=count * ncomp, offset=offset).reshape(count, ncomp)
    else:
        raw = np.frombuffer(bin_chunk, dtype="u1", count=(count - 1) * stride + size * ncomp, offset=offset)
        idx = np.arange(count)[:, None] * stride + np.arange(size * ncomp)[None, :]
        arr = raw[idx].copy().view(dtype).reshape(count, ncomp)
    arr = arr.astype(np.float64) if dtype == "<f4" else arr
    if acc.get("normalized") and dtype != "<f4":
        arr = arr.astype(np.float64) / np.iinfo(np.dtype(dtype)).max
    return arr


def _node_matrix(node):
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T        # glTF stores column-major
    M = np.eye(4)
    if "scale" in node:
        M = np.diag(list(node["scale"]) + [1.0]) @ M
    if "rotation" in node:
        x, y, z, w = node["rotation"]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = np.eye(4)
        T[:3, :3] = R
        M = T @ M
    if "translation" in node:
        T = np.eye(4)
        T[:3, 3] = node["translation"]
        M = T @ M
    return M


def _texture_image(gltf, bin_chunk, tex_index, cache):
    """RGBA uint8 array [H,W,4] of texture `tex_index` (None if it cannot be resolved)."""
    if tex_index in cache:
        return cache[tex_index]
    img = None
    try:
        source = gltf["textures"][tex_index]["source"]
        rec = gltf["images"][source]
        if "bufferView" in rec:
            view = gltf["bufferViews"][rec["bufferView"]]
            raw = bytes(bin_chunk[view.get("byteOffset", 0): view.get("byteOffset", 0) + view["byteLength"]])
        elif str(rec.get("uri", "")).startswith("data:"):
            raw = base64.b64decode(rec["uri"].split(",", 1)[1])
        else:
            raw = None                                          # external file: not followed
        if raw is not None:
            from PIL import Image
            img = np.asarray(Image.open(io.BytesIO(raw)).convert("RGBA"))
    except (KeyError, IndexError, ValueError, OSError):
        img = None
    cache[tex_index] = img
    return img


def uv_to_color(uv, image):
    """trimesh.visual.color.uv_to_color with the glTF loader's v-flip folded in: uv are the file's
    TEXCOORD values, image an [H,W,4] uint8 array -> uint8 [N,4]."""
    h, w = image.shape[:2]
    x = np.round(uv[:, 0] * (w - 1)).astype(np.int64) % w
    y = np.round(uv[:, 1] * (h - 1)).astype(np.int64) % h
    return image[y, x]


def _material_colors(gltf, bin_chunk, prim, nverts, cache):
    """Per-vertex colours in [0,1] from the primitive's material, or None."""
    if "material" not in prim:
        return None
    pbr = gltf["materials"][prim["material"]].get("pbrMetallicRoughness", {})
    tex = pbr.get("baseColorTexture")
    if tex is not None:
        attr = "TEXCOORD_%d" % tex.get("texCoord", 0)
        img = _texture_image(gltf, bin_chunk, tex["index"], cache)
        if img is not None and attr in prim["attributes"]:
            uv = np.asarray(_accessor(gltf, bin_chunk, prim["attributes"][attr]), np.float64)[:, :2]
            return uv_to_color(uv, img)[:, :3].astype(np.float64) / 255.0
    if "baseColorFactor" in pbr:
        # (trimesh stores the factor as uint8 RGBA)
        f8 = np.round(np.clip(np.asarray(pbr["baseColorFactor"], np.float64)[:3], 0, 1) * 255.0)
        return np.tile(f8 / 255.0, (nverts, 1))
    return None


def load_glb(path):
    """-> (vertices float64 [N,3], faces int64 [M,3], colors float64 [N,3] in [0,1] or None)."""
    with open(path, "rb") as f:
        data = f.read()
    magic, version, length = struct.unpack_from("<4sII", data, 0)
    if magic != b"glTF" or version != 2:
        raise ValueError("%s: not a binary glTF 2.0 file" % path)
    off, gltf, bin_chunk = 12, None, b""
    while off < length:
        clen, ctype = struct.unpack_from("<I4s", data, off)
        chunk = data[off + 8: off + 8 + clen]
        if ctype == b"JSON":
            gltf = json.loads(chunk.decode("utf-8"))
        elif ctype == b"BIN\x00":
            bin_chunk = chunk
        off += 8 + clen
    if gltf is None:
        raise ValueError("%s: no JSON chunk" % path)
    verts, faces, cols, base = [], [], [], 0
    any_color = False
    tex_cache = {}

    def visit(ni, parent):
        nonlocal base, any_color
        node = gltf["nodes"][ni]
        M = parent @ _node_matrix(node)
        if "mesh" in node:
            for prim in gltf["meshes"][node["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue                                   # triangles only
                v = _accessor(gltf, bin_chunk, prim["attributes"]["POSITION"])[:, :3]
                v = v @ M[:3, :3].T + M[:3, 3]
                if "indices" in prim:
                    fi = _accessor(gltf, bin_chunk, prim["indices"]).astype(np.int64).reshape(-1, 3)
                else:
                    fi = np.arange(len(v), dtype=np.int64).reshape(-1, 3)
                if "COLOR_0" in prim["attributes"]:
                    c = _accessor(gltf, bin_chunk, prim["attributes"]["COLOR_0"])[:, :3].astype(np.float64)
                    any_color = True
                else:
                    c = _material_colors(gltf, bin_chunk, prim, len(v), tex_cache)
                    if c is None:
                        c = np.full((len(v), 3), np.nan)
                    else:
                        any_color = True
                verts.append(v)
                faces.append(fi + base)
                cols.append(c)
                base += len(v)
        for child in node.get("children", []):
            visit(child, M)

    scenes = gltf.get("scenes")
    roots = scenes[gltf.get("scene", 0)]["nodes"] if scenes else range(len(gltf.get("nodes", [])))
    for r in roots:
        visit(r, np.eye(4))
    if not verts:
        raise ValueError("%s: no triangle geometry" % path)
    V, F = np.concatenate(verts), np.concatenate(faces)
    C = np.nan_to_num(np.concatenate(cols), nan=0.5) if any_color else None
    return V, F, C


def sample_surface(vertices, faces, count, rng=None):
    """-> (points [count,3], face_index [count]); area-weighted, uniform within a face."""
    rng = np.random.default_rng() if rng is None else rng
    tri = vertices[faces]                                       # [M,3,3]
    origin, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    cum = np.cumsum(area)
    face_index = np.searchsorted(cum, rng.random(count) * cum[-1])
    face_index = np.minimum(face_index, len(faces) - 1)
    r = rng.random((count, 2))
    fold = r.sum(axis=1) > 1.0
    r[fold] = 1.0 - r[fold]
    pts = origin[face_index] + e1[face_index] * r[:, :1] + e2[face_index] * r[:, 1:]
    return pts, face_index


def glb2point(glb_path, down_sample=None, num_points=16384, rng=None, device=None):
    """utils/dataUtils.py:217-250 without open3d/trimesh, same leading arguments: (points [n,3],
    colours [n,3]); colours are barycentric blends of the vertex colours (COLOR_0, or the baked
    base-colour texture / factor), 0.5 grey when the file has none.  down_sample: voxel size of the
    open3d-style voxel_down_sample applied to points and colours afterwards (:248-249; on the GPU).
    The reference samples from trimesh's unseeded global RNG; pass `rng` for a reproducible cloud."""
    pts, col = _glb2point_full(glb_path, num_points, rng)
    if down_sample:
        from .dataUtils import voxel_down_sample_colored
        pts, col = voxel_down_sample_colored(pts, col, down_sample, device)
    return pts, col


def _glb2point_full(glb_path, num_points, rng):
    V, F, C = load_glb(glb_path)
    pts, fi = sample_surface(V, F, num_points, rng)
    if C is None:
        return pts, np.full((num_points, 3), 0.5)
    tri = V[F[fi]]
    v0, v1, v2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], pts - tri[:, 0]
    d00, d01, d11 = (v0 * v0).sum(1), (v0 * v1).sum(1), (v1 * v1).sum(1)
    d20, d21 = (v2 * v0).sum(1), (v2 * v1).sum(1)
    den = d00 * d11 - d01 * d01
    b1 = (d11 * d20 - d01 * d21) / den
    b2 = (d00 * d21 - d01 * d20) / den
    bary = np.stack([1 - b1 - b2, b1, b2], axis=1)
    col = (C[F[fi]] * bary[:, :, None]).sum(axis=1)
    return pts, np.clip(col, 0, 1)
