"""Golden vectors produced by the REFERENCE'S OWN Python code, executed in the build container.

    python tests/golden/make_reference_vectors.py          # needs /root/reference (read-only)

The reference's Python modules cannot be imported whole here: their top-level imports pull in
pytorch3d, open3d, kaolin, cv2, torchvision, trimesh, fpsample ... (all absent from the image, SURVEY.md
section 8c).  But the functions on the hot path that are pure torch / numpy never touch those
libraries.  This script therefore performs a SELECTIVE import: it parses the reference source with
``ast``, takes the definitions named below -- unmodified, with their original line numbers -- and
executes exactly those in a namespace that holds only real torch / numpy / math.  No stand-in for any
absent library is written: a function that needs one is simply not taken.  The reference's text is
never copied into the repository; only the arrays it computes are stored, under tests/golden/ref_py_*.npz,
each with the inputs that produced it.  tests/test_reference_vectors.py checks the CPU oracle (and, with
``-m gpu``, the HIP library) against them, which pins these rows to the reference itself:

  ref_py_mask_loss.npz   optim_registration/diff_obj_pose.py  compute_loss_function (with normalize_images,
                         compute_soft_mask, dice_loss, soft_iou_loss, :166-336) on float32 [S,S,3] images:
                         total / mse / mask loss and d total / d result through torch autograd   (row a16)
  ref_py_paint.npz       DepthPrompting.paintPixels / getRawDepth (:292-391) on CPU tensors      (row a14)
  ref_py_uvs.npz         DepthPrompting.getUvs (:239-271): the per-camera bounding-box rescale of
                         transformed points (the kaolin camera transform itself is absent: the
                         transformed points are an input of the vector)                          (row a13)
  ref_py_loss_util.npz   utils/loss_util.py Completionloss.chamfer_l1 / chamfer_l2 / chamfer_partial_l1 /
                         chamfer_partial_l2 / emd_loss (:25-49): the five reductions on given
                         distance arrays (the CUDA extensions behind chamfer_dist / EMD are absent:
                         their outputs are an input of the vector)                               (row a12)
  ref_py_utils.npz       utils/camera_utils.py fibonacci_sphere, calculate_up_vector (:86-113);
                         utils/dataUtils.py get_rotate_matrix, normalize_numpy (:455-472,561-581);
                         diff_obj_pose.py build_transform (:464-468)                             (rows a13, a16, a17)

The inputs are seeded; the oracle is used only to draw realistic test IMAGES for the mask-loss vectors
(any float32 image would do -- the expected outputs come from the reference code alone).
"""
import ast
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GENPC_REFERENCE", "/root/reference")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def take(relpath, names, namespace):
    """Execute the definitions `names` of reference file `relpath` ("func" or "Class.method") in `namespace`."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path, encoding="utf-8").read(), filename=path)
    want = set(names)
    picked = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            picked.append(node)
            want.discard(node.name)
        elif isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and "%s.%s" % (node.name, sub.name) in want:
                    picked.append(sub)
                    want.discard("%s.%s" % (node.name, sub.name))
    if want:
        raise KeyError("not found in %s: %s" % (relpath, sorted(want)))
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), namespace)
    return namespace


def mask_loss_vectors():
    from oracle import oracle as O
    ns = {"torch": torch, "F": torch.nn.functional, "np": np, "math": math}
    take("optim_registration/diff_obj_pose.py",
         ["compute_mask_from_rendering", "soft_iou_loss", "normalize_images", "dice_loss", "compute_soft_mask",
          "edge_loss", "compute_loss_function", "build_transform"], ns)
    rng = np.random.default_rng(20260301)
    out = {}
    names = []

    def cloud(n, spread, dark_frac=0.0, white=False):
        pts = ((rng.random((n, 3)) - 0.5) * spread).astype(np.float32)
        col = np.ones((n, 3), np.float32) if white else (0.25 + 0.75 * rng.random((n, 3))).astype(np.float32)
        k = int(n * dark_frac)
        if k:
            col[:k] *= np.float32(0.08)
        return pts, col

    cases = [
        # name, S, (n_ref, spread_ref, radius_ref), (n_res, spread_res, radius_res), dark_frac, white
        ("colour48", 48, (500, 0.8, 0.03), (300, 0.9, 0.033), 0.0, False),
        ("white40", 40, (400, 0.9, 0.04), (60, 0.9, 0.025), 0.0, True),
        ("darkhalf48", 48, (500, 0.8, 0.03), (400, 0.8, 0.033), 0.5, False),
        ("saturating36", 36, (40, 0.3, 0.03), (900, 1.0, 0.05), 0.0, False),
        ("dense64", 64, (3000, 0.7, 0.02), (2500, 0.75, 0.022), 0.2, False),
    ]
    for name, S, (nr, sr, rr), (nq, sq, rq), dark, white in cases:
        pr, cr = cloud(nr, sr, 0.0, white)
        pq, cq = cloud(nq, sq, dark, white)
        ref = O.splat_image(pr, rr, S, cr)
        res = O.splat_image(pq, rq, S, cq)
        tr = torch.from_numpy(ref)
        tq = torch.from_numpy(res).requires_grad_(True)
        total, mse, edge, iou_l, iou_v, cd, mask = ns["compute_loss_function"](tr, tq, None, None, None, None)
        total.backward()
        out[name + "_ref"] = ref
        out[name + "_result"] = res
        out[name + "_total"] = np.float32(total.item())
        out[name + "_mse"] = np.float32(mse.item())
        out[name + "_mask"] = np.float32(mask.item())
        out[name + "_grad"] = tq.grad.numpy().copy()
        # the reference's hard mask of the reference image (render_reference_image's second output, :132)
        out[name + "_hardmask"] = ns["compute_mask_from_rendering"](tr).numpy()
        names.append(name)
        print("  mask_loss %-13s S=%d total=%.6f mask=%.6f |grad|max=%.3e" % (name, S, total.item(), mask.item(),
                                                                              float(tq.grad.abs().max())))
    out["cases"] = np.array(names)
    # build_transform (:464-468)
    R = torch.tensor([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]])
    out["build_transform"] = ns["build_transform"](R, torch.tensor([0.1, -0.2, 0.3]), torch.tensor(0.8)).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_py_mask_loss.npz"), **out)


def paint_vectors():
    ns = {"torch": torch, "np": np}
    take("DepthPrompting.py", ["DepthPrompting.paintPixels", "DepthPrompting.getRawDepth", "DepthPrompting.getUvs"], ns)
    rng = np.random.default_rng(20260302)
    out = {}
    names = []
    for name, res, n, point_size, rate in (("p1", 64, 700, 1, 3), ("p2", 64, 400, 2, 3), ("p3", 96, 900, 3, 2),
                                           ("p1edge", 32, 300, 1, 3), ("p2edge", 32, 200, 2, 3)):
        me = types.SimpleNamespace(device="cpu", cfg=types.SimpleNamespace(res=res))
        me.paintPixels = types.MethodType(ns["paintPixels"], me)
        lo, hi = (0, res) if "edge" in name else (4, res - 4)
        pix = rng.integers(lo, hi, size=(n, 2)).astype(np.int64)
        pix[n // 2:n // 2 + n // 10] = pix[:n // 10]                      # duplicates: the last writer wins
        colors = rng.random((n, 3)).astype(np.float32)
        depth = (rng.random(n) * 3 + 0.5).astype(np.float32)
        s_img, s_dep, h1, h2 = ns["getRawDepth"](me, torch.from_numpy(pix), torch.from_numpy(depth), "redwood",
                                                 colors=torch.from_numpy(colors), res=res, point_size=point_size,
                                                 mask_pixel_rate=rate)
        out[name + "_pix"] = pix.astype(np.int32)
        out[name + "_colors"] = colors
        out[name + "_depth"] = depth
        out[name + "_params"] = np.array([res, point_size, rate], np.int32)
        out[name + "_sparse_img"] = s_img.numpy()
        out[name + "_sparse_depth"] = s_dep.numpy()
        out[name + "_hole_mask1"] = h1.numpy()
        out[name + "_hole_mask2"] = h2.numpy()
        names.append(name)
        print("  paint %-7s res=%d n=%d point_size=%d painted=%d" % (name, res, n, point_size, int((s_img != 0).sum())))
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ref_py_paint.npz"), **out)

    # getUvs: cams are duck-typed -- cam.transform(points) returns a row of the stored array
    out = {}
    names = []
    for name, c, n, rescale, padding in (("r1", 5, 400, True, 0.15), ("r2", 3, 257, True, 0.05), ("n1", 4, 100, False, 0.15)):
        tp = (rng.standard_normal((c, n, 3)) * np.array([0.4, 0.3, 0.05]) + np.array([0.02, -0.03, 0.97])).astype(np.float32)
        me = types.SimpleNamespace(device="cpu")

        class Cam:
            def __init__(self, row):
                self.row = row

            def transform(self, points):
                return torch.from_numpy(self.row)
        pts = torch.zeros(n, 3)
        uv, dp, tr = ns["getUvs"](me, [Cam(tp[i]) for i in range(c)], pts, rescale=rescale, padding=padding)
        out[name + "_transformed"] = tp
        out[name + "_params"] = np.array([float(rescale), padding], np.float64)
        out[name + "_uv"] = uv.numpy()
        out[name + "_depth"] = dp.numpy()
        names.append(name)
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ref_py_uvs.npz"), **out)


def loss_util_vectors():
    ns = {"torch": torch}
    take("utils/loss_util.py", ["Completionloss.chamfer_l1", "Completionloss.chamfer_l2", "Completionloss.chamfer_partial_l1",
                                "Completionloss.chamfer_partial_l2", "Completionloss.emd_loss"], ns)
    rng = np.random.default_rng(20260303)
    out = {}
    names = []
    for name, b, n, m in (("b1", 1, 2048, 2048), ("b3", 3, 1000, 777), ("b13", 13, 512, 512)):
        d1 = (rng.random((b, n)) ** 2 * 0.01).astype(np.float32)
        d2 = (rng.random((b, m)) ** 2 * 0.02).astype(np.float32)
        de = (rng.random((b, n)) ** 2 * 0.03).astype(np.float32)
        me = types.SimpleNamespace(
            chamfer_dist=lambda p1, p2, d1=d1, d2=d2: (torch.from_numpy(d1), torch.from_numpy(d2), None, None),
            EMD=lambda p1, p2, eps, iters, de=de: (torch.from_numpy(de), None))
        out[name + "_d1"], out[name + "_d2"], out[name + "_demd"] = d1, d2, de
        for fn in ("chamfer_l1", "chamfer_l2", "chamfer_partial_l1", "chamfer_partial_l2", "emd_loss"):
            out[name + "_" + fn] = np.float32(ns[fn](me, None, None).item())
        names.append(name)
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ref_py_loss_util.npz"), **out)


def utils_vectors():
    ns = {"np": np, "math": math}
    take("utils/camera_utils.py", ["fibonacci_sphere", "calculate_up_vector"], ns)
    take("utils/dataUtils.py", ["get_rotate_matrix", "normalize_numpy"], ns)
    rng = np.random.default_rng(20260304)
    out = {}
    out["fib_1024_1p6"] = ns["fibonacci_sphere"](1024, 1.6)
    out["fib_7_2"] = ns["fibonacci_sphere"](7, 2.0)
    eyes = np.concatenate([out["fib_7_2"], np.array([[0.0, 1.6, 0.0], [0.0, -1.6, 0.0], [1e-12, 1.0, 0.0]])])
    out["up_eyes"] = eyes
    out["up_vectors"] = np.stack([ns["calculate_up_vector"](e.copy(), np.zeros(3)).astype(np.float64) for e in eyes])
    for ax in "xyz":
        for ang in (90, 37.5, -120):
            out["rot_%s_%s" % (ax, str(ang).replace(".", "p").replace("-", "m"))] = ns["get_rotate_matrix"](ax, ang)
    xyz = (rng.standard_normal((500, 3)) * np.array([1.0, 0.3, 2.0]) + np.array([0.5, -1.0, 3.0]))
    for r in (0.5, 1.0):
        nx, c, s = ns["normalize_numpy"](xyz.copy(), range=r)
        out["norm_in"] = xyz
        out["norm_out_%s" % str(r).replace(".", "p")] = nx
        out["norm_center"] = c
        out["norm_scale"] = np.float64(s)
    np.savez_compressed(os.path.join(HERE, "ref_py_utils.npz"), **out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not found at %s (this script only runs in the build container)" % REF)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    mask_loss_vectors()
    paint_vectors()
    loss_util_vectors()
    utils_vectors()
    for f in sorted(os.listdir(HERE)):
        if f.startswith("ref_py_"):
            print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))
