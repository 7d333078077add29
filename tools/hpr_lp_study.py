"""CPU study (numpy, a few views): what the points that reach hpr_overflow_kernel (undecided after the home tiles and the
centroid trial) end up as, and how many "most violated constraint" iterations would find a strictly feasible normal
for the visible ones.  Not a test, not the oracle: approximate direction order, same arithmetic otherwise.

    python tools/hpr_lp_study.py [blob|scan] [nviews]
"""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
BOX = 1.0e4
TILE = 128


def clip(poly, A, B, C):
    s = poly[:, 0] * A + poly[:, 1] * B - C
    out = s > 0
    if not out.any():
        return poly, False
    if out.all():
        return poly[:0], True
    n = len(poly)
    res = []
    for k in range(n):
        k2 = (k + 1) % n
        if not out[k]:
            res.append(poly[k])
        if out[k] != out[k2] and s[k] != 0 and s[k2] != 0:
            t = s[k] / (s[k] - s[k2])
            res.append(poly[k] + t * (poly[k2] - poly[k]))
    return np.array(res).reshape(-1, 2), True


def frame(p):
    rho = np.linalg.norm(p)
    u = p / rho
    a = np.abs(u)
    if a[0] <= a[1] and a[0] <= a[2]:
        e1 = np.array([0.0, u[2], -u[1]])
    elif a[1] <= a[2]:
        e1 = np.array([-u[2], 0.0, u[0]])
    else:
        e1 = np.array([u[1], -u[0], 0.0])
    e1 /= np.linalg.norm(e1)
    return rho, u, e1, np.cross(u, e1)


def morton_order(dirs, axis):
    # 2-D Morton order of the directions in a frame whose third axis points at the cloud
    w = axis / np.linalg.norm(axis)
    t = np.array([1.0, 0, 0]) if abs(w[0]) < 0.9 else np.array([0, 1.0, 0])
    a = np.cross(w, t); a /= np.linalg.norm(a)
    b = np.cross(w, a)
    x = dirs @ a / np.maximum(dirs @ w, 1e-9)
    y = dirs @ b / np.maximum(dirs @ w, 1e-9)
    def q(v):
        v = (v - v.min()) / (v.max() - v.min() + 1e-30)
        return np.minimum((v * 1024).astype(np.int64), 1023)
    def spread(v):
        r = np.zeros_like(v)
        for i in range(10):
            r |= ((v >> i) & 1) << (2 * i)
        return r
    return np.argsort(spread(q(x)) | (spread(q(y)) << 1), kind="stable")


def study(pts, eye, radius=1.0e4, home_tiles=3):
    n = len(pts)
    v = pts.astype(np.float64) - eye
    r = np.linalg.norm(v, axis=1)
    fl = v + (2.0 * (radius - r) / r)[:, None] * v
    order = morton_order(fl / np.linalg.norm(fl, axis=1)[:, None], -eye)
    fl = fl[order]
    rho = np.linalg.norm(fl, axis=1)
    u = fl / rho[:, None]
    # accept: u.q < rho for all others (margin 1e-8 rho)
    hard = np.zeros(n, bool)
    for s in range(0, n, 1000):
        d = rho[s:s + 1000, None] - u[s:s + 1000] @ fl.T
        d[np.arange(min(1000, n - s)), np.arange(s, min(s + 1000, n))] = np.inf
        hard[s:s + 1000] = (d < 1e-8 * rho[s:s + 1000, None]).any(1)
    stats = dict(n=n, hard=int(hard.sum()), died_home=0, verified=0, surv=0, surv_visible=0, surv_hidden=0)
    lp_iters, lp_fail, hid_clips = [], 0, []
    ntiles = (n + TILE - 1) // TILE
    for i in np.nonzero(hard)[0]:
        rho_i, ui, e1, e2 = frame(fl[i])
        A_all = fl @ e1; B_all = fl @ e2; C_all = rho_i - fl @ ui
        A_all[i] = 0; B_all[i] = 0; C_all[i] = 1.0
        poly = np.array([[-BOX, -BOX], [BOX, -BOX], [BOX, BOX], [-BOX, BOX]])
        home = i // TILE
        alive = True
        for tile in (home, home + 1, home - 1)[:home_tiles]:
            if tile < 0 or tile >= ntiles or not alive:
                continue
            js = np.arange(tile * TILE, min(n, tile * TILE + TILE))
            if tile == home:
                rot = (i - home * TILE) >> 5
                js = np.concatenate([js[(js - home * TILE) >> 5 == ((c + rot) & 3)] for c in range(4)])
            for j in js:
                poly, cut = clip(poly, A_all[j], B_all[j], C_all[j])
                if cut and len(poly) < 3:
                    alive = False
                    break
        if not alive:
            stats["died_home"] += 1
            continue
        ctr = poly.mean(0)
        if ctr @ ctr >= 1e6:
            r2 = (poly ** 2).sum(1)
            k = np.argsort(r2)[:2]
            v0, v1 = poly[k[0]], poly[k[1]]
            d = ctr - v0; ln = np.linalg.norm(d)
            h = min(np.linalg.norm(v1 - v0), 0.5 * ln)
            ctr = v0 + h * d / ln if ln > 0 and h > 0 else ctr
        sv = ctr[0] * A_all + ctr[1] * B_all - C_all
        if (sv < -1e-10 * rho_i * (1 + ctr @ ctr)).all():
            stats["verified"] += 1
            continue
        stats["surv"] += 1
        # exact decision: clip by everything else
        p2 = poly.copy(); nc = 0
        cand = np.nonzero(np.ones(n, bool))[0]
        for j in cand:
            if ((p2[:, 0] * A_all[j] + p2[:, 1] * B_all[j] - C_all[j]) > 0).any():
                p2, _ = clip(p2, A_all[j], B_all[j], C_all[j]); nc += 1
                if len(p2) < 3:
                    break
        vis = len(p2) >= 3
        stats["surv_visible" if vis else "surv_hidden"] += 1
        if not vis:
            hid_clips.append(nc)
        # LP: clip a scratch polygon by the most violated constraint of its interior point until none is violated
        for mode in ("raw", "norm"):
            p3 = poly.copy(); it = 0; res = "fallback"
            while it < 6:
                c = p3.mean(0)
                if c @ c >= 1e6:
                    r2 = (p3 ** 2).sum(1); k = np.argsort(r2)[:2]; v0, v1 = p3[k[0]], p3[k[1]]
                    d = c - v0; ln = np.linalg.norm(d); h = min(np.linalg.norm(v1 - v0), 0.5 * ln)
                    if ln > 0 and h > 0:
                        c = v0 + h * d / ln
                sv = c[0] * A_all + c[1] * B_all - C_all
                thr = 1e-10 * rho_i * (1 + c @ c)
                if (sv < -thr).all():
                    res = "visible"
                    break
                j = int(np.argmax(sv if mode == "raw" else sv / np.sqrt(A_all ** 2 + B_all ** 2 + 1e-300)))
                sj = p3[:, 0] * A_all[j] + p3[:, 1] * B_all[j] - C_all[j]
                mg = 1e-9 * (np.abs(p3[:, 0] * A_all[j]) + np.abs(p3[:, 1] * B_all[j]) + abs(C_all[j]))
                it += 1
                if (sj > mg).all():
                    res = "hidden"
                    break
                p3, _ = clip(p3, A_all[j], B_all[j], C_all[j])
                if len(p3) < 3:
                    break
            assert not (res == "visible" and not vis) and not (res == "hidden" and vis)
            key = mode + "_" + ("V" if vis else "H") + "_" + res
            stats[key] = stats.get(key, 0) + 1
            stats[key + "_iters"] = stats.get(key + "_iters", 0) + it
    stats["hidden_clips_mean"] = float(np.mean(hid_clips)) if hid_clips else 0
    return stats


lp_iters_h = []
if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "blob"
    nviews = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    if which == "blob":
        vv = rng.normal(size=(165546, 3)); vv /= np.linalg.norm(vv, axis=1, keepdims=True)
        pts = (vv * (0.3 + 0.2 * np.abs(np.sin(3 * vv[:, :1])))).astype(np.float32)
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
        pts = g["partial"][0].astype(np.float32)
    sub = pts[O.fps(pts, 10000)] if len(pts) > 10000 else pts
    for k in range(nviews):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        c = (sub.max(0) + sub.min(0)) / 2
        for ht in (3, 1):
            print(which, "view", k, "home tiles", ht, study(sub, c + 1.6 * d, home_tiles=ht), flush=True)
