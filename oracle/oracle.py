"""ctypes front-end of the CPU oracle (oracle/genpc_oracle*.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, by bench.py's
``cpu_baseline`` leg and by ``__graft_entry__.smoke()`` -- never by genpc_amd/.
Parity status: see the header of genpc_oracle.c ("parity unpinned" by the
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
def fps(xyz, k):
    """Deterministic farthest point sampling (start 0, first arg-max) -> idx[k]."""
    xyz, p = _f(xyz)
    out = np.zeros(k, np.int32)
    lib().oracle_fps(xyz.shape[0], p, int(k), out.ctypes.data_as(_i32p))
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
