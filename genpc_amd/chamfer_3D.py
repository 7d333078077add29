"""Drop-in for the reference's pybind module ``chamfer_3D``
(loss_functions/Chamfer3D/chamfer_cuda.cpp:30-33): same two functions, same
argument order, same caller-allocates contract, same return codes -- but the
work is done by libgenpc_hip.so through its C ABI, on torch's current stream.
Unlike the reference (dist_chamfer_3D.py:45 ignores the code) callers in this
package raise on a non-1 return.
"""
from . import _lib

_L = _lib.lib
_p = _lib.ptr


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_cuda.cpp:17-19 -> chamfer3D.cu:136-154."""
    _lib.check_tensors((("xyz1", xyz1), ("xyz2", xyz2), ("dist1", dist1), ("dist2", dist2)),
                       (("idx1", idx1), ("idx2", idx2)))
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    return _lib.on_device_of(xyz1, _L.genpc_chamfer_forward, b, n, _p(xyz1), m, _p(xyz2), _p(dist1), _p(idx1),
                             _p(dist2), _p(idx2))


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """chamfer_cuda.cpp:22-26 -> chamfer3D.cu:176-195."""
    _lib.check_tensors((("xyz1", xyz1), ("xyz2", xyz2), ("gradxyz1", gradxyz1), ("gradxyz2", gradxyz2),
                        ("graddist1", graddist1), ("graddist2", graddist2)), (("idx1", idx1), ("idx2", idx2)))
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    return _lib.on_device_of(xyz1, _L.genpc_chamfer_backward, b, n, _p(xyz1), m, _p(xyz2), _p(graddist1), _p(idx1),
                             _p(graddist2), _p(idx2), _p(gradxyz1), _p(gradxyz2))


def nm_distance(xyz, xyz2, result, result_i):
    """One direction only (NmDistanceKernel, chamfer3D.cu:12-134)."""
    _lib.check_tensors((("xyz", xyz), ("xyz2", xyz2), ("result", result)), (("result_i", result_i),))
    b, n, _ = xyz.shape
    m = xyz2.shape[1]
    return _lib.on_device_of(xyz, _L.genpc_nm_distance, b, n, _p(xyz), m, _p(xyz2), _p(result), _p(result_i))


def nm_distance_within(xyz, xyz2, radius2, result, result_i):
    """nm_distance with a search limit (squared): queries without a target within it get
    (+inf, -1); the others exactly what nm_distance returns (reg_xyz.py:41-52)."""
    _lib.check_tensors((("xyz", xyz), ("xyz2", xyz2), ("result", result)), (("result_i", result_i),))
    b, n, _ = xyz.shape
    m = xyz2.shape[1]
    return _lib.on_device_of(xyz, _L.genpc_nm_distance_within, b, n, _p(xyz), m, _p(xyz2), float(radius2), _p(result),
                             _p(result_i))
