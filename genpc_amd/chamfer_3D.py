"""Drop-in for the reference's pybind module ``chamfer_3D``
(loss_functions/Chamfer3D/chamfer_cuda.cpp:30-33): same two functions, same
argument order, same caller-allocates contract, same return codes -- but the
work is done by libgenpc_hip.so through its C ABI, on torch's current stream.
Unlike the reference (dist_chamfer_3D.py:45 ignores the code) callers in this
package raise on a non-1 return.
"""
import torch

from . import _lib


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_cuda.cpp:17-19 -> chamfer3D.cu:136-154."""
    _lib.require_gpu(xyz1, xyz2, dist1, dist2, idx1, idx2)
    _lib.require(xyz1, torch.float32, "xyz1")
    _lib.require(xyz2, torch.float32, "xyz2")
    _lib.require(dist1, torch.float32, "dist1")
    _lib.require(dist2, torch.float32, "dist2")
    _lib.require(idx1, torch.int32, "idx1")
    _lib.require(idx2, torch.int32, "idx2")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    with torch.cuda.device(xyz1.device):
        return _lib.lib.genpc_chamfer_forward(
            b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(idx1),
            _lib.ptr(dist2), _lib.ptr(idx2), _lib.stream_of(xyz1))


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """chamfer_cuda.cpp:22-26 -> chamfer3D.cu:176-195."""
    _lib.require_gpu(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
    for t, nme in ((xyz1, "xyz1"), (xyz2, "xyz2"), (gradxyz1, "gradxyz1"), (gradxyz2, "gradxyz2"),
                   (graddist1, "graddist1"), (graddist2, "graddist2")):
        _lib.require(t, torch.float32, nme)
    _lib.require(idx1, torch.int32, "idx1")
    _lib.require(idx2, torch.int32, "idx2")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    with torch.cuda.device(xyz1.device):
        return _lib.lib.genpc_chamfer_backward(
            b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(graddist1), _lib.ptr(idx1),
            _lib.ptr(graddist2), _lib.ptr(idx2), _lib.ptr(gradxyz1), _lib.ptr(gradxyz2),
            _lib.stream_of(xyz1))


def nm_distance(xyz, xyz2, result, result_i):
    """One direction only (NmDistanceKernel, chamfer3D.cu:12-134)."""
    _lib.require_gpu(xyz, xyz2, result, result_i)
    _lib.require(xyz, torch.float32, "xyz")
    _lib.require(xyz2, torch.float32, "xyz2")
    _lib.require(result, torch.float32, "result")
    _lib.require(result_i, torch.int32, "result_i")
    b, n, _ = xyz.shape
    m = xyz2.shape[1]
    with torch.cuda.device(xyz.device):
        return _lib.lib.genpc_nm_distance(b, n, _lib.ptr(xyz), m, _lib.ptr(xyz2), _lib.ptr(result),
                                          _lib.ptr(result_i), _lib.stream_of(xyz))
