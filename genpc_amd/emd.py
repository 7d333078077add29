"""Drop-in for the reference's pybind module ``emd``
(loss_functions/emd/emd.cpp:25-29): ``forward`` with the same 16 arguments and
``backward`` with the same 5, over libgenpc_hip.so's C ABI.
"""
import torch

from . import _lib


def forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments,
            max_increments, unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters):
    """emd.cpp:12-17 -> emd_cuda.cu:228-282.  Returns 1 ok / 0 HIP error / -1 bad shape."""
    f32 = (("xyz1", xyz1), ("xyz2", xyz2), ("dist", dist), ("price", price),
           ("bid_increments", bid_increments), ("max_increments", max_increments))
    i32 = (("assignment", assignment), ("assignment_inv", assignment_inv), ("bid", bid),
           ("unass_idx", unass_idx), ("unass_cnt", unass_cnt), ("unass_cnt_sum", unass_cnt_sum),
           ("cnt_tmp", cnt_tmp), ("max_idx", max_idx))
    _lib.require_gpu(*[t for _, t in f32 + i32])
    for nme, t in f32:
        _lib.require(t, torch.float32, nme)
    for nme, t in i32:
        _lib.require(t, torch.int32, nme)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    p = _lib.ptr
    with torch.cuda.device(xyz1.device):
        return _lib.lib.genpc_emd_forward(
            b, n, m, p(xyz1), p(xyz2), p(dist), p(assignment), p(price), p(assignment_inv), p(bid),
            p(bid_increments), p(max_increments), p(unass_idx), p(unass_cnt), p(unass_cnt_sum),
            p(cnt_tmp), p(max_idx), float(eps), int(iters), _lib.stream_of(xyz1))


def backward(xyz1, xyz2, gradxyz, graddist, idx):
    """emd.cpp:19-23 -> emd_cuda.cu:302-316."""
    _lib.require_gpu(xyz1, xyz2, gradxyz, graddist, idx)
    for nme, t in (("xyz1", xyz1), ("xyz2", xyz2), ("gradxyz", gradxyz), ("graddist", graddist)):
        _lib.require(t, torch.float32, nme)
    _lib.require(idx, torch.int32, "idx")
    b, n, _ = xyz1.shape
    p = _lib.ptr
    with torch.cuda.device(xyz1.device):
        return _lib.lib.genpc_emd_backward(b, n, p(xyz1), p(xyz2), p(gradxyz), p(graddist), p(idx),
                                           _lib.stream_of(xyz1))
