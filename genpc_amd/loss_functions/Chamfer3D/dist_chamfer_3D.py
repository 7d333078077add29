"""``chamfer_3DDist`` / ``chamfer_3DFunction`` with the reference's interface
(loss_functions/Chamfer3D/dist_chamfer_3D.py:26-74) on the gfx950 library.

Differences from the reference, none of them visible in results:
  * outputs are allocated on the device directly (the reference builds zero
    tensors on the host and copies them over: four H2D copies per call, :33-42);
  * the library's return code is checked (the reference drops it, :45);
  * work is enqueued on torch's current stream of the input's device.
GPU tensors only, as in the reference (:25).
"""
import torch
from torch import nn
from torch.autograd import Function

from ... import _lib, chamfer_3D


class chamfer_3DFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.size(2) != 3 or xyz2.size(2) != 3:
            raise ValueError("chamfer_3DDist expects [B,N,3] and [B,M,3]")
        if xyz1.size(0) != xyz2.size(0):
            raise ValueError("chamfer_3DDist: batch sizes differ")
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        device = xyz1.device
        # zeros, not empty: with an empty cloud the kernels write nothing and the
        # reference returns its zero-initialised buffers
        alloc = torch.zeros if (n == 0 or m == 0 or batchsize == 0) else torch.empty
        dist1 = alloc(batchsize, n, device=device, dtype=torch.float32)
        dist2 = alloc(batchsize, m, device=device, dtype=torch.float32)
        idx1 = alloc(batchsize, n, device=device, dtype=torch.int32)
        idx2 = alloc(batchsize, m, device=device, dtype=torch.int32)
        rc = chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        if rc != 1:
            raise RuntimeError("chamfer_3D.forward failed: " + _lib.last_error())
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = graddist1.contiguous()
        graddist2 = graddist2.contiguous()
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        rc = chamfer_3D.backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        if rc != 1:
            raise RuntimeError("chamfer_3D.backward failed: " + _lib.last_error())
        return gradxyz1, gradxyz2


class chamfer_3DDist(nn.Module):
    def __init__(self):
        super(chamfer_3DDist, self).__init__()

    def forward(self, input1, input2):
        input1 = input1.contiguous()
        input2 = input2.contiguous()
        return chamfer_3DFunction.apply(input1, input2)
