"""``emdModule`` / ``emdFunction`` with the reference's interface
(loss_functions/emd/emd_module.py:29-95) on the gfx950 library.

Input: xyz1 (prediction), xyz2 (ground truth): [B, n, 3], same n, n % 256 == 0,
B <= 512; eps, iters as in the reference.  Output: dist [B, n] (squared distance
to the assigned point; sqrt -> L2), assignment [B, n] int32 (not guaranteed to be
a bijection: the last round force-assigns, emd_cuda.cu:201).  Gradient flows to
xyz1 only (:83-87).

Differences from the reference: scratch is allocated on the INPUT's device (the
reference hard-codes "cuda" = current device, :41-54), and the return code is
checked (-1 -> ValueError like the asserts at :36-39 would).
"""
import torch
from torch import nn
from torch.autograd import Function

from ... import _lib, emd


def alloc_state(batchsize, n, m, device):
    """The 12 scratch/output tensors of emd_module.py:43-54, same shapes, dtypes and initial values.  They are views of
    two allocations -- one zero-filled, one filled with -1 -- instead of twelve: the reference's twelve fills are twelve
    launches in front of the auction on the same stream (3-4 us each: 40 us of a 1 ms call at n = 16384)."""
    bn, bm = batchsize * n, batchsize * m
    sizes = [("dist", bn), ("price", bm), ("bid", bn), ("bid_increments", bn), ("max_increments", bm), ("unass_idx", bn),
             ("max_idx", bm), ("unass_cnt", 512), ("unass_cnt_sum", 512), ("cnt_tmp", 512)]
    pad = lambda k: (k + 63) // 64 * 64            # 256-byte aligned views
    zero = torch.zeros(sum(pad(k) for _, k in sizes), dtype=torch.int32, device=device)
    neg = torch.full((pad(bn) + pad(bm),), -1, dtype=torch.int32, device=device)
    out, off = {}, 0
    for name, k in sizes:
        out[name] = zero[off:off + k]
        off += pad(k)
    for name in ("dist", "price", "bid_increments", "max_increments"):
        out[name] = out[name].view(torch.float32)
    for name, rows, cols in (("dist", batchsize, n), ("price", batchsize, m), ("bid", batchsize, n), ("bid_increments", batchsize, n),
                             ("max_increments", batchsize, m)):
        out[name] = out[name].view(rows, cols)
    out["assignment"] = neg[:bn].view(batchsize, n)
    out["assignment_inv"] = neg[pad(bn):pad(bn) + bm].view(batchsize, m)
    return out


class emdFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()

        assert n == m
        assert xyz1.size()[0] == xyz2.size()[0]
        assert n % 256 == 0
        assert batchsize <= 512

        # emd_module.py:41-42 moves its inputs to the GPU itself (`.contiguous().float().cuda()`);
        # so does this, when there is a GPU -- there is no CPU implementation to fall back to
        if not (xyz1.is_cuda and xyz2.is_cuda):
            if not torch.cuda.is_available():
                raise RuntimeError("genpc_amd: GPU tensors only (no GPU is available to move the inputs to); "
                                   "the HIP path has no CPU fallback")
            dev = xyz1.device if xyz1.is_cuda else (xyz2.device if xyz2.is_cuda else torch.device("cuda"))
            xyz1, xyz2 = xyz1.to(dev), xyz2.to(dev)
        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        s = alloc_state(batchsize, n, m, xyz1.device)
        rc = emd.forward(xyz1, xyz2, s["dist"], s["assignment"], s["price"], s["assignment_inv"],
                         s["bid"], s["bid_increments"], s["max_increments"], s["unass_idx"],
                         s["unass_cnt"], s["unass_cnt_sum"], s["cnt_tmp"], s["max_idx"], eps, iters)
        if rc == -1:
            raise ValueError("emd.forward: invalid shape (n != m, B > 512 or n % 256 != 0)")
        if rc != 1:
            raise RuntimeError("emd.forward failed: " + _lib.last_error())
        ctx.save_for_backward(xyz1, xyz2, s["assignment"])
        ctx.mark_non_differentiable(s["assignment"])
        return s["dist"], s["assignment"]

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        graddist = graddist.contiguous()
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        rc = emd.backward(xyz1, xyz2, gradxyz1, graddist, assignment)
        if rc != 1:
            raise RuntimeError("emd.backward failed: " + _lib.last_error())
        return gradxyz1, gradxyz2, None, None


class emdModule(nn.Module):
    def __init__(self):
        super(emdModule, self).__init__()

    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)
