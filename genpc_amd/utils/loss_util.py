"""``Completionloss`` -- the reference's loss facade (utils/loss_util.py:8-53),
same constructor argument, same five reductions, same method names.

The reference wraps EMD in ``torch.nn.DataParallel`` (:12), its only multi-device
mechanism; with the batch-of-one inputs every caller passes it is a pass-through.
Here EMD runs on the input's device; scans are sharded one process per GPU
instead (genpc_amd.sharding), so no DataParallel.
"""
import torch

from ..loss_functions import chamfer_3DDist, emdModule


class Completionloss:
    def __init__(self, loss_func='cd_l1'):
        self.loss_func = loss_func
        self.chamfer_dist = chamfer_3DDist()
        self.EMD = emdModule()

        if loss_func == 'cd_l1':
            self.metric = self.chamfer_l1
            self.partial_matching = self.chamfer_partial_l1
        elif loss_func == 'cd_l2':
            self.metric = self.chamfer_l2
            self.partial_matching = self.chamfer_partial_l2
        elif loss_func == 'emd':
            self.metric = self.emd_loss
        else:
            raise Exception('loss function {} not supported yet!'.format(loss_func))

    # utils/loss_util.py:25-29
    def chamfer_l1(self, p1, p2):
        d1, d2, _, _ = self.chamfer_dist(p1, p2)
        return (torch.mean(torch.sqrt(d1)) + torch.mean(torch.sqrt(d2))) / 2

    # :31-33
    def chamfer_l2(self, p1, p2):
        d1, d2, _, _ = self.chamfer_dist(p1, p2)
        return torch.mean(d1) + torch.mean(d2)

    # :35-38  (the reference computes both directions and drops d2)
    def chamfer_partial_l1(self, pcd1, pcd2):
        d1, _, _, _ = self.chamfer_dist(pcd1, pcd2)
        return torch.mean(torch.sqrt(d1))

    # :40-43
    def chamfer_partial_l2(self, pcd1, pcd2):
        d1, _, _, _ = self.chamfer_dist(pcd1, pcd2)
        return torch.mean(d1)

    # :45-49
    def emd_loss(self, p1, p2):
        d1, _ = self.EMD(p1, p2, eps=0.005, iters=50)
        return torch.sqrt(d1).mean(1).mean()

    def get_loss(self, gen, gt):
        return self.metric(gen, gt)
