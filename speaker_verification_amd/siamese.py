"""siamese.py drop-in (`/root/reference/siamese.py`): `Siamese(LAMBDA, M)` with
`forward(model, y, output1, output2)` and `l2_dist(output1, output2)`.

Differences from the reference, all deliberate (Q20): tensors stay on the device
they arrive on (no hard-coded `.cuda()`), nothing is printed, and `l2_dist` on
inference tensors (no autograd) is one gfx950 kernel (`svk_l2_dist`)."""
import torch
import torch.nn as nn

from .engine import get_engine


class Siamese(nn.Module):
    def __init__(self, LAMBDA, M):
        super(Siamese, self).__init__()
        self.LAMBDA = LAMBDA
        self.M = M

    def forward(self, model, y, output1, output2):
        """Contrastive loss (siamese.py:10-27): per pair y * d^2 / 2 for same-speaker,
        (1 - y) * max(0, M - d)^2 / 2 for different, plus LAMBDA * sum_p ||p||_2
        added to every pair, averaged over the batch."""
        dist = self.l2_dist(output1, output2)
        y = y.to(dist.dtype)
        l_gen = 0.5 * dist.pow(2)
        l_imp = 0.5 * torch.clamp(self.M - dist, min=0.0).pow(2)
        l2_reg = torch.zeros((), device=dist.device, dtype=dist.dtype)
        for param in model.parameters():
            l2_reg = l2_reg + torch.norm(param)
        return 1 / y.size()[0] * (y * l_gen + (1 - y) * l_imp + self.LAMBDA * l2_reg).sum()

    def l2_dist(self, output1, output2):
        """Row-wise Euclidean distance (siamese.py:29-30)."""
        needs_grad = torch.is_grad_enabled() and (output1.requires_grad or output2.requires_grad)
        if needs_grad or not output1.is_cuda:
            if not output1.is_cuda and not needs_grad:
                raise RuntimeError("Siamese.l2_dist runs on the GPU (svk_l2_dist); move the embeddings to the "
                                   "device -- there is no CPU fallback")
            return (output1 - output2).pow(2).sum(dim=1).sqrt()
        return get_engine(output1.device.index).l2_dist(output1, output2)
