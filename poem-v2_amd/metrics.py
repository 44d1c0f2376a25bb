"""Mean end-point error accumulated on the device and reduced across ranks with one all-reduce.

Same definition as the reference's ``MeanEPE.feed`` (lib/metrics/mean_epe.py:23-33 upstream): per sample the mean over
points of the L2 distance, summed over the batch; the running average divides by the number of samples.  The
reference calls ``.item()`` every batch (a host sync); here the two sums stay on the device until ``result()``."""
import torch

from . import dist as pdist


class MeanEPE:
    def __init__(self, name="", device="cpu"):
        self.name = f"{name}_mepe"
        self.acc = torch.zeros(2, dtype=torch.float64, device=device)   # [sum of per-sample means, n samples]

    def reset(self):
        self.acc.zero_()

    def feed(self, pred_kp, gt_kp):
        assert pred_kp.dim() == 3, "pred shape should be (BATCH, NPOINTS, 1|2|3)"
        d = torch.norm(pred_kp - gt_kp, p="fro", dim=2).mean(dim=1)
        self.acc[0] += d.sum().double()
        self.acc[1] += d.shape[0]

    def reduce(self):
        """all-reduce(sum) of [sum, count] over the process group (16 bytes: the path's only collective)."""
        pdist.all_reduce_sum_(self.acc)
        return self

    def result(self):
        s, n = self.acc.tolist()
        return s / max(n, 1.0)

    def __str__(self):
        return f"{self.name}: {self.result():6.4f}"
