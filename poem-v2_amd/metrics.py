"""Mean end-point error accumulated on the device and reduced across ranks with one all-reduce.

Same definition as the reference's ``MeanEPE.feed`` (lib/metrics/mean_epe.py:23-33 upstream): per sample the mean over
points of the L2 distance, summed over the batch; the running average divides by the number of samples.  The
reference calls ``.item()`` every batch (a host sync); here the two sums stay on the device until ``result()``."""
import torch

from . import dist as pdist


class MeanEPE:
    def __init__(self, name="", device="cpu"):
        self.name = f"{name}_mepe"
        self.acc = torch.zeros(2, dtype=torch.float64, device=device)   # this rank's [sum of per-sample means, n samples]
        self._global = None                                               # the all-reduced pair of the last reduce()

    def reset(self):
        self.acc.zero_()
        self._global = None

    def feed(self, pred_kp, gt_kp):
        assert pred_kp.dim() == 3, "pred shape should be (BATCH, NPOINTS, 1|2|3)"
        d = torch.norm(pred_kp - gt_kp, p="fro", dim=2).mean(dim=1)
        self.acc[0] += d.sum().double()
        self.acc[1] += d.shape[0]
        self._global = None

    def reduce(self):
        """all-reduce(sum) of [sum, count] over the process group (16 bytes: the path's only collective).  The local
        sums are left alone, so reducing after every step does not count earlier steps once per rank."""
        self._global = pdist.all_reduce_sum_(self.acc.clone())
        return self

    def result(self):
        s, n = (self.acc if self._global is None else self._global).tolist()
        return s / max(n, 1.0)

    def __str__(self):
        return f"{self.name}: {self.result():6.4f}"


# ---- the remaining evaluation metrics, accumulated on the device ---------------------------------
def _dev32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class PAEval:
    """Procrustes-aligned MPJPE / MPVPE -- device counterpart of ``PAEval`` (lib/metrics/pa_eval.py:16-124 upstream).

    Same ``feed`` signature, ``get_measures`` keys, ``get_result`` and string form.  The reference copies every batch to
    the host and calls scipy once per sample; here ``poem_pa_epe`` aligns the whole batch in one launch and the four
    sums stay on the device.  No CPU fallback."""

    def __init__(self, cfg=None, mesh_score=False, device="cuda"):
        self.mesh_score = mesh_score
        self.device = torch.device(device)
        self.acc = torch.zeros(5, dtype=torch.float64, device=self.device)   # this rank's pa_j, j, pa_v, v, count
        self._global = None                                                   # the all-reduced sums of the last reduce()
        self.count = 0

    def reset(self):
        self.acc.zero_()
        self._global = None
        self.count = 0

    def _pair(self, pred, gt):
        from . import hip
        pred, gt = _dev32(pred, self.device), _dev32(gt, self.device)
        if not pred.is_cuda:
            raise RuntimeError("PAEval runs on the MI355X HIP path only (no CPU fallback)")
        out = torch.empty(pred.shape[0], 2, dtype=torch.float32, device=self.device)
        hip.check(hip.lib().poem_pa_epe(hip.ptr(pred), hip.ptr(gt), hip.ptr(out), pred.shape[0], pred.shape[1],
                                        hip.stream()), "poem_pa_epe")
        return out.double().sum(0)

    def feed(self, pred_joints_3d_abs, joints_3d_abs, pred_verts_3d_abs=None, verts_3d_abs=None, **kwargs):
        b = pred_joints_3d_abs.shape[0]
        self.acc[0:2] += self._pair(pred_joints_3d_abs, joints_3d_abs)
        if self.mesh_score:
            self.acc[2:4] += self._pair(pred_verts_3d_abs, verts_3d_abs)
        self.acc[4] += b
        self.count += b
        self._global = None

    def reduce(self):
        """all-reduce(sum) of a copy: the rank-local sums stay what this rank fed, so a second ``reduce()`` or a ``feed``
        after it never counts anything once per rank (same contract as ``MeanEPE.reduce``)."""
        self._global = pdist.all_reduce_sum_(self.acc.clone())
        return self

    def get_measures(self, **kwargs):
        a = (self.acc if self._global is None else self._global).tolist()
        n = max(a[4], 1.0)
        m = {"pa_mpjpe": a[0] / n, "mpjpe": a[1] / n}
        if self.mesh_score:
            m["pa_mpvpe"], m["mpvpe"] = a[2] / n, a[3] / n
        return m

    def get_result(self):
        return self.get_measures()["pa_mpjpe"]

    def __str__(self):
        m = self.get_measures()
        s = f"pa_mpjpe(mm): {m['pa_mpjpe'] * 1000.0 :6.4f} | mpjpe: {m['mpjpe']:6.4f}"
        if self.mesh_score:
            s += f" | pa_mpvpe(mm): {m['pa_mpvpe'] * 1000.0:6.4f} | mpvpe: {m['mpvpe']:6.4f}"
        return s


class _PCKMetric:
    """PCK / AUC accumulators -- device counterpart of ``_PCKMetric`` (lib/metrics/pck.py:11-148 upstream): same config
    keys (VAL_MIN, VAL_MAX, STEPS, EVAL_TYPE), ``feed(preds, targs)``, ``get_pck_all`` and ``get_measures`` keys."""
    num_kp = 0
    _keys = {}

    def __init__(self, device="cuda", **cfg):
        self.val_min, self.val_max, self.steps = float(cfg["VAL_MIN"]), float(cfg["VAL_MAX"]), int(cfg["STEPS"])
        self.device = torch.device(device)
        self.reset()

    def reset(self):
        self.counts = torch.zeros(self.num_kp, self.steps, dtype=torch.int32, device=self.device)
        self.sum = torch.zeros(self.num_kp, dtype=torch.float64, device=self.device)
        self.n = torch.zeros(self.num_kp, dtype=torch.int32, device=self.device)
        self.dists = []          # per-batch (B, num_kp) distance tensors on the device (arbitrary-threshold queries)
        self.count = 0
        self._global = None      # (counts int64, n int64, sum fp64) over all ranks, set by reduce()

    def _get_predictions(self, preds, targs):
        pk_, tk_ = self._keys[self.eval_type]
        return preds[pk_].reshape(-1, self.num_kp, 3), targs[tk_].reshape(-1, self.num_kp, 3)

    def feed(self, preds, targs, **kwargs):
        from . import hip
        p, t = self._get_predictions(preds, targs)
        p, t = _dev32(p, self.device), _dev32(t, self.device)
        if not p.is_cuda:
            raise RuntimeError("PCK metrics run on the MI355X HIP path only (no CPU fallback)")
        d = torch.empty(p.shape[0], self.num_kp, dtype=torch.float32, device=self.device)
        hip.check(hip.lib().poem_pck_accumulate(hip.ptr(p), hip.ptr(t), p.shape[0], self.num_kp, self.val_min, self.val_max,
                                                self.steps, self.counts.data_ptr(), self.sum.data_ptr(), self.n.data_ptr(),
                                                hip.ptr(d), hip.stream()), "poem_pck_accumulate")
        self.dists.append(d)
        self.count += p.shape[0]
        self._global = None

    def reduce(self):
        """Sum the accumulators over ranks into a separate global copy (counts as int64 to stay exact); the rank-local
        accumulators are untouched, so reducing twice or feeding afterwards never double counts."""
        g = (self.counts.long(), self.n.long(), self.sum.clone())
        for t in g:
            pdist.all_reduce_sum_(t)
        self._global = g
        return self

    def get_pck_all(self, threshold):
        """Fraction of key points within ``threshold`` (every sample carries all key points, so the mean of per-key-point
        means is hits / total).  Before ``reduce()``: this rank's samples.  After it: all ranks' -- read from the reduced
        threshold histogram when ``threshold`` is one of its steps (the usual 0.02 = VAL_MAX is; no communication, so rank 0
        alone may print the metric); for any other threshold this rank's [hits, total] pair is all-reduced here and every
        rank must make the call."""
        import numpy as np
        if self._global is not None:
            steps = np.linspace(self.val_min, self.val_max, self.steps)
            hit = np.nonzero(np.isclose(steps, threshold, rtol=0, atol=1e-12))[0]
            if hit.size:
                counts, n, _ = self._global
                return float(counts[:, int(hit[0])].sum()) / max(float(n.sum()), 1.0)
        d = torch.cat(self.dists, 0) if self.dists else torch.zeros(0, self.num_kp, device=self.device)
        pair = torch.stack([(d.double() <= threshold).sum().double(), torch.tensor(float(d.numel()), dtype=torch.float64, device=d.device)])
        if self._global is not None:
            pdist.all_reduce_sum_(pair)
        hits, total = pair.tolist()
        return hits / max(total, 1.0)

    def get_measures(self):
        import numpy as np
        thresholds = np.linspace(self.val_min, self.val_max, self.steps)
        area_under_one = getattr(np, "trapezoid", getattr(np, "trapz", None))(np.ones_like(thresholds), thresholds)
        counts, n, dsum = (self.counts, self.n, self.sum) if self._global is None else self._global
        n = n.cpu().numpy().astype(np.float64)
        valid = n > 0
        curve = counts.cpu().numpy().astype(np.float64)[valid] / n[valid][:, None]
        epe = dsum.cpu().numpy()[valid] / n[valid]
        auc = getattr(np, "trapezoid", getattr(np, "trapz", None))(curve, thresholds, axis=1) / area_under_one
        return {"epe_mean_per_kp": epe, "pck_curve_per_kp": curve, "auc_per_kp": auc, "epe_mean_all": float(np.mean(epe)),
                "auc_all": float(np.mean(auc)), "thresholds": thresholds}

    def __str__(self):
        return f"h3dpck: {self.get_pck_all(0.02):6.4f}"


class Joint3DPCK(_PCKMetric):
    num_kp = 21
    _keys = {"joints_3d": ("pred_joints_3d", "master_joints_3d"), "joints_3d_rel": ("pred_joints_3d_rel", "master_joints_3d_rel")}

    def __init__(self, device="cuda", **cfg):
        self.eval_type = cfg.get("EVAL_TYPE", "joints_3d")
        if self.eval_type not in self._keys:
            raise ValueError(f"Unknown eval_type {self.eval_type} in {type(self).__name__}")
        super().__init__(device=device, **cfg)


class Vert3DPCK(_PCKMetric):
    num_kp = 778
    _keys = {"verts_3d": ("pred_verts_3d", "master_verts_3d"), "verts_3d_rel": ("pred_verts_3d_rel", "master_verts_3d_rel")}

    def __init__(self, device="cuda", **cfg):
        self.eval_type = cfg.get("EVAL_TYPE", "verts_3d")
        if self.eval_type not in self._keys:
            raise ValueError(f"Unknown eval_type {self.eval_type} in {type(self).__name__}")
        super().__init__(device=device, **cfg)


def mano_to_openpose(J_regressor, mano_verts):
    """MANO vertices (B,778,3) -> joints (B,21,3) in OpenPose order: the reference function of the same name
    (lib/utils/transform.py:836-872 upstream), which ``testing_step`` applies to predicted and ground-truth vertices
    before the joint metrics (lib/models/POEM.py:602-603).  ``J_regressor`` is MANO's ``th_J_regressor`` (16,778) -- an
    input, the asset is licence-gated.  One launch on the device; no CPU fallback."""
    from . import hip
    if not mano_verts.is_cuda:
        raise RuntimeError("mano_to_openpose runs on the MI355X HIP path only (no CPU fallback)")
    v = _dev32(mano_verts, mano_verts.device)
    w = _dev32(J_regressor, mano_verts.device)
    if tuple(w.shape) != (16, 778) or v.dim() != 3 or tuple(v.shape[1:]) != (778, 3):
        raise ValueError("J_regressor must be (16,778) and mano_verts (B,778,3)")
    out = torch.empty(v.shape[0], 21, 3, dtype=torch.float32, device=v.device)
    hip.check(hip.lib().poem_mano_to_openpose(hip.ptr(w), hip.ptr(v), hip.ptr(out), v.shape[0], 778, hip.stream()),
              "poem_mano_to_openpose")
    return out
