"""CPU oracle for the evaluation metrics.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

numpy restatement of ``PAEval`` (lib/metrics/pa_eval.py:40-124 upstream) and ``_PCKMetric`` (lib/metrics/pck.py:36-148).
PINNED: tests/golden/metrics.npz holds the measures the reference's own classes produce on seeded inputs
(tests/golden/make_golden.py::run_metrics)."""
import numpy as np


def orthogonal_procrustes(A, B):
    """scipy.linalg.orthogonal_procrustes (published algorithm): R = argmin ||A R - B||_F, scale = sum of singular
    values of A^T B.  u, w, vt = svd((B^T A)^T); R = u vt."""
    u, w, vt = np.linalg.svd(B.T.dot(A).T)
    return u.dot(vt), w.sum()


def align_w_scale(mtx1, mtx2):
    """pa_eval.py:104-124."""
    t1, t2 = mtx1.mean(0), mtx2.mean(0)
    m1, m2 = mtx1 - t1, mtx2 - t2
    s1 = np.linalg.norm(m1) + 1e-8
    m1 = m1 / s1
    s2 = np.linalg.norm(m2) + 1e-8
    m2 = m2 / s2
    R, s = orthogonal_procrustes(m1, m2)
    return np.dot(m2, R.T) * s * s1 + t1


def get_dist(x, y):
    """pa_eval.py:40-43."""
    return np.mean(np.linalg.norm(x - y, axis=2), axis=1)


def pa_measures(pred_j, gt_j, pred_v=None, gt_v=None):
    """Averages PAEval reports after feeding one batch (pa_eval.py:45-83)."""
    out = {}
    al = np.stack([align_w_scale(gt_j[i], pred_j[i]) for i in range(len(pred_j))])
    out["pa_mpjpe"] = float(np.sum(get_dist(al, gt_j)) / len(pred_j))
    out["mpjpe"] = float(np.sum(get_dist(pred_j, gt_j)) / len(pred_j))
    if pred_v is not None:
        al = np.stack([align_w_scale(gt_v[i], pred_v[i]) for i in range(len(pred_v))])
        out["pa_mpvpe"] = float(np.sum(get_dist(al, gt_v)) / len(pred_v))
        out["mpvpe"] = float(np.sum(get_dist(pred_v, gt_v)) / len(pred_v))
    return out


def pck_measures(pred, gt, val_min, val_max, steps):
    """pck.py:60-148 for all-visible key points: per-kp distance lists -> epe, pck curve, auc."""
    d = np.sqrt(np.sum(np.square(pred - gt), axis=-1))            # (B, P) float32
    thresholds = np.linspace(val_min, val_max, steps)
    area = getattr(np, "trapezoid", getattr(np, "trapz", None))(np.ones_like(thresholds), thresholds)
    curve = np.stack([[np.mean((d[:, k] <= t).astype("float")) for t in thresholds] for k in range(d.shape[1])])
    auc = getattr(np, "trapezoid", getattr(np, "trapz", None))(curve, thresholds, axis=1) / area
    epe = d.mean(0)
    return {"epe_mean_per_kp": epe, "pck_curve_per_kp": curve, "auc_per_kp": auc, "epe_mean_all": float(np.mean(epe)),
            "auc_all": float(np.mean(auc)), "pck_002": float(np.mean([(d[:, k] <= 0.02).mean() for k in range(d.shape[1])]))}


MANO_TIP_VERTICES = (744, 320, 443, 555, 672)          # CONST.MANO_KPID_2_VERTICES, lib/utils/misc.py:76-82
OPENPOSE_FROM_MANO = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)


def mano_to_openpose(J_regressor, mano_verts):
    """lib/utils/transform.py:836-872: regressed joints (16,778)@(B,778,3), the five tip vertices appended, re-ordered."""
    J = np.asarray(J_regressor, dtype=np.float32)
    V = np.asarray(mano_verts, dtype=np.float32)
    joints = np.einsum("jv,bvd->bjd", J.astype(np.float64), V.astype(np.float64)).astype(np.float32)
    tips = V[:, list(MANO_TIP_VERTICES)]
    return np.concatenate([joints, tips], axis=1)[:, list(OPENPOSE_FROM_MANO)]


def synthetic_j_regressor(seed=11):
    """Seeded stand-in for MANO's th_J_regressor (licence-gated): (16,778), rows non-negative, ~20 non-zeros each, sum 1."""
    rng = np.random.RandomState(seed)
    J = np.zeros((16, 778), dtype=np.float32)
    for r in range(16):
        cols = rng.choice(778, size=20, replace=False)
        w = rng.rand(20).astype(np.float32)
        J[r, cols] = w / w.sum()
    return J


def synthetic_mano_verts(B=3, seed=11):
    rng = np.random.RandomState(seed + 1)
    return (0.08 * rng.randn(B, 778, 3) + np.array([0.0, 0.0, 0.6])).astype(np.float32)
