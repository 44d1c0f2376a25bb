"""Pin the CPU oracle against golden vectors captured from the imported reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

import poem_oracle as po
from util import case_setup, load_golden, run_oracle


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


@pytest.mark.parametrize("name", ["tiny", "tinymano"])
def test_tiny_stage_taps(name):
    z, meta = load_golden(name)
    cfg, w, consts, batch = case_setup(meta["spec"])
    taps = {}
    out = run_oracle(cfg, w, consts, batch, taps=taps)
    # sampling stage: full tensors
    assert _maxdiff(taps["x"], z["tap.x"]) < 2e-5                      # input_proj + positional table
    assert _maxdiff(taps["g"], z["tap.g"]) < 2e-5                      # projection + bilinear sampling
    assert _maxdiff(taps["bps_feat"], z["tap.bps_feat"]) < 5e-5        # Q1 + merge (sv and mv)
    assert _maxdiff(taps["pt_xyz"], z["tap.pt_xyz"]) == 0.0
    assert _maxdiff(taps["query_xyz"], z["tap.query_xyz"]) == 0.0
    for i in range(3):
        for k, tol in (("h_cross", 2e-5), ("f_self", 2e-5), ("f_cross", 2e-5), ("feats", 5e-5)):
            assert _maxdiff(taps[f"b{i}.{k}"][:, ::9], z[f"tap.b{i}.{k}"]) < tol, (i, k)
        assert _maxdiff(taps[f"b{i}.xyz"], z[f"tap.b{i}.xyz"]) < 2e-5, i
    assert _maxdiff(out["all_coords_preds"], z["all_coords_preds"]) < 2e-6   # metres
    if meta["spec"]["parametric"]:
        assert _maxdiff(out["pred_pose"], z["pred_pose"]) < 1e-4
        assert _maxdiff(out["pred_shape"], z["pred_shape"]) < 1e-5


@pytest.mark.parametrize("name", ["small", "medium", "large", "ragged", "mediummano"])
def test_release_shapes(name):
    z, meta = load_golden(name)
    cfg, w, consts, batch = case_setup(meta["spec"])
    taps = {}
    out = run_oracle(cfg, w, consts, batch, taps=taps)
    assert _maxdiff(taps["bps_feat"][:, ::64], z["tap.bps_feat"]) < 1e-4
    ref = z["all_coords_preds"]
    got = out["all_coords_preds"].numpy()
    err = np.linalg.norm(got[-1, :, 21:] - ref[-1, :, 21:], axis=-1)     # per-vertex error, metres
    # MPVPE-vs-reference bar of BASELINE.json: 1e-3 mm = 1e-6 m
    assert err.mean() < 1e-6, err.mean()
    assert _maxdiff(got, ref) < 5e-5
    if meta["spec"]["parametric"]:     # medium_MANO tail (Q3 + rot6d -> axis-angle), toy MANO stand-in on both sides
        assert _maxdiff(out["pred_pose"], z["pred_pose"]) < 1e-4
        assert _maxdiff(out["pred_shape"], z["pred_shape"]) < 1e-5


def test_hoisted_cross_attention_is_equivalent():
    z, meta = load_golden("tiny")
    cfg, w, consts, batch = case_setup(meta["spec"])
    out = run_oracle(cfg, w, consts, batch, hoist=True)
    assert _maxdiff(out["all_coords_preds"], z["all_coords_preds"]) < 2e-6


def test_grid_sample_restatement_matches_torch():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 16, 16, generator=g)
    grid = torch.rand(3, 200, 2, generator=g) * 2.6 - 1.3      # includes out-of-range taps (zero padding)
    ref = torch.nn.functional.grid_sample(x, grid[:, :, None], align_corners=False).squeeze(-1)
    assert _maxdiff(po.grid_sample_bilinear(x, grid), ref) < 1e-5


def test_q1_quirk_addressing():
    # SURVEY Q1 probe: N=2, C=256, S=4096, (s,n,c)=(5,1,7) -> n'=0, c'=0, s'=2823
    N, C, S = 2, 256, 4096
    g = torch.arange(N * C * S, dtype=torch.float32).view(N, C, S)
    q = po.q1_rows(g)
    assert q.shape == (S, N, C)
    assert q[5, 1, 7] == g[0, 0, 2823]


def test_mean_epe_known_answer():
    z = np.load(__import__("os").path.join(__import__("util").GOLDEN, "mepe.npz"))
    pred, gt = torch.from_numpy(z["pred"]), torch.from_numpy(z["gt"])
    s1, n1 = po.mean_epe(pred, gt)
    s2, n2 = po.mean_epe(pred[:2] * 2, gt[:2])
    assert abs(s1 - float(z["sum1"])) < 1e-7 and abs(s2 - float(z["sum2"])) < 1e-7
    assert abs((s1 + s2) / (n1 + n2) - float(z["avg"])) < 1e-8
