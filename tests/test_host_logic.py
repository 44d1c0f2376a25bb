"""CPU-side tests: plugin boundary, checkpoint surface, C-ABI export list, synthetic inputs, DP glue (gloo, world 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import poem_v2_amd as pk
from poem_v2_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- plugin boundary ------------------------------------------------------------------------------------------
def test_registry_and_build_from_cfg():
    assert pk.HEAD.get("POEM_Generalized_Head") is pk.POEM_Generalized_Head
    assert pk.TRANSFORMER.get("PtEmbedTRv4") is pk.PtEmbedTRv4
    reg = pk.Registry("toy")

    @reg.register_module()
    class Foo:
        def __init__(self, cfg):
            self.cfg = cfg

    with pytest.raises(KeyError):
        reg.register_module(module=Foo)
    obj = pk.build_from_cfg(pk.CN({"TYPE": "Foo", "A": 1}), reg, data_preset=pk.CN({"X": 2}))
    assert obj.cfg.A == 1 and obj.cfg.DATA_PRESET.X == 2          # kwargs arrive as UPPER-CASE keys
    with pytest.raises(KeyError):
        pk.build_from_cfg(pk.CN({"TYPE": "Nope"}), reg)
    with pytest.raises(TypeError):
        reg.register_module(force=1)


def test_cn_semantics():
    c = pk.CN({"A": {"B": 1}, "L": [{"x": 1}]})
    assert c.A.B == 1 and c.get("Z", 7) == 7 and c.L[0].x == 1
    d = c.clone()
    d.A.B = 2
    assert c.A.B == 1
    c.freeze()
    with pytest.raises(AttributeError):
        c.A.B = 3
    c.defrost()
    c.merge_from_other_cfg(pk.CN({"A": {"C": 5}}))
    assert c.A.B == 1 and c.A.C == 5


def test_head_state_dict_keys_and_checkpoint_loading():
    head = pk.build_head(pk.configs.model_head_cfg("medium"), data_preset=pk.CN({}))
    want = pk.weights.live_key_shapes(256)
    sd = head.state_dict()
    assert list(sd.keys()) == list(want.keys()) or set(sd.keys()) == set(want.keys())
    assert sum(v.numel() for v in sd.values()) == 7209481          # 7.21 M live parameters
    assert head.num_preds == 3
    # a "reference checkpoint": full-model prefixes + dead tensors
    ref = {"module.ptEmb_head." + k: v for k, v in pk.weights.seeded_state_dict(256, seed=5).items()}
    ref["module.ptEmb_head.center_shift_layer.0.weight"] = torch.zeros(799, 799)
    ref["module.ptEmb_head.transformer.pt_metro_encoder.0.embeddings.word_embeddings.weight"] = torch.zeros(8, 256)
    ref["module.img_backbone.conv1.weight"] = torch.zeros(4)
    ignored = head.load_reference_state_dict(ref)
    assert len(ignored) == 3
    assert torch.equal(head.state_dict()["input_proj.weight"], ref["module.ptEmb_head.input_proj.weight"])
    bad = dict(ref)
    bad["module.ptEmb_head.input_proj.bias"] = torch.zeros(3)
    with pytest.raises(ValueError):
        head.load_reference_state_dict(bad)
    del bad["module.ptEmb_head.input_proj.bias"]
    with pytest.raises(KeyError):
        head.load_reference_state_dict(bad)


def test_parametric_head_has_mano_tail_keys():
    head = pk.build_head(pk.configs.model_head_cfg("medium_MANO"), data_preset=pk.CN({}))
    keys = set(head.state_dict().keys())
    assert "transformer.pt_metro_encoder.2.flat_verts.weight" in keys
    assert "transformer.pt_metro_encoder.2.mano_linear.bias" in keys
    assert keys == set(pk.weights.live_key_shapes(256, parametric=True).keys())


def test_no_cpu_fallback():
    head = pk.build_head(pk.configs.model_head_cfg("small"), data_preset=pk.CN({})).eval()
    b = pk.inputs.synthetic_batch([2], seed=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"), torch.no_grad():
        head(b["mlvl_feat"], b["img_metas"], b["reference_joints"])
    with pytest.raises(RuntimeError):
        hip.ptr(torch.zeros(3))


def test_head_refuses_training():
    """The HIP head has no backward: the reference's training loop (scripts/train_ddp.py:84, losses at lib/models/POEM.py:363-466
    upstream) must get an error, not a head that silently returns graph-less tensors."""
    head = pk.build_head(pk.configs.model_head_cfg("small"), data_preset=pk.CN({}))
    b = pk.inputs.synthetic_batch([2], seed=0)
    assert head.training
    with pytest.raises(RuntimeError, match="inference-only"):
        head(b["mlvl_feat"], b["img_metas"], b["reference_joints"])                  # train mode, autograd on
    head.eval()
    with pytest.raises(RuntimeError, match="inference-only"):
        head(b["mlvl_feat"].clone().requires_grad_(True), b["img_metas"], b["reference_joints"])   # eval mode, but a gradient is asked for
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        head(b["mlvl_feat"], b["img_metas"], b["reference_joints"])                  # eval mode, nothing asks for a gradient: past the guard
    head.train()
    with pytest.raises(RuntimeError, match="no CPU fallback"), torch.no_grad():
        head(b["mlvl_feat"], b["img_metas"], b["reference_joints"])                  # train mode under no_grad (validation inside a training script)


def test_engine_is_rebuilt_when_a_parameter_object_is_replaced(monkeypatch):
    """The engine holds packed copies of the weights and raw pointers into the Parameters: any way a weight can change must
    rebuild it -- in-place edits (version counter), load_state_dict, .data swaps, and a Parameter OBJECT being replaced
    (load_state_dict(assign=True), `module.weight = nn.Parameter(...)`), which a cached list of Parameter objects cannot see."""
    built = []

    class FakeEngine:
        def __init__(self, cfg, weights, *a):
            built.append(float(weights["input_proj.weight"].flatten()[0]))

    monkeypatch.setattr(hip, "Engine", FakeEngine)
    head = pk.build_head(pk.configs.model_head_cfg("small"), data_preset=pk.CN({})).eval()
    head._engine_for("cpu"); head._engine_for("cpu")
    assert len(built) == 1
    with torch.no_grad():
        head.input_proj.weight.mul_(2.0)                                  # in place: version counter
    head._engine_for("cpu")
    assert len(built) == 2
    head.input_proj.weight = torch.nn.Parameter(torch.full_like(head.input_proj.weight, 3.0))     # the object is replaced
    head._engine_for("cpu")
    assert len(built) == 3 and built[-1] == 3.0
    sd = {k: torch.full_like(v, 5.0) for k, v in head.state_dict().items()}
    head.load_state_dict(sd, assign=True)
    head._engine_for("cpu")
    assert len(built) == 4 and built[-1] == 5.0
    head._engine_for("cpu")
    assert len(built) == 4
    # a replaced SUBMODULE (round 5: seen at the very next forward, not within 256): a fresh Conv2d, a parametrization
    import copy
    conv = copy.deepcopy(head.input_proj)
    with torch.no_grad():
        conv.weight.fill_(7.0)
    head.input_proj = conv
    head._engine_for("cpu")
    assert len(built) == 5 and built[-1] == 7.0

    class Twice(torch.nn.Module):
        def forward(self, w):
            return 2.0 * w

    torch.nn.utils.parametrize.register_parametrization(head.input_proj, "weight", Twice())
    head._engine_for("cpu")
    assert len(built) == 6 and built[-1] == 14.0
    head._engine_for("cpu")
    assert len(built) == 6


# ---- C ABI ------------------------------------------------------------------------------------------------------
def _header_functions():
    txt = open(os.path.join(ROOT, "include", "poem_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(poem_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 25
    L = hip.lib()                                   # loads without a GPU
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/poem_hip.h but not exported"
    assert sorted(hip.SIGNATURES) == names          # the ctypes table mirrors the header one to one
    assert L.poem_abi_version() == hip.ABI_VERSION == 3
    assert L.poem_error_string(-2) == b"workspace too small"


def test_every_option_name_is_documented_in_the_header():
    """poem_set_option's names (csrc/handle.cpp) against the comment of its declaration in include/poem_hip.h: a switch a
    maintainer cannot find in the header does not exist for them."""
    src = open(os.path.join(ROOT, "poem-v2_amd", "csrc", "handle.cpp")).read()
    body = src[src.index("int poem_set_option("):]
    body = body[:body.index("\n}\n")]
    names = sorted(set(re.findall(r'k == "([a-z0-9_]+)"', body)))
    assert len(names) >= 20, names
    hdr = open(os.path.join(ROOT, "include", "poem_hip.h")).read()
    doc = hdr[:hdr.index("int poem_set_option(")]
    doc = doc[doc.rindex("/*"):]
    missing = [n for n in names if f'"{n}"' not in doc]
    assert not missing, missing


def test_library_tensor_table_matches_python():
    import ctypes
    L = hip.lib()
    for model, par in (("small", False), ("medium", False), ("large", False), ("medium", True)):
        C = pk.weights.MODEL_EMBED[model]
        cfg = hip.make_config(C, parametric=par)
        shapes = pk.weights.live_key_shapes(C, parametric=par)
        assert L.poem_num_weight_tensors(ctypes.byref(cfg)) == len(shapes)
        for i, s in enumerate(shapes.values()):
            assert L.poem_weight_tensor_numel(ctypes.byref(cfg), i) == int(np.prod(s))
        assert L.poem_packed_bytes(ctypes.byref(cfg)) > 0
    bad = hip.make_config(100)
    assert L.poem_packed_bytes(ctypes.byref(bad)) == 0
    assert L.poem_num_weight_tensors(ctypes.byref(bad)) < 0
    assert L.poem_gemm(None, 8, None, None, None, 0, None, 8, 1, 8, 8, 0, None) == -1     # argument check, no launch


# ---- synthetic inputs -------------------------------------------------------------------------------------------
def test_synthetic_rig_projects_inside_the_image():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import poem_oracle as po
    b = pk.inputs.synthetic_batch([1, 2, 8, 10], seed=3)
    m = b["img_metas"]
    assert m["cam_view_num"].tolist() == [1, 2, 8, 10]
    offs = np.concatenate([[0], np.cumsum(m["cam_view_num"])])
    for o in offs[:-1]:
        assert torch.equal(m["cam_extr"][o], torch.eye(4))          # view 0 = master = identity
    bps = torch.from_numpy(np.load(os.path.join(hip.ASSETS, "bps.npy")))[0]
    centre = b["reference_joints"][:, 9]
    vs = torch.repeat_interleave(torch.arange(4), torch.tensor([1, 2, 8, 10]))
    uv = po.project_points(bps[None] + centre[:, None], m["cam_intr"], m["cam_extr"], vs)
    assert float(uv.min()) > 0 and float(uv.max()) < 256


# ---- DP glue ----------------------------------------------------------------------------------------------------
def test_shard_ranges_cover_everything():
    from poem_v2_amd.dist import shard_by_views, shard_range
    for n, w in ((32, 8), (33, 8), (5, 8), (64, 3)):
        got = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            got += list(range(lo, hi))
        assert got == list(range(n))
    views = np.random.RandomState(5).randint(2, 11, size=64)
    got = []
    for r in range(8):
        lo, hi = shard_by_views(views, r, 8)
        got += list(range(lo, hi))
    assert got == list(range(64))


def pdist_free_port():
    from poem_v2_amd.dist import free_port
    return free_port()


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import poem_v2_amd as pk
from poem_v2_amd import dist as pdist
from poem_v2_amd.metrics import MeanEPE, PAEval, Joint3DPCK
rank, local, world = pdist.init_from_env(backend="gloo")
assert world == 2
g = torch.Generator().manual_seed(0)
pred = torch.randn(10, 778, 3, generator=g); gt = torch.randn(10, 778, 3, generator=g)
lo, hi = pdist.shard_range(10, rank, world)
m = MeanEPE("v"); m.feed(pred[lo:hi], gt[lo:hi]); m.reduce()
full = MeanEPE("v"); full.feed(pred, gt)
assert abs(m.result() - full.result()) < 1e-6, (m.result(), full.result())
t = torch.tensor([float(rank + 1)], dtype=torch.float64); pdist.all_reduce_max_(t); assert t.item() == 2.0
# reduce() is non-destructive for every metric: twice in a row, then feed-after-reduce, never counts once per rank.
# (feed() itself is a HIP launch; the accumulators are filled by hand here: rank r holds r+1 samples)
m.reduce(); assert abs(m.result() - full.result()) < 1e-6
pa = PAEval(None, mesh_score=True, device="cpu")
pa.acc += torch.tensor([1.0, 2.0, 3.0, 4.0, 1.0], dtype=torch.float64) * (rank + 1)
for _ in range(2):
    pa.reduce()
    meas = pa.get_measures()
    assert abs(meas["pa_mpjpe"] - 1.0) < 1e-12 and abs(meas["mpvpe"] - 4.0) < 1e-12, meas
    assert pa.acc[4].item() == rank + 1                      # local sums untouched
pck = Joint3DPCK(device="cpu", VAL_MIN=0.0, VAL_MAX=0.02, STEPS=5)
pck.n += (rank + 1); pck.counts[:, 2:] += (rank + 1); pck.sum += 0.01 * (rank + 1)
pck.dists.append(torch.full((rank + 1, 21), 0.03 if rank else 0.01))
for _ in range(2):
    pck.reduce()
    g = pck.get_measures()
    assert abs(g["epe_mean_all"] - 0.01) < 1e-12 and abs(g["pck_curve_per_kp"][0, 2] - 1.0) < 1e-12, g
    assert pck.get_pck_all(0.02) == 1.0                      # a histogram step: from the reduced counts, no collective
    assert abs(pck.get_pck_all(0.017) - 1.0 / 3.0) < 1e-12   # any other threshold: global [hits, total] (1 of 3 samples)
    assert int(pck.n[0]) == rank + 1
# input side (N4): shards are dealt by rank (webdataset.split_by_node), no data-path collective; the union is the epoch
to = pk.inputs
d = sys.argv[2]
if rank == 0:
    for si in range(4):
        pk.wds.write_shard(os.path.join(d, f"Toy_mv_test-{si:06d}.tar"),
                           [to.synthetic_frame(10 * si + i, n_cams=2, raw=(48, 32)) for i in range(3)])
pdist.barrier()
cfg = pk.wds.dataset_cfg(os.path.join(d, "Toy_mv_test-{000000..000003}.tar"))
ds = pk.MultiviewWebDataset(cfg, data_preset=cfg.DATA_PRESET, is_train=False, defer_images=True)
assert len(ds.shards) == 2 and ds.shards == pk.wds.expand_urls(cfg.URLS)[rank::2]
keys = [f["__key__"] for f in ds]
allk = [None, None]; dist.all_gather_object(allk, keys)
assert len(keys) == 6 and sorted(allk[0] + allk[1]) == sorted(f"frame{10 * si + i:06d}" for si in range(4) for i in range(3))
assert not set(allk[0]) & set(allk[1])
pdist.barrier()
if rank == 0: print("DP_OK", m.result())
dist.destroy_process_group()
'''


def test_dp_metric_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(pdist_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT, str(tmp_path)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "DP_OK" in out.stdout


_WORKER8 = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from poem_v2_amd import dist as pdist
from poem_v2_amd.metrics import MeanEPE
rank, local, world = pdist.init_from_env(backend="gloo")
assert world == 8
# config c5 as one global batch: 64 ragged samples split by shard_by_views; config c3: 256 samples split by shard_range
views = np.random.RandomState(5).randint(2, 11, size=64)
lo, hi = pdist.shard_by_views(views, rank, world)
spans = [None] * world; dist.all_gather_object(spans, (lo, hi))
assert spans[0][0] == 0 and spans[-1][1] == 64 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1)), spans
loads = [int(views[a:b].sum()) for a, b in spans]
assert max(hi_ - lo_ for lo_, hi_ in spans) <= 9 and max(loads) <= 1.25 * sum(loads) / world, (spans, loads)
l3, h3 = pdist.shard_range(256, rank, world)
assert (l3, h3) == (32 * rank, 32 * rank + 32)
# the path's only collective: metric sums over the shards == the single-process metric, and reduce() twice changes nothing
g = torch.Generator().manual_seed(0)
pred = torch.randn(64, 778, 3, generator=g); gt = torch.randn(64, 778, 3, generator=g)
m = MeanEPE("v"); m.feed(pred[lo:hi], gt[lo:hi]); m.reduce(); m.reduce()
full = MeanEPE("v"); full.feed(pred, gt)
assert abs(m.result() - full.result()) < 1e-6, (m.result(), full.result())
t = torch.tensor([float(rank)], dtype=torch.float64); pdist.all_reduce_max_(t); assert t.item() == 7.0
pdist.barrier()
if rank == 0: print("DP8_OK", spans)
dist.destroy_process_group()
'''


def test_dp_sharding_and_metric_allreduce_gloo_world8(tmp_path):
    """The N = 8 layout of BASELINE configs[2] / [4] on CPU (gloo): shard_range / shard_by_views cover the batch with balanced,
    contiguous ranges, and the sharded metric equals the single-process one (the world-2 test covers the other metrics)."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    port = str(pdist_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "DP8_OK" in out.stdout


def test_eval_single_cfg_edits_match_reference():
    """scripts/eval_single.py reproduces the reference script's YAML edits and exp_id (golden: tests/golden/evalcfg.json,
    recorded by running the reference's own main() with the shell-out stubbed)."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("eval_single", os.path.join(root, "scripts", "eval_single.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for g in json.load(open(os.path.join(root, "tests", "golden", "evalcfg.json"))):
        cfg = m.default_cfg()
        vr = m.edit_cfg(cfg, g["dataset"], g["model"], [g["view_min"], g["view_max"]])
        t, h = cfg["DATASET"]["TEST"], cfg["MODEL"]["HEAD"]
        assert t["TARGET"]["URLS"] == g["urls"] and t["EPOCH_SIZE"] == g["epoch_size"]
        assert t["TARGET"]["EPOCH_SIZE"] == g["target_epoch_size"] and t["TARGET"]["VIEW_RANGE"] == g["view_range"]
        assert h["POSITIONAL_ENCODING"]["NUM_FEATS"] == g["num_feats"] and h["EMBED_DIMS"] == g["embed_dims"]
        assert h["TRANSFORMER"]["INPUT_FEAT_DIM"] == g["input_feat_dim"] and h["POINTS_FEAT_DIM"] == g["points_feat_dim"]
        assert h["TRANSFORMER"]["PARAMETRIC_OUTPUT"] == g["parametric"]
        assert f"{g['dataset']}_view_{vr[0]}_{vr[1]}_{g['model']}" == g["exp_id"]
    import pytest
    with pytest.raises(AssertionError):
        m.edit_cfg(m.default_cfg(), "NoSuchSet", "medium", [1, 2])


def test_internal_launcher_declarations_match_their_definitions():
    """csrc/launchers.h declares the kernel launchers of the .hip files as extern "C": such symbols carry no signature, so
    a drifted declaration still links and then corrupts the call (stream read from the wrong slot).  Compare the
    parameter type lists textually."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_abi_decls", os.path.join(root, "tools", "check_abi_decls.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    decls, bad = mod.main()
    assert len(decls) >= 30 and not bad, bad


def test_bench_refuses_to_report_n_gpus_from_fewer_devices():
    """``python bench.py --gpus N`` without a launcher starts its own ranks; with fewer than N devices it must fail loudly,
    never print a line that claims N GPUs (here: no GPU at all)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "POEM_SINGLE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout), (out.stdout[-500:], out.stderr[-500:])
    assert "\"n_gpus\"" not in out.stdout
