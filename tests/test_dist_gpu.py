"""The data-parallel path on the one GPU a test box has: (i) the RCCL branch itself -- `nccl` process group bound to
the device, device-tensor all-reduces of every metric, barrier, destroy -- as a ONE-rank group (legal for RCCL; forced by
POEM_DIST_FORCE_INIT=1), i.e. the reference's ``setup_ddp`` sequence (scripts/eval.py:30-43,105 upstream); (ii) `bench.py --gpus 8`
rehearsed with eight gloo ranks sharing cuda:0, including the claim the sharding rests on, checked across processes: the shards of
one global ragged batch of 64 put together are bit-equal to one process running all 64."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import poem_v2_amd as pk
from poem_v2_amd import dist as pdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DROP = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "POEM_SINGLE_DEVICE", "POEM_DIST_BACKEND", "POEM_DIST_FORCE_INIT", "MASTER_PORT",
         "MASTER_ADDR")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in _DROP}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def _launch(nproc, script_and_args, env, timeout):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(pdist.free_port())] + script_and_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)


def _err(out):
    """stderr without RCCL's topology chatter (hundreds of `NCCL WARN Could not read node` lines on a container)."""
    return "\n".join(ln for ln in out.stderr.splitlines() if "NCCL WARN" not in ln and ln.strip())[-3000:]


def _bench_line(out):
    assert out.returncode == 0, (out.stdout[-1500:], _err(out))
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


# ---- host logic of the global-batch leg (CPU) -------------------------------------------------------------------------
def test_global_batch_slices_tile_the_batch():
    """bench.slice_batch over dist.shard_by_views ranges: the shards' views, cameras and joints concatenate to the global
    batch exactly, for every world size of the scaling run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    views = np.random.RandomState(5).randint(2, 11, size=64)
    b = pk.inputs.synthetic_batch(views.tolist(), seed=5000, in_channels=4, feat_hw=2)
    for world in (1, 2, 4, 8):
        parts = [bench.slice_batch(b, *pdist.shard_by_views(views, r, world)) for r in range(world)]
        assert torch.equal(torch.cat([p[0] for p in parts]), b["mlvl_feat"])
        assert torch.equal(torch.cat([p[1]["cam_extr"] for p in parts]), b["img_metas"]["cam_extr"])
        assert torch.equal(torch.cat([p[1]["cam_intr"] for p in parts]), b["img_metas"]["cam_intr"])
        assert torch.equal(torch.cat([p[2] for p in parts]), b["reference_joints"])
        assert np.concatenate([p[1]["cam_view_num"] for p in parts]).tolist() == views.tolist()
        sizes = [len(p[1]["cam_view_num"]) for p in parts]
        assert sum(sizes) == 64 and (world < 8 or (min(sizes) >= 7 and max(sizes) <= 9)), sizes


def test_forced_one_rank_group_gloo():
    """POEM_DIST_FORCE_INIT=1 builds the group at WORLD_SIZE 1 and every helper then really calls the collective (gloo here:
    the host-side half of the RCCL test below)."""
    code = ("import os, sys, torch; sys.path.insert(0, sys.argv[1]);\n"
            "import torch.distributed as dist\n"
            "from poem_v2_amd import dist as pdist\n"
            "from poem_v2_amd.metrics import MeanEPE\n"
            "assert not pdist.active()\n"
            "r, l, w = pdist.init_from_env('gloo'); assert (r, w) == (0, 1) and pdist.active() and dist.get_world_size() == 1\n"
            "t = torch.tensor([3.0], dtype=torch.float64); pdist.all_reduce_sum_(t); pdist.all_reduce_max_(t); assert t.item() == 3.0\n"
            "m = MeanEPE('v'); m.feed(torch.ones(4, 5, 3), torch.zeros(4, 5, 3)); m.reduce(); assert abs(m.result() - 3 ** 0.5) < 1e-6\n"
            "pdist.barrier(); pdist.shutdown(); assert not pdist.active(); pdist.shutdown(); print('FORCED_OK')\n")
    out = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=300,
                         env=_env(POEM_DIST_FORCE_INIT="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert out.returncode == 0 and "FORCED_OK" in out.stdout, out.stderr[-2000:]
    # a multi-rank group without MASTER_PORT is refused, never rendezvoused on a guessed port
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, sys.argv[1]);\n"
                          "from poem_v2_amd import dist as pdist\npdist.init_from_env('gloo')", ROOT],
                         capture_output=True, text=True, timeout=300, env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert out.returncode != 0 and "MASTER_PORT" in out.stderr


# ---- the RCCL branch on the GPU ---------------------------------------------------------------------------------------
_RCCL_WORKER = r'''
import os, sys, torch
for p in ("", "oracle", "tests"):
    sys.path.insert(0, os.path.join(sys.argv[1], p))
import torch.distributed as dist
import poem_v2_amd as pk
from poem_v2_amd import dist as pdist
from poem_v2_amd.metrics import MeanEPE, PAEval, Joint3DPCK, Vert3DPCK
rank, local, world = pdist.init_from_env()                       # default backend on a GPU box: nccl == RCCL, device_id bound
assert (rank, local, world) == (0, 0, 1) and pdist.active()
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1, dist.get_backend()
dev = torch.device("cuda", local)
pdist.barrier()                                                  # setup_ddp's barrier (scripts/eval.py:43 upstream)
g = torch.Generator().manual_seed(0)
pred = torch.randn(6, 778, 3, generator=g).to(dev) * 0.01; gt = torch.randn(6, 778, 3, generator=g).to(dev) * 0.01
jp, jg = pred[:, :21].contiguous(), gt[:, :21].contiguous()
# fp64 device pair through the RCCL all-reduce (the path's only collective)
m = MeanEPE("v", device=dev); m.feed(pred, gt); local_val = m.result(); m.reduce(); m.reduce()
assert m._global is not None and m._global.is_cuda and m._global.dtype == torch.float64
assert m.result() == local_val, (m.result(), local_val)
# five fp64 sums
pa = PAEval(None, mesh_score=True, device=dev); pa.feed(jp, jg, pred, gt); before = pa.get_measures(); pa.reduce()
assert pa._global.is_cuda and pa.get_measures() == before and before["pa_mpjpe"] > 0
# int64 histogram + int64 counts + fp64 sums, and the [hits, total] pair of an off-histogram threshold
for cls, key_p, key_t, p_, t_ in ((Joint3DPCK, "pred_joints_3d", "master_joints_3d", jp, jg), (Vert3DPCK, "pred_verts_3d", "master_verts_3d", pred, gt)):
    pck = cls(device=dev, VAL_MIN=0.0, VAL_MAX=0.02, STEPS=20)
    pck.feed({key_p: p_}, {key_t: t_}); a0 = pck.get_measures(); h0 = pck.get_pck_all(0.0137); pck.reduce()
    assert pck._global[0].dtype == torch.int64 and pck._global[0].is_cuda
    a1 = pck.get_measures()
    assert a1["auc_all"] == a0["auc_all"] and a1["epe_mean_all"] == a0["epe_mean_all"] and pck.get_pck_all(0.0137) == h0
    assert pck.get_pck_all(0.02) == float(pck.counts[:, -1].sum()) / float(pck.n.sum())
t = torch.tensor([5.0], dtype=torch.float64, device=dev); pdist.all_reduce_max_(t); assert t.item() == 5.0
# one step of the bench's shape: head forward -> metric feed -> all-reduce, twice (plain launches, then the captured graph)
from util import batch_to, build_hip_head, case_setup
spec = dict(embed=128, nsample=4096, views=[2, 3], seed=7, parametric=False)
cfg, w, consts, batch = case_setup(spec)
head = build_hip_head(spec, dev)
feat, metas, rj = batch_to(batch, dev)
meter = MeanEPE("verts", device=dev)
with torch.no_grad():
    outs = []
    for _ in range(3):
        o = head(feat, metas, rj)["all_coords_preds"]
        meter.feed(o[-1, :, 21:], gt[:2]); meter.reduce(); outs.append(o.clone())
torch.cuda.synchronize()
assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.isfinite(torch.tensor(meter.result()))
big = torch.zeros(3, 64, 799, 3, device=dev); big[:, :2] = outs[0]; pdist.all_reduce_sum_(big); assert torch.equal(big[:, :2], outs[0])
pdist.barrier()
pdist.shutdown()                                                 # dist.destroy_process_group() (scripts/eval.py:105 upstream)
assert not pdist.active()
print("RCCL1_OK", local_val)
'''


@pytest.mark.gpu
def test_rccl_one_rank_group_runs_every_collective_of_the_path(tmp_path):
    """torch.distributed.run --nproc-per-node=1 with the DEFAULT backend (nccl == RCCL): communicator creation with device_id=,
    MeanEPE / PAEval / PCK reduces on device fp64 / int64 tensors, max-reduce, barrier, a head step with the metric all-reduce,
    destroy, clean exit."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    out = _launch(1, [str(script), ROOT], _env(POEM_DIST_FORCE_INIT="1"), 900)
    assert out.returncode == 0, (out.stdout[-1500:], _err(out))
    assert "RCCL1_OK" in out.stdout


@pytest.mark.gpu
def test_bench_line_over_a_one_rank_rccl_group():
    """bench.py as the driver launches it (torch.distributed.run), one rank, nccl: the branches only a process group reaches --
    max-over-ranks of the time, scaling_diagnostics with the all-reduce latency, the c5 global batch with the shard join --
    execute over RCCL."""
    out = _launch(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "8", "--rotate", "2",
                      "--cpu-samples", "0", "--no-e2e", "--no-extra-configs"], _env(POEM_DIST_FORCE_INIT="1"), 1200)
    res = _bench_line(out)
    assert res["n_gpus"] == 1 and res["config"]["process_group"] == "nccl world_size=1" and res["value"] > 0
    sd = res["scaling_diagnostics"]
    assert len(sd["ms_per_step_by_rank"]) == 1 and sd["allreduce_16B_latency_us"] > 0
    c5 = res["c5_global_ragged_batch64"]
    assert "error" not in c5, c5
    assert c5["samples_by_rank"] == [64] and c5["sharded_bit_equal_to_single_process"] is True and c5["value"] > 0
    assert res["ms_per_step_min"] <= res["ms_per_step_median"] <= res["ms_per_step_max"]


@pytest.mark.gpu
def test_bench_gpus_8_rehearsal_shards_are_bit_equal_to_one_process():
    """`bench.py --gpus 8` end to end with eight ranks (gloo, all on cuda:0 -- RCCL refuses two ranks on one device): all eight
    join, every rank reports its step time, and c5's global batch of 64 -- shards of 7..9 samples -- joined across the
    processes is bit-equal to rank 0 running the 64 samples in one forward."""
    if torch.cuda.device_count() >= 8:
        env = _env()                      # a real node: the default path (nccl, one GPU per rank)
    else:
        env = _env(POEM_SINGLE_DEVICE="1")
    env["OMP_NUM_THREADS"] = "4"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--rotate", "2", "--cpu-samples", "0", "--no-e2e", "--no-extra-configs"],
                         capture_output=True, text=True, env=env, timeout=1800)
    res = _bench_line(out)
    assert res["n_gpus"] == 8 and res["config"]["ranks_joined"] == 8 and "world_size=8" in res["config"]["process_group"]
    assert res["scaling"] == "weak" and res["value"] > 0
    sd = res["scaling_diagnostics"]
    assert len(sd["ms_per_step_by_rank"]) == 8 and all(v > 0 for v in sd["ms_per_step_by_rank"]) and sd["allreduce_16B_latency_us"] > 0
    c5 = res["c5_global_ragged_batch64"]
    assert "error" not in c5, c5
    assert sum(c5["samples_by_rank"]) == 64 and min(c5["samples_by_rank"]) >= 7 and max(c5["samples_by_rank"]) <= 9, c5
    assert c5["sharded_bit_equal_to_single_process"] is True, c5
    assert c5["scaling"] == "strong" and c5["value"] > 0
