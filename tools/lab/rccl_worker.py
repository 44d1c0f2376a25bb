
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
