import sys, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import poem_oracle as po
import poem_v2_amd as pk
hip = pk.hip
for NS in (1024, 4096):
    mode = "cluster"
    g = torch.Generator().manual_seed(NS + len(mode))
    B, NQ = 2, 70
    qx = torch.rand(B, NQ, 3, generator=g) * 2 - 1
    sx = torch.rand(B, NS, 3, generator=g) * 2 - 1
    sx[:, : NS // 2] = sx[:, :1] + 1e-4 * torch.randn(B, NS // 2, 3, generator=g)
    qx[:, :8] = sx[:, :8]
    ref = po.knn_indices(qx, sx, 32)
    got = hip.knn(qx.cuda(), sx.cuda()).cpu().long()
    d = qx[:, :, None, :] - sx[:, None, :, :]
    d = d*d
    dist = ((d[...,0]+d[...,1])+d[...,2])
    for b in range(B):
        for q in range(NQ):
            if not torch.equal(ref[b,q], got[b,q]):
                pos = (ref[b,q] != got[b,q]).nonzero().flatten().tolist()
                print(NS, b, q, "first diff at", pos[:6], "ref", ref[b,q,pos[:4]].tolist(), "got", got[b,q,pos[:4]].tolist(),
                      "dref", dist[b,q,ref[b,q,pos[:4]]].tolist(), "dgot", dist[b,q,got[b,q,pos[:4]]].tolist())
    print(NS, "done")
