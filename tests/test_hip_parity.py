"""Parity of the HIP path (through the C ABI of libpoem_hip.so) against the CPU oracle and the golden vectors captured
from the reference.  All tests need a real MI355X: run with ``-m gpu``."""
import math

import numpy as np
import pytest
import torch

import poem_oracle as po
import poem_v2_amd as pk
from poem_v2_amd import hip
from util import batch_to, build_hip_head, case_setup, load_golden, oracle_consts, run_oracle, stage_report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _md(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests must run on the GPU box"
    hip.lib()   # raises if libpoem_hip.so is missing -- never silently fall back


@pytest.mark.parametrize("M,N,K", [(64, 32, 8), (799, 256, 256), (1000, 128, 256), (130, 16, 48), (4096, 1024, 256),
                                   (257, 256, 1024), (33, 96, 160),
                                   # the K-slab kernel's shapes (K >= 512 in 128-deep slabs, 64-column blocks, M >= 512): POEM-huge's
                                   # Linears, ragged last row block, one and two row tiles per wave
                                   (6392, 1024, 1024), (700, 128, 512), (6392, 64, 4096), (1000, 2048, 1024), (33000, 1024, 1024)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_matches_torch(M, N, K, act):
    g = torch.Generator().manual_seed(M * 7 + N + K + act)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = [ref, torch.relu(ref), torch.nn.functional.gelu(ref)][act] + r.double()
    wp = hip.pack_linear(w.to(DEV))
    y = hip.gemm(x.to(DEV), wp, N, bias=b.to(DEV), residual=r.to(DEV), act=act)
    assert _md(y, ref) < 2e-5
    y2 = hip.gemm(x.to(DEV), wp, N, act=0)           # no bias / residual
    assert _md(y2, torch.nn.functional.linear(x.double(), w.double())) < 2e-5


@pytest.mark.parametrize("M,N,K", [(799, 256, 256), (1000, 128, 256), (4096, 1536, 256), (33, 96, 160), (25568, 768, 256),
                                   (64, 32, 16), (300, 64, 512)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_split_matches_torch(M, N, K, act):
    """Operator level of the opt-in split-precision panel GEMM (f16 hi/lo splits, fp32 accumulation): same tolerance as
    the exact GEMM above, incl. rows of very different weight magnitude (per-tile scales) and small / large activations."""
    g = torch.Generator().manual_seed(M * 7 + N + K + act)
    x = torch.randn(M, K, generator=g) * torch.logspace(-3, 1, K)[None, :].clamp(max=3.0)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    w[: N // 2] *= 1e-3                                     # tiles three orders of magnitude apart
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = [ref, torch.relu(ref), torch.nn.functional.gelu(ref)][act] + r.double()
    y = hip.gemm_split(x.to(DEV), w.to(DEV), bias=b.to(DEV), residual=r.to(DEV), act=act)
    wp = hip.pack_linear(w.to(DEV))
    ye = hip.gemm(x.to(DEV), wp, N, bias=b.to(DEV), residual=r.to(DEV), act=act)
    e_split, e_exact = _md(y, ref), _md(ye, ref)
    assert e_split < 2e-5 and e_split < 4 * e_exact + 2e-6, (e_split, e_exact)
    y2 = hip.gemm_split(x.to(DEV), w.to(DEV))
    assert _md(y2, torch.nn.functional.linear(x.double(), w.double())) < 2e-5


@pytest.mark.parametrize("M,N,K", [(6392, 1024, 1024), (777, 192, 640), (16500, 1024, 512)])
def test_kslab_gemm_is_bit_identical_to_the_operands_from_l2_gemm(M, N, K):
    """The K-slab kernel (weights staged through LDS slab by slab) accumulates every output element over k in the same order
    as gemm2_kernel (operands from L2): the two must agree bit for bit, bias / activation / residual included."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b, r = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    wp = hip.pack_linear(w)
    for act in (0, 1, 2):
        a = hip.gemm(x, wp, N, bias=b, residual=r, act=act)                        # K-slab (default dispatch)
        c = hip.gemm_ex(x, wp, M, N, K, bias=b, residual=r, act=act)               # gemm2_kernel
        assert torch.equal(a, c), act


def test_gemm_is_exact_fma_chain_on_small_integers():
    # integer-valued operands: every partial sum is exactly representable -> bit-exact, catches any k/lane mix-up
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-4, 5, (200, 64), generator=g).float()
    w = torch.randint(-4, 5, (96, 64), generator=g).float()
    y = hip.gemm(x.to(DEV), hip.pack_linear(w.to(DEV)), 96)
    assert torch.equal(y.cpu(), x @ w.t())


def test_layernorm():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(799, 256, generator=g) * 3 + 1
    gm, bt = torch.randn(256, generator=g), torch.randn(256, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (256,), gm.double(), bt.double(), 1e-12)
    assert _md(hip.layernorm(x.to(DEV), gm.to(DEV), bt.to(DEV), 1e-12), ref) < 1e-5


@pytest.mark.parametrize("C,heads,NQ,NK", [(32, 4, 100, 64), (128, 4, 799, 1024), (256, 4, 799, 4096), (512, 4, 257, 512),
                                           (512, 4, 799, 4096), (1024, 4, 130, 256), (1024, 4, 799, 4096)])
def test_cross_attention(C, heads, NQ, NK):
    g = torch.Generator().manual_seed(C + NQ)
    B = 2
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (NQ, NK, NK))
    q = q * 2.0   # sharpen the softmax so the online rescaling path is exercised
    dh = C // heads
    sp = lambda t: t.double().view(B, -1, heads, dh).permute(0, 2, 1, 3)   # noqa: E731
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh)
    ref = (torch.softmax(s, -1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, NQ, C)
    out = hip.cross_attention(q.to(DEV), k.to(DEV), v.to(DEV), heads)
    assert _md(out, ref) < 2e-5


@pytest.mark.parametrize("C,heads,NQ,NK", [(128, 4, 799, 1024), (256, 4, 799, 4096), (256, 4, 33, 64), (64, 2, 100, 256)])
def test_cross_attention_split(C, heads, NQ, NK):
    """Operator level of the opt-in split-precision cross attention (head dims 32 / 64; both contractions as f16 hi/lo
    splits from the same fp32 fragment images, fp32 accumulation and softmax): the exact kernel's tolerance, incl. a
    spiked key that forces the rescale branch."""
    g = torch.Generator().manual_seed(C + NQ + 1)
    B = 2
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (NQ, NK, NK))
    q = q * 2.0
    k[0, NK - 7] = q[0, 5] * 3
    dh = C // heads
    sp = lambda t: t.double().view(B, -1, heads, dh).permute(0, 2, 1, 3)   # noqa: E731
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh)
    ref = (torch.softmax(s, -1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, NQ, C)
    exact = hip.cross_attention(q.to(DEV), k.to(DEV), v.to(DEV), heads)
    out = hip.cross_attention(q.to(DEV), k.to(DEV), v.to(DEV), heads, split=True)
    assert not torch.equal(out, exact)
    e_split, e_exact = _md(out, ref), _md(exact, ref)
    assert e_split < 2e-5 and e_split < 4 * e_exact + 2e-6, (e_split, e_exact)


def test_cross_attention_split_needs_head_dim_32_or_64():
    q = torch.randn(1, 40, 512, device=DEV)
    k = torch.randn(1, 64, 512, device=DEV)
    with pytest.raises(RuntimeError):
        hip.cross_attention(q, k, k, 4, split=True)            # head dim 128: exact kernels only


@pytest.mark.parametrize("B,NQ", [(1, 799), (3, 799), (2, 40), (5, 97)])
def test_cross_attention_merged_in_kernel_equals_partials_and_combine(B, NQ):
    """attn.hip xattn_kernel MERGE (option "xattn_merge"): the four key chunks of a query tile on four waves of one block,
    merged through LDS -- against the partials-in-HBM + attn_combine form BIT for bit (same merge arithmetic, chunk order),
    at the head's shape (4096 keys, head dim 64), several batch sizes / query counts incl. a partial last query tile and
    blocks whose wave groups run out of items at different times; a spiked late key exercises the lazy stabiliser; shapes
    the merged kernel does not take say so."""
    g = torch.Generator().manual_seed(B * 1000 + NQ)
    NK, C, heads = 4096, 256, 4
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (NQ, NK, NK))
    k[0, 3000] = q[0, 5] * 4
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    a = hip.cross_attention(qd, kd, vd, heads)
    b = hip.cross_attention(qd, kd, vd, heads, merged=True)
    assert torch.equal(a, b)
    dh = C // heads
    sp = lambda t: t.double().view(B, -1, heads, dh).permute(0, 2, 1, 3)   # noqa: E731
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh), -1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, NQ, C)
    assert _md(b, ref) < 2e-5
    with pytest.raises(RuntimeError):
        hip.cross_attention(qd, kd[:, :1024].contiguous(), vd[:, :1024].contiguous(), heads, merged=True)   # one chunk: not taken


@pytest.mark.parametrize("B,NQ", [(32, 799), (16, 799), (8, 799), (24, 799), (12, 799), (32, 130)])
def test_cross_attention_remainder_items_as_halves_are_bit_identical(B, NQ):
    """attn.hip xattn_half_item (round 6; option "xattn_tail"): the remainder items of a launch -- the 13th item of half the SIMDs at
    the headline batch -- run as two channel-tile halves on different SIMDs.  Every output element of a partial is the same fma
    chain over the keys: the context rows equal the all-full-items run BIT for bit.  Batches whose CU-pair ranges leave a
    remainder of 1 .. 4 items (halves) and of more (12 samples: none), a query count with another tile count, a spiked key."""
    g = torch.Generator().manual_seed(B * 11 + NQ)
    NK, C, heads = 4096, 256, 4
    q = torch.randn(B, NQ, C, generator=g) * 2.0
    k, v = torch.randn(B, NK, C, generator=g), torch.randn(B, NK, C, generator=g)
    k[B - 1, 3000] = q[B - 1, 5] * 4
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    L = hip.lib()
    try:
        L.poem_cross_attention_tail_halves(0)
        full = hip.cross_attention(qd, kd, vd, heads)
    finally:
        L.poem_cross_attention_tail_halves(1)
    halves = hip.cross_attention(qd, kd, vd, heads)
    assert torch.equal(full, halves)
    sl = slice(B - 1, B)      # (the fp64 reference on one sample)
    dh = C // heads
    sp = lambda t: t[sl].double().view(1, -1, heads, dh).permute(0, 2, 1, 3)   # noqa: E731
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh), -1) @ sp(v)).permute(0, 2, 1, 3).reshape(1, NQ, C)
    assert _md(halves[sl], ref) < 2e-5


def test_cross_attention_spiked_key_forces_rescale():
    # one key dominates late in the sequence: the running max jumps -> alpha-rescale branch must be right
    g = torch.Generator().manual_seed(9)
    B, NQ, NK, C, heads = 1, 64, 256, 64, 4
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (NQ, NK, NK))
    k[0, 200] = q[0, 5] * 6
    dh = C // heads
    sp = lambda t: t.double().view(B, -1, heads, dh).permute(0, 2, 1, 3)   # noqa: E731
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh), -1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, NQ, C)
    assert _md(hip.cross_attention(q.to(DEV), k.to(DEV), v.to(DEV), heads), ref) < 2e-5


@pytest.mark.parametrize("NQ,NS", [(799, 799), (799, 4096), (799, 1024), (40, 33)])
def test_knn_matches_oracle(NQ, NS):
    g = torch.Generator().manual_seed(NS)
    B = 2
    qx = torch.rand(B, NQ, 3, generator=g) * 2 - 1
    sx = qx.clone() if NQ == NS else torch.rand(B, NS, 3, generator=g) * 2 - 1
    ref = po.knn_indices(qx, sx, 32)
    got = hip.knn(qx.to(DEV), sx.to(DEV)).cpu().long()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,NQ,NS", [(300, 40, 64), (1, 5, 4096), (7, 799, 130)])
def test_knn_block_shapes(B, NQ, NS):
    """knn.hip sizes its blocks to the batch (one block per CU over the batch, a block = a query range of one sample, waves
    pull queries from an LDS counter): more samples than CUs (one block per sample), fewer queries than waves, a source
    count that is no multiple of the 128-candidate step."""
    g = torch.Generator().manual_seed(B + NQ + NS)
    qx = torch.rand(B, NQ, 3, generator=g) * 2 - 1
    sx = torch.rand(B, NS, 3, generator=g) * 2 - 1
    assert torch.equal(hip.knn(qx.to(DEV), sx.to(DEV)).cpu().long(), po.knn_indices(qx, sx, 32))


@pytest.mark.parametrize("NS,mode", [(4096, "same"), (4096, "grid"), (799, "grid"), (1024, "cluster"), (4096, "cluster"),
                                     (300, "nan")])
def test_knn_selection_paths(NS, mode):
    """The threshold-selection fast path of knn.hip and its fall-back (more than 128 candidates tie at or below the
    bound; fewer than 32 finite distances) must give the extraction loop's answer: (distance, index) order."""
    g = torch.Generator().manual_seed(NS + len(mode))
    B, NQ = 2, 70
    qx = torch.rand(B, NQ, 3, generator=g) * 2 - 1
    if mode == "same":
        sx = torch.zeros(B, NS, 3)                               # every distance equal: 4096 survivors -> fall-back
    elif mode == "grid":
        sx = torch.round((torch.rand(B, NS, 3, generator=g) * 2 - 1) * 2) / 2      # 5^3 lattice: ties by the hundred
        qx = torch.round(qx * 2) / 2
    elif mode == "cluster":
        sx = torch.rand(B, NS, 3, generator=g) * 2 - 1
        sx[:, : NS // 2] = sx[:, :1] + 1e-4 * torch.randn(B, NS // 2, 3, generator=g)   # half the set in one lane's reach
        qx[:, :8] = sx[:, :8]
    else:
        sx = torch.rand(B, NS, 3, generator=g) * 2 - 1
        sx[:, 20:] = float("nan")                                # 20 finite candidates < 32
    got = hip.knn(qx.to(DEV), sx.to(DEV)).cpu().long()
    if mode == "nan":
        ref = po.knn_indices(qx, sx[:, :20], 20)
        assert torch.equal(got[..., :20], ref)
        return
    ref = po.knn_indices(qx, sx, 32)
    assert torch.equal(got, ref)


def test_knn_both_roundings_match_the_oracle_on_near_tied_points():
    """poem_knn / poem_knn_ex(fma_contract): the CPU kernel's rounding ((dx*dx + dy*dy) + dz*dz) and the CUDA kernel's
    (fma(dz, dz, fma(dy, dy, dx*dx))) of pytorch3d's accumulation loop, each against the oracle's restatement of the same
    rounding, INDEX for index -- on points where thousands of candidates sit within an ulp of each other, so that one
    mis-rounded product anywhere would show; and the two roundings must not give the same answer there."""
    from test_oracle_golden import near_tie_points
    q, s = near_tie_points()
    got0 = hip.knn(q.to(DEV), s.to(DEV)).cpu().long()
    got1 = hip.knn(q.to(DEV), s.to(DEV), fma=True).cpu().long()
    assert torch.equal(got0, po.knn_indices(q, s, 32, False))
    assert torch.equal(got1, po.knn_indices(q, s, 32, True))
    assert not torch.equal(got0, got1)
    g = torch.Generator().manual_seed(7)                           # and on ordinary points
    q, s = torch.rand(2, 799, 3, generator=g) * 2 - 1, torch.rand(2, 4096, 3, generator=g) * 2 - 1
    assert torch.equal(hip.knn(q.to(DEV), s.to(DEV), fma=True).cpu().long(), po.knn_indices(q, s, 32, True))


def test_knn_ties_take_lower_index():
    sx = torch.zeros(1, 64, 3)
    sx[0, :, 0] = torch.arange(64).float() // 2          # every distance appears twice
    qx = torch.zeros(1, 1, 3)
    got = hip.knn(qx.to(DEV), sx.to(DEV)).cpu().long()
    assert got[0, 0].tolist() == list(range(32))


@pytest.mark.parametrize("C", [32, 128, 256, 512])
@pytest.mark.parametrize("anchors", [False, True])
def test_vector_attention_core(C, anchors):
    g = torch.Generator().manual_seed(C + anchors)
    B, Q, NS = 2, 101, 300
    qxyz = torch.rand(B, Q, 3, generator=g) * 2 - 1
    sxyz = torch.rand(B, NS, 3, generator=g) * 2 - 1
    q, k, v = torch.randn(B, Q, C, generator=g), torch.randn(B, NS, C, generator=g), torch.randn(B, NS, C, generator=g)
    w = {}
    for n, shp in (("fc_delta.0", (C, 3)), ("fc_delta.2", (C, C)), ("fc_gamma.0", (C, C)), ("fc_gamma.2", (C, C))):
        w["p." + n + ".weight"] = torch.randn(*shp, generator=g) / math.sqrt(shp[1])
        w["p." + n + ".bias"] = torch.randn(shp[0], generator=g) * 0.1
    if anchors:
        idx = torch.randperm(NS, generator=g)[:32]
        axyz = torch.rand(32, 3, generator=g) * 2 - 1
        idx_full = idx.view(1, 1, 32).expand(B, Q, 32)
        nxyz = axyz.view(1, 1, 32, 3).expand(B, Q, 32, 3)
    else:
        idx_full = po.knn_indices(qxyz, sxyz, 32)
        nxyz = po.gather_xyz(sxyz, idx_full)
    ref = po._vec_attn_core(w, "p.", q, po.index_points(k, idx_full), po.index_points(v, idx_full),
                            qxyz[:, :, None] - nxyz, C)
    d = lambda t: t.to(DEV).contiguous()   # noqa: E731
    out = hip.vector_attention(d(qxyz), d(sxyz), d(axyz) if anchors else None,
                               d(idx.int()) if anchors else d(idx_full.int()), d(q), d(k), d(v),
                               d(w["p.fc_delta.0.weight"]), d(w["p.fc_delta.0.bias"]),
                               hip.pack_linear(d(w["p.fc_delta.2.weight"])), d(w["p.fc_delta.2.bias"]),
                               hip.pack_linear(d(w["p.fc_gamma.0.weight"])), d(w["p.fc_gamma.0.bias"]),
                               hip.pack_linear(d(w["p.fc_gamma.2.weight"])), d(w["p.fc_gamma.2.bias"]))
    assert _md(out, ref) < 5e-5


@pytest.mark.parametrize("C", [128, 256, 512])
@pytest.mark.parametrize("anchors", [False, True])
def test_vector_attention_split_core(C, anchors):
    """Operator level: the opt-in split-precision kernel (composed form, f16 hi/lo splits, fp32 accumulation) against the
    oracle's fp32 vector attention -- same tolerance as the exact kernel's test above."""
    g = torch.Generator().manual_seed(7 * C + anchors)
    B, Q, NS = 2, 101, 300
    qxyz = torch.rand(B, Q, 3, generator=g) * 2 - 1
    sxyz = torch.rand(B, NS, 3, generator=g) * 2 - 1
    q, k, v = torch.randn(B, Q, C, generator=g), torch.randn(B, NS, C, generator=g), torch.randn(B, NS, C, generator=g)
    w = {}
    for n, shp in (("fc_delta.0", (C, 3)), ("fc_delta.2", (C, C)), ("fc_gamma.0", (C, C)), ("fc_gamma.2", (C, C))):
        w["p." + n + ".weight"] = torch.randn(*shp, generator=g) / math.sqrt(shp[1])
        w["p." + n + ".bias"] = torch.randn(shp[0], generator=g) * 0.1
    if anchors:
        idx = torch.randperm(NS, generator=g)[:32]
        axyz = torch.rand(32, 3, generator=g) * 2 - 1
        idx_full = idx.view(1, 1, 32).expand(B, Q, 32)
        nxyz = axyz.view(1, 1, 32, 3).expand(B, Q, 32, 3)
    else:
        idx_full = po.knn_indices(qxyz, sxyz, 32)
        nxyz = po.gather_xyz(sxyz, idx_full)
    ref = po._vec_attn_core(w, "p.", q, po.index_points(k, idx_full), po.index_points(v, idx_full),
                            qxyz[:, :, None] - nxyz, C)
    Wg1, Wd2 = w["p.fc_gamma.0.weight"].double(), w["p.fc_delta.2.weight"].double()
    cvec = Wg1 @ w["p.fc_delta.2.bias"].double() + w["p.fc_gamma.0.bias"].double()
    qg = (q.double() @ Wg1.T + cvec).float()
    kg = (k.double() @ Wg1.T).float()
    d = lambda t: t.to(DEV).contiguous()   # noqa: E731
    i1, s1 = hip.pack_split_linear(d(Wd2.float()))
    i2, s2 = hip.pack_split_linear(d((Wg1 @ Wd2).float()))
    i3, s3 = hip.pack_split_linear(d(w["p.fc_gamma.2.weight"]))
    for sc, W in ((s1, Wd2), (s2, Wg1 @ Wd2), (s3, w["p.fc_gamma.2.weight"])):
        m = float(W.abs().max()) * float(sc)
        assert 8.0 <= m < 16.0 and math.log2(float(sc)) == int(math.log2(float(sc)))     # power of two, max |w'| in [8,16)
    out = hip.vector_attention_split(d(qxyz), d(sxyz), d(axyz) if anchors else None,
                                     d(idx.int()) if anchors else d(idx_full.int()), d(qg), d(kg), d(v),
                                     d(w["p.fc_delta.0.weight"]), d(w["p.fc_delta.0.bias"]), i1, d(w["p.fc_delta.2.bias"]),
                                     i2, i3, torch.cat([s1, s2, s3]))
    assert _md(out, ref) < 5e-5


def test_project_and_sample_matches_oracle():
    _, meta = load_golden("tiny")
    cfg, w, consts, batch = case_setup(meta["spec"])
    g = torch.Generator().manual_seed(1)
    views = meta["spec"]["views"]
    BN = sum(views)
    x = torch.randn(BN, 32, 16, 16, generator=g)
    m = batch["img_metas"]
    centre = batch["reference_joints"][:, 9].contiguous()
    vs = torch.repeat_interleave(torch.arange(len(views)), torch.tensor(views))
    uv = po.project_points(consts["bps"][None] + centre[:, None], m["cam_intr"], m["cam_extr"], vs)
    ref = po.grid_sample_bilinear(x, uv * (1.0 / 256.0) * 2 - 1)
    got, _ = hip.project_sample(x.to(DEV), consts["bps"].to(DEV), centre.to(DEV), vs.int().to(DEV), m["cam_intr"].to(DEV),
                                m["cam_extr"].to(DEV), (256, 256))
    assert _md(got, ref) < 2e-5


@pytest.mark.parametrize("lid,depth_num,fh,fw,img", [(False, 32, 16, 16, (256, 256)), (True, 16, 16, 16, (256, 256)), (False, 8, 8, 24, (384, 128))])
def test_frustum_features_match_the_oracle(lid, depth_num, fh, fw, img):
    """poem_frustum_features (the input of position_encoder, ptEmb_head.py:113-181 + inverse_sigmoid) against the oracle's
    restatement, which tests/test_oracle_golden.py pins to the reference through the tinypetr / tinypetrlid / mediumpetr
    fixtures.  The 4 x 4 product and logf round differently from the CPU's (1 ulp of a coordinate of magnitude ~1, i.e. ~1e-7 of
    the normalised coordinate t) and d/dt log(t / (1 - t)) = 1 / (t (1 - t)) grows towards the clamps: compared as t
    (3e-7) everywhere and as values (2e-5) where |value| < 5; the clamped entries (|v| = log(1e5)) sit at the same places."""
    views = [3, 1, 4]
    b = pk.inputs.synthetic_batch(views, seed=5)
    m = b["img_metas"]
    ocfg = po.PathConfig(petr=True, lid=lid, depth_num=depth_num, depth_start=0.05 if lid else 0.0, depth_end=1.3,
                         position_range=(-0.5, -0.7, 0.1, 0.7, 0.5, 1.4) if lid else (-0.6, -0.6, 0.0, 0.6, 0.6, 1.2))
    want = po.frustum_features(ocfg, m["cam_intr"], m["cam_extr"], fh, fw, img)
    cfg = hip.make_config(256, feat_h=fh, feat_w=fw, petr_embedding=True, depth_num=depth_num, lid=lid, depth_start=ocfg.depth_start,
                          depth_end=ocfg.depth_end, position_range=ocfg.position_range)
    out = torch.full((sum(views), 3 * depth_num, fh, fw), float("nan"), device=DEV)
    import ctypes
    K, E = m["cam_intr"].to(DEV), m["cam_extr"].to(DEV)
    hip.check(hip.lib().poem_frustum_features(ctypes.byref(cfg), hip.ptr(K), hip.ptr(E), sum(views), img[0], img[1], hip.ptr(out),
                                              hip.stream()), "poem_frustum_features")
    got = out.cpu()
    assert bool(torch.isfinite(got).all())
    clamp = math.log(1e5)
    edge = (want.abs() > clamp - 1e-3)
    assert 0.02 < float(edge.float().mean()) < 0.8           # the case has both saturated and interior points
    assert _md(torch.sigmoid(got.double()), torch.sigmoid(want.double())) < 3e-7
    mid = want.abs() < 5.0
    assert float(mid.float().mean()) > 0.2 and _md(got[mid], want[mid]) < 2e-5
    assert float(((got.abs() > clamp - 1e-3) != edge).float().mean()) < 1e-4


@pytest.mark.parametrize("name", ["tiny", "tinymano", "tinypetr", "tinynonorm", "tinypetrlid", "tinyk", "tinycfg2", "tinycfg4"])
def test_head_tiny_stage_taps_vs_golden(name):
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    head._engine_for(torch.device(DEV)).enable_taps(True)
    with torch.no_grad():
        out = head(feat, metas, rj)
    eng = head._engine
    B, BN, C, S, Q = len(spec["views"]), sum(spec["views"]), spec["embed"], spec["nsample"], 799
    assert _md(eng.tap("x", (BN, C, 16, 16)), torch.from_numpy(z["tap.x"])) < 3e-5
    assert _md(eng.tap("g", (BN, C, S)), torch.from_numpy(z["tap.g"])) < 3e-5
    assert _md(eng.tap("bps_feat", (B, S, C)), torch.from_numpy(z["tap.bps_feat"])) < 1e-4
    assert _md(eng.tap("pt_xyz", (B, S, 3)), torch.from_numpy(z["tap.pt_xyz"])) == 0.0
    assert _md(eng.tap("query_xyz", (B, Q, 3)), torch.from_numpy(z["tap.query_xyz"])) == 0.0
    for i in range(spec.get("nblocks", 3)):
        for k, tol in (("h_cross", 5e-5), ("f_self", 5e-5), ("f_cross", 5e-5), ("feats", 1e-4)):
            assert _md(eng.tap(f"b{i}.{k}", (B, Q, C))[:, ::9], torch.from_numpy(z[f"tap.b{i}.{k}"])) < tol, (i, k)
        assert _md(eng.tap(f"b{i}.xyz", (B, Q, 3)), torch.from_numpy(z[f"tap.b{i}.xyz"])) < 5e-5, i
    ref = torch.from_numpy(z["all_coords_preds"])
    got = out["all_coords_preds"].cpu()
    assert got.shape == ref.shape
    assert _md(got, ref) < 5e-6                                                   # metres
    if spec["parametric"]:
        assert _md(out["pred_pose"], torch.from_numpy(z["pred_pose"])) < 2e-4
        assert _md(out["pred_shape"], torch.from_numpy(z["pred_shape"])) < 2e-5


@pytest.mark.parametrize("name", ["small", "medium", "large", "huge", "ragged", "mediummano", "mediumpetr", "mediumk"])
def test_head_release_shapes_vs_golden_and_oracle(name):
    """BASELINE.json bar: MPVPE of the HIP path vs the reference <= 1e-3 mm (1e-6 m), last decoder layer."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        res = head(feat, metas, rj)
    got = res["all_coords_preds"].cpu()
    if spec["parametric"]:
        assert _md(res["pred_pose"], torch.from_numpy(z["pred_pose"])) < 2e-4
        assert _md(res["pred_shape"], torch.from_numpy(z["pred_shape"])) < 2e-5
    ref = torch.from_numpy(z["all_coords_preds"])
    mpvpe = torch.norm(got[-1, :, 21:] - ref[-1, :, 21:], dim=-1).mean(dim=1)        # metres, per sample
    assert float(mpvpe.max()) < 1e-6, mpvpe
    orc = run_oracle(cfg, w, consts, batch)["all_coords_preds"]
    mp2 = torch.norm(got[-1, :, 21:] - orc[-1, :, 21:], dim=-1).mean(dim=1)
    assert float(mp2.max()) < 1e-6, mp2
    assert _md(got, ref) < 1e-4


@pytest.mark.parametrize("name", ["small", "medium", "large", "huge", "ragged", "small_hot", "medium_hot", "large_hot"])
def test_release_shape_stage_taps_vs_reference(name):
    """Default fp32 path at the RELEASE shapes, stage by stage against the reference's own per-block tensors (fixture taps),
    incl. the hot-weight cases (block Linears x2.5: coordinate updates and neighbour changes are O(1)):

    * every block's h_cross / f_self / f_cross / feats / xyz on the rows whose neighbour sets are the reference's: within a
      scale-relative fp32 tolerance (1e-5 of the tensor's max magnitude ~ 80 ulp) and within 6x the CPU restatement's own
      distance from the reference (+ 2 ulp of the scale) -- i.e. the path sits inside the reference's own re-ordering noise;
    * neighbour sets of blocks 1, 2 reported on their own: >= 99.5 % of the queries pick the reference's set, and every
      query that does not is *attributed*: the reference's 32nd and 33rd candidate distances differ by < 1e-5 relative
      (a tie at fp32 round-off, where the reference itself flips with the summation order);
    * MPVPE of HIP-vs-reference next to oracle-vs-reference, every layer: both under the 1e-3 mm bar."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    with torch.no_grad():
        got = head(feat, metas, rj)["all_coords_preds"].cpu()
    otaps = {}
    orc = run_oracle(cfg, w, consts, batch, taps=otaps)["all_coords_preds"]
    rep = stage_report(z, spec, lambda n, shape, dt=torch.float32: eng.tap(n, shape, dt).cpu(), otaps)
    eng.enable_taps(False)
    for key, st in rep["stages"].items():
        assert st["clean_rows"] > 0.9, (key, st)
        assert st["path_clean"] <= 1e-5 * max(st["scale"], 1.0), (key, st)
        assert st["path_clean"] <= 6 * st["oracle_clean"] + 2.4e-7 * max(st["scale"], 1.0), (key, st)
    for key, nb in rep["neighbours"].items():
        assert nb["set_equal"] >= 0.995, (key, nb)
        for b, q, gap in nb["flips"]:
            assert gap < 1e-5, (key, b, q, gap)
    ref = torch.from_numpy(z["all_coords_preds"])
    for layer in range(3):
        mp_hip = torch.norm(got[layer, :, 21:] - ref[layer, :, 21:], dim=-1).mean(dim=1)
        mp_orc = torch.norm(orc[layer, :, 21:] - ref[layer, :, 21:], dim=-1).mean(dim=1)
        assert float(mp_hip.max()) < 1e-6 and float(mp_orc.max()) < 1e-6, (layer, mp_hip, mp_orc)
    if "hot" in name:
        assert float((ref[0] - ref[1]).abs().max()) > 0.01           # metres: the layers really move the mesh


@pytest.mark.parametrize("name", ["medium_g1", "medium_g4", "medium_g6", "small_tie", "small_tie_fma"])
def test_conditioning_sweep_and_cuda_rounding_vs_reference(name):
    """Round-3 fixtures (tests/test_oracle_golden.py has the CPU side).  medium_g{1,4,6} + medium_hot = the same case at
    gains 1 / 2.5 / 4 / 6 of the block Linears; *_fma = the reference's neighbour search rounding its distances like
    pytorch3d's CUDA kernel.  For the default path (and, on the *_fma cases, with poem_set_option(knn_fma) too):
    neighbour sets >= 99.5 % the reference's, every other set attributed to a near-tie (32nd / 33rd gap < 1e-5; < 1e-3 at
    gain 6, where the coordinates themselves are only known to 1e-5 relative); stages on
    the clean rows within the gain's tolerance and within 6x the CPU restatement's own distance; MPVPE of every layer
    <= max(1e-3 mm, 3 x the CPU restatement's MPVPE) -- past gain 2.5 two correct fp32 evaluations no longer agree to 1e-3 mm
    (profiles/r03_parity.txt has the curve)."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    otaps = {}
    orc = run_oracle(cfg, w, consts, batch, taps=otaps)["all_coords_preds"]
    ref = torch.from_numpy(z["all_coords_preds"])
    gain = spec.get("gain", 1.0)
    stage_tol = 1e-5 if gain <= 2.5 else (3e-5 if gain <= 4 else 1.5e-4)
    flip_tol = 1e-5 if gain <= 4 else 1e-3        # (gain 6: coordinates of hundreds of metres known to ~1e-5 relative -- a "tie" is wider)
    for fma in ([0, 1] if spec.get("knn_fma") else [0]):
        eng.set_option("knn_fma", fma)
        with torch.no_grad():
            got = head(feat, metas, rj)["all_coords_preds"].cpu()
        assert torch.isfinite(got).all()
        rep = stage_report(z, spec, lambda n, shape, dt=torch.float32: eng.tap(n, shape, dt).cpu(), otaps)
        flips = 0
        for key, nb in rep["neighbours"].items():
            assert nb["set_equal"] >= (0.995 if gain <= 4 else 0.99), (fma, key, nb)
            flips += len(nb["flips"])
            if gain > 4 and not key.startswith("b1."):
                continue            # (block 2's coordinates already carry block 1's flips: its gaps are not round-off any more)
            for b, q, gap in nb["flips"]:
                assert gap < flip_tol, (fma, key, b, q, gap)
        # Past gain 4 a flipped neighbour changes its query's features by O(1) and the next block GATHERS those rows: the
        # "clean rows" of a later block are no longer clean, and the output moves by ~1e-2 of its magnitude per flip -- for
        # HIP and for the CPU restatement alike (whichever of them flips).  The stage / MPVPE bars are then only meaningful for
        # a run without flips; the neighbour attribution above is what remains checkable (profiles/r03_parity.txt has the numbers).
        if gain > 4 and flips:
            # ... what IS checkable with flips: everything in front of the first neighbour search (block 0: fixed anchors) on every
            # row, and block 1 on the rows whose block-1 neighbour sets are the reference's -- their inputs are block 0's outputs,
            # which no flip has touched; the first decoder layer's mesh (block 0's coordinates) to the gain's MPVPE bar
            for key, st in rep["stages"].items():
                if key.startswith("b0.") or key.startswith("b1."):
                    assert st["clean_rows"] > 0.9, (key, st)
                    assert st["path_clean"] <= stage_tol * max(st["scale"], 1.0), (fma, key, st)
            mp0_hip = float(torch.norm(got[0, :, 21:] - ref[0, :, 21:], dim=-1).mean(dim=1).max())
            mp0_orc = float(torch.norm(orc[0, :, 21:] - ref[0, :, 21:], dim=-1).mean(dim=1).max())
            assert mp0_hip <= max(1e-6, 3 * mp0_orc), (fma, mp0_hip, mp0_orc)
            continue
        # (the same one step earlier: a run with the OTHER rounding than the fixture's whose near-tie did flip -- small_tie_fma
        #  under the default rounding -- has block-2 rows that gathered the flipped row; stage bars for the matching rounding only)
        mismatched = bool(fma) != bool(spec.get("knn_fma")) and flips > 0
        for key, st in rep["stages"].items():
            if mismatched:
                break
            assert st["clean_rows"] > 0.9, (key, st)
            assert st["path_clean"] <= stage_tol * max(st["scale"], 1.0), (fma, key, st)
            assert st["path_clean"] <= 6 * st["oracle_clean"] + 2.4e-7 * max(st["scale"], 1.0), (fma, key, st)
        for layer in range(3):
            mp_hip = float(torch.norm(got[layer, :, 21:] - ref[layer, :, 21:], dim=-1).mean(dim=1).max())
            mp_orc = float(torch.norm(orc[layer, :, 21:] - ref[layer, :, 21:], dim=-1).mean(dim=1).max())
            assert mp_hip <= max(1e-6, 3 * mp_orc), (fma, layer, mp_hip, mp_orc)
            if gain <= 2.5:
                assert mp_hip < 1e-6, (fma, layer, mp_hip)
    eng.set_option("knn_fma", 0)
    eng.enable_taps(False)


def test_tie_pair_poem_knn_reproduces_each_rounding_of_the_reference_run():
    """The pair of reference runs that differ by the third party's distance rounding (tests/test_oracle_golden.py
    test_tie_pair_...): poem_knn (CPU-kernel rounding) / poem_knn_ex(fma_contract) on the coordinates each run's searches saw
    must return that run's recorded neighbours -- every query, no attribution -- and the other rounding must not."""
    from test_oracle_golden import tie_pair_searches
    wrong = 0
    for fma, blk, which, xyz, src, want in tie_pair_searches():
        got = hip.knn(xyz.to(DEV), src.to(DEV), fma=fma).cpu().long()
        assert torch.equal(torch.sort(got, -1).values, torch.sort(want, -1).values), (fma, blk, which)
        assert float((got == want).float().mean()) > 0.999, (fma, blk, which)       # same ORDER too, up to exactly tied distances
        other = hip.knn(xyz.to(DEV), src.to(DEV), fma=not fma).cpu().long()
        wrong += int((torch.sort(other, -1).values != torch.sort(want, -1).values).any(-1).sum())
    assert wrong >= 2, wrong


@pytest.mark.parametrize("name", ["small", "medium", "large", "huge", "ragged", "mediummano"])
def test_anchor_tables_vs_per_sample_form(name):
    """Block 0 (Q2: fixed anchors; query coordinates = the template for every sample): the default path computes the
    positional products of both vector attentions once per forward from t/r (csrc/vecattn.hip MODE 1/2); with
    set_anchor_tables(False) they are evaluated per sample from ((c + t) - c)/r as the reference does.  The two agree to
    round-off (the inputs differ by the rounding of c + t), both meet the 1e-3 mm bar against the reference fixture, and
    the table form stays at the per-sample form's own distance from it."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    ref = torch.from_numpy(z["all_coords_preds"])
    with torch.no_grad():
        tab = head(feat, metas, rj)["all_coords_preds"].cpu()
        head.set_anchor_tables(False)
        per = head(feat, metas, rj)["all_coords_preds"].cpu()
        head.set_anchor_tables(True)
        again = head(feat, metas, rj)["all_coords_preds"].cpu()
    assert torch.equal(tab, again)
    assert not torch.equal(tab, per)                                   # the table kernels really ran
    mp = lambda a, b: float(torch.norm(a[-1, :, 21:] - b[-1, :, 21:], dim=-1).mean(dim=1).max())
    d_tab, d_per, d_between = mp(tab, ref), mp(per, ref), mp(tab, per)
    assert d_per < 1e-6 and d_tab < 1e-6, (d_tab, d_per)               # metres: the 1e-3 mm bar, both forms
    assert d_tab < 2 * d_per + 2e-8, (d_tab, d_per)
    assert d_between < 2e-7, d_between
    assert _md(tab, per) < 2e-5
    # each form against the oracle's restatement of the SAME arithmetic
    o_tab = run_oracle(cfg, w, consts, batch, anchor_tables=True)["all_coords_preds"]
    o_per = run_oracle(cfg, w, consts, batch)["all_coords_preds"]
    assert mp(tab, o_tab) < 1e-6 and mp(per, o_per) < 1e-6, (mp(tab, o_tab), mp(per, o_per))


@pytest.mark.parametrize("name", ["small", "medium", "large", "ragged", "mediummano", "medium_hot"])
def test_chain_kernels_on_both_matrix_shapes_are_bit_identical(name):
    """csrc/chain16.hip (v_mfma_f32_16x16x4_f32, row tiles of 1..4 units of 16 rows) against csrc/chain.hip (32x32x2, 32- / 64-row
    tiles): every output element is the same k-ordered fma chain and LayerNorm sums in the same order, so every stage tap and
    the output agree BIT FOR BIT -- which is what lets the tile height follow the batch size."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(case_setup(spec)[3], DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    B, C, Q = len(spec["views"]), spec["embed"], 799
    got = {}
    for tile in (2, 3, 1):
        eng.set_option("chain_tile", tile)
        with torch.no_grad():
            res = head(feat, metas, rj)
        got[tile] = {"out": res["all_coords_preds"].cpu()}
        got[tile].update({f"b{i}.{k}": eng.tap(f"b{i}.{k}", (B, Q, 3 if k == "xyz" else C)).cpu()
                          for i in range(3) for k in ("h_cross", "f_self", "f_cross", "feats", "xyz")})
        if spec["parametric"]:
            got[tile]["pose"] = res["pred_pose"].cpu()
    eng.set_option("chain_tile", 0)
    eng.enable_taps(False)
    for k in got[2]:
        assert torch.equal(got[2][k], got[3][k]), (k, float((got[2][k] - got[3][k]).abs().max()))
        assert torch.equal(got[2][k], got[1][k]), k


@pytest.mark.parametrize("views", [[8], [8, 8], [3, 10, 1, 6, 8, 2], [8] * 12])
def test_round5_scheduling_switches_and_tile_heights_change_no_bit(views):
    """Round 5's small-batch work is scheduling only: `d2_first` / `wait_merge` (which successor of a launch the replayed graph
    keeps on its hardware queue; which launch carries a cross-stream wait) and the chain kernels' one- and two-unit tiles on the
    weight ring over native 16x16x4 images (chain16.hip; batches of 1-12 samples put 1-3 units on a CU, i.e. every ring /
    non-ring tile height).  Graph replays with every switch flipped, the 32-row chain kernels (no ring, no native images) and
    plain launches give the same bits, stage taps included."""
    spec = dict(embed=256, nsample=4096, views=views, seed=55, parametric=False)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(case_setup(spec)[3], DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    B, C, Q = len(views), 256, 799

    def run():
        with torch.no_grad():
            for _ in range(3):                                   # the third forward replays the captured graph
                res = head(feat, metas, rj)
        out = {"out": res["all_coords_preds"].cpu()}
        out.update({f"b{i}.{k}": eng.tap(f"b{i}.{k}", (B, Q, 3 if k == "xyz" else C)).cpu()
                    for i in range(3) for k in ("h_cross", "f_self", "f_cross", "feats", "xyz")})
        return out

    want = run()
    assert bool(torch.isfinite(want["out"]).all())
    settings = [dict(d2_first=0), dict(wait_merge=0), dict(wait_merge=7), dict(d2_first=0, wait_merge=0), dict(chain_tile=1),
                dict(chain_tile=3), dict(graphs=0)]
    defaults = dict(d2_first=1, wait_merge=-1, chain_tile=0, graphs=1)
    for st in settings:
        for k, v in st.items():
            eng.set_option(k, v)
        got = run()
        for k, v in defaults.items():
            eng.set_option(k, v)
        for k in want:
            assert torch.equal(want[k], got[k]), (st, k, float((want[k] - got[k]).abs().max()))
    assert eng.graph_stats()["replays"] >= 1
    eng.enable_taps(False)


@pytest.mark.parametrize("name", ["small", "medium", "large", "ragged", "mediummano", "medium_hot"])
def test_row_tile_chains_vs_operator_launches(name):
    """csrc/chain.hip (default): the query-side Linears / residual adds / LayerNorms of a block as four LDS-resident chain
    launches; set_chains(False) = one launch per operator.  Same arithmetic per element except the summation order inside
    LayerNorm's moments, so the two agree to fp32 round-off at every stage tap and in the output; both meet the bar against
    the reference fixture, and the chain path is the one whose block-0 queries are read in place (no broadcast copy)."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    B, C, Q = len(spec["views"]), spec["embed"], 799
    out, taps = {}, {}
    for mode in (True, False, True):
        head.set_chains(mode)
        with torch.no_grad():
            res = head(feat, metas, rj)
        if mode in out:
            assert torch.equal(out[mode], res["all_coords_preds"].cpu())          # deterministic, and the switch switches back
            continue
        out[mode] = res["all_coords_preds"].cpu()
        taps[mode] = {f"b{i}.{k}": eng.tap(f"b{i}.{k}", (B, Q, 3 if k == "xyz" else C)).cpu()
                      for i in range(3) for k in ("h_cross", "f_self", "f_cross", "feats", "xyz")}
        if spec["parametric"]:
            taps[mode]["pose"] = res["pred_pose"].cpu()
    eng.enable_taps(False)
    assert not torch.equal(out[True], out[False])                                  # the chain kernels really ran
    for k in taps[True]:
        a, b = taps[True][k], taps[False][k]
        scale = max(float(b.abs().max()), 1.0)
        if "hot" in name and not k.startswith("b0"):
            # past the first neighbour search a near-tie may flip a set in one path and not the other: compare the bulk
            d = (a - b).abs().amax(-1).flatten()
            assert float(d.kthvalue(int(0.99 * d.numel())).values) <= 2e-5 * scale, (k, float(d.max()))
        else:
            assert _md(a, b) <= 2e-5 * scale, (k, _md(a, b), scale)
    ref = torch.from_numpy(z["all_coords_preds"])
    mp = lambda a, b: float(torch.norm(a[-1, :, 21:] - b[-1, :, 21:], dim=-1).mean(dim=1).max())
    assert mp(out[True], ref) < 1e-6 and mp(out[False], ref) < 1e-6, (mp(out[True], ref), mp(out[False], ref))
    assert mp(out[True], out[False]) < (1e-6 if "hot" in name else 2e-7)


@pytest.mark.parametrize("name", ["small", "medium", "large", "ragged", "mediummano", "medium_hot"])
def test_fused_sampling_vs_operator_sequence(name):
    """csrc/merge.hip (default): bilinear sampling + the Q1 re-interpretation + the merge MLP in two kernels (the sampled
    tensor g and merge_net[0]'s hidden layer never reach HBM); set_option("fused_sampling", 0) = sample.hip + gemm.hip, one
    launch per operator.  Same fma chain per sampled value and per Linear output; only the reduction order of the cross-view
    dot products differs.  Both forms: `bps_feat` against the REFERENCE's own tensor (fixture tap, every 64th basis point,
    views [2] .. [3,10,1,6] incl. the single-view branch), against each other, and the final vertices against the fixture."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    B, C, S = len(spec["views"]), spec["embed"], spec["nsample"]
    out, bf = {}, {}
    for mode in (1, 0, 1):
        eng.set_option("fused_sampling", mode)
        with torch.no_grad():
            res = head(feat, metas, rj)["all_coords_preds"].cpu()
        if mode in out:
            assert torch.equal(out[mode], res)
            continue
        out[mode] = res
        bf[mode] = eng.tap("bps_feat", (B, S, C)).cpu()
        x = eng.tap("x", (sum(spec["views"]), C, 16, 16)).cpu()
        if mode == 1:
            x_fused = x
            with pytest.raises(RuntimeError):
                eng.tap("g", (1,))                     # the sampled tensor does not exist in the fused form
        else:
            assert torch.equal(x, x_fused)             # input_proj writes both layouts from the same registers
    eng.enable_taps(False)
    ref_bf = torch.from_numpy(z["tap.bps_feat"])
    scale = float(ref_bf.abs().max())
    for mode in (1, 0):
        assert _md(bf[mode][:, ::64], ref_bf) < 2e-5 * max(scale, 1.0), (mode, _md(bf[mode][:, ::64], ref_bf), scale)
    assert not torch.equal(bf[1], bf[0])                # the fused kernels really ran
    assert _md(bf[1], bf[0]) < 1e-5 * max(scale, 1.0), (_md(bf[1], bf[0]), scale)
    ref = torch.from_numpy(z["all_coords_preds"])
    mp = lambda a, b: float(torch.norm(a[-1, :, 21:] - b[-1, :, 21:], dim=-1).mean(dim=1).max())
    assert mp(out[1], ref) < 1e-6 and mp(out[0], ref) < 1e-6, (mp(out[1], ref), mp(out[0], ref))
    assert mp(out[1], out[0]) < (1e-6 if "hot" in name else 2e-7)


_DECODER_ORACLE = {}


@pytest.mark.parametrize("embed,chains", [(128, 1), (256, 1), (256, 0), (512, 1)])
def test_decoder_entry_with_caller_queries_vs_oracle(embed, chains):
    """PtEmbedTRv4.forward / poem_decoder_forward (ptEmb_transformer.py:371-376): the decoder on the CALLER's query coordinates
    and features -- per-sample, nothing like the head's shared template, so block 0 takes the per-sample form, the chains read
    their residual rows per sample and no anchor table is used -- against the oracle's three decoder blocks on the same inputs."""
    spec = dict(embed=embed, nsample=4096, views=[2, 2, 2], seed=5, parametric=False)
    cfg, w, consts, _ = case_setup(spec)
    head = build_hip_head(spec, DEV)
    head.set_chains(bool(chains))
    g = torch.Generator().manual_seed(embed)
    B = 3
    qxyz = torch.randn(B, 799, 3, generator=g) * 0.4
    pxyz = torch.rand(B, 4096, 3, generator=g) * 2 - 1
    qf, pf = torch.randn(B, 799, embed, generator=g), torch.randn(B, 4096, embed, generator=g)
    with torch.no_grad():
        got, pose, shape = head.transformer(qxyz.to(DEV), qf.to(DEV), pxyz.to(DEV), pf.to(DEV))
        if embed not in _DECODER_ORACLE:      # (the same inputs for chains = 0 / 1: the oracle's three blocks once per width)
            feats, xyz, ref = qf, qxyz, []
            for i in range(cfg.nblocks):
                feats, xyz, _, _ = po.decoder_block(w, cfg, i, xyz, feats, pxyz, pf, consts)
                ref.append(xyz)
            _DECODER_ORACLE[embed] = torch.stack(ref)
    assert pose is None and shape is None
    ref = _DECODER_ORACLE[embed]
    got = got.cpu()
    assert got.shape == ref.shape == (3, B, 799, 3)
    for layer in range(3):
        d = torch.norm(got[layer] - ref[layer], dim=-1)                 # normalised units (x 0.1 m)
        # a query whose 32nd / 33rd neighbour distances tie at fp32 round-off may pick the other set in blocks 1, 2
        df = d.flatten()
        assert float(df.mean()) < 2e-5 and float(df.kthvalue(int(0.995 * df.numel())).values) < 1e-4, (layer, float(df.mean()), float(df.max()))


def test_last_block_feed_forward_is_computed_only_when_read():
    """PtEmbedTRv4.forward returns the coordinate stack only (ptEmb_transformer.py:115-121,371-376): the last block's
    feed-forward output feeds nothing unless the parametric tail or a debug tap reads it, and the path does not compute
    it then.  The coordinates must not notice: bit-equal with the taps (which force it) on and off."""
    z, meta = load_golden("medium")
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        lean = head(feat, metas, rj)["all_coords_preds"].clone()
        eng = head._engine_for(torch.device(DEV))
        eng.enable_taps(True)
        full = head(feat, metas, rj)["all_coords_preds"].clone()
        eng.enable_taps(False)
    assert torch.equal(lean, full)


@pytest.mark.parametrize("mode", ["split_f16x3", "split_f16x3_all"])
@pytest.mark.parametrize("name", ["small", "medium", "large", "huge", "ragged"])
def test_split_precision_mode_vs_golden(name, mode):
    """Opt-in POEM_PRECISION_SPLIT_F16X3 (csrc/vecattn_split.hip: the vector attention's C x C GEMMs as hi/lo f16 splits on
    the f16 matrix cores, fp32 accumulation): same bar as the default path -- MPVPE vs the reference <= 1e-3 mm -- and it
    must stay at the fp32 kernel's own distance from the reference (no more than 4x it + 2e-5 mm)."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    ref = torch.from_numpy(z["all_coords_preds"])
    with torch.no_grad():
        exact = head(feat, metas, rj)["all_coords_preds"].cpu()
        head.set_precision(mode)
        got = head(feat, metas, rj)["all_coords_preds"].cpu()
        head.set_precision("fp32")
        again = head(feat, metas, rj)["all_coords_preds"].cpu()
    assert torch.equal(exact, again)                                   # the switch really switches back
    assert not torch.equal(exact, got)                                 # ... and the split kernel really ran
    d_exact = float(torch.norm(exact[-1, :, 21:] - ref[-1, :, 21:], dim=-1).mean(dim=1).max())
    d_split = float(torch.norm(got[-1, :, 21:] - ref[-1, :, 21:], dim=-1).mean(dim=1).max())
    assert d_split < 1e-6, (d_split, d_exact)                          # metres: the 1e-3 mm bar
    assert d_split < 4 * d_exact + 2e-8, (d_split, d_exact)
    assert _md(got, ref) < 1e-4


def test_split_precision_stage_taps_vs_reference():
    """Stage by stage (POEM-medium fixture: the reference's own per-block tensors): at every tap the split-precision path is
    no further from the reference than 2x the exact path's distance + fp32 round-off of the tensor's scale."""
    z, meta = load_golden("medium")
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    B, C, Q = len(spec["views"]), spec["embed"], 799
    dist = {}
    for mode in ("fp32", "split_f16x3", "split_f16x3_all"):
        head.set_precision(mode)
        with torch.no_grad():
            head(feat, metas, rj)
        for i in range(3):
            for k in ("f_self", "f_cross", "feats", "xyz"):
                ref = torch.from_numpy(z[f"tap.b{i}.{k}"])
                got = eng.tap(f"b{i}.{k}", (B, Q, 3 if k == "xyz" else C))
                got = got if k == "xyz" else got[:, ::(Q + ref.shape[1] - 1) // ref.shape[1]]
                assert got.shape == ref.shape, (k, got.shape, ref.shape)
                dist[(mode, i, k)] = (_md(got, ref), float(ref.abs().max()))
    head.set_precision("fp32")
    for i in range(3):
        for k in ("f_self", "f_cross", "feats", "xyz"):
            for mode in ("split_f16x3", "split_f16x3_all"):
                (de, scale), (ds, _) = dist[("fp32", i, k)], dist[(mode, i, k)]
                assert ds <= 2 * de + 4e-6 * scale, (mode, i, k, ds, de, scale)


def test_split_precision_needs_embed_128():
    z, meta = load_golden("tiny")
    head = build_hip_head(meta["spec"], DEV)
    cfg, w, consts, batch = case_setup(meta["spec"])
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        head(feat, metas, rj)
        with pytest.raises(RuntimeError):
            head.set_precision("split_f16x3")                           # embed 32: no split images
    with pytest.raises(ValueError):
        head.set_precision("bf16")


def test_ragged_views_and_batch_independence():
    """Samples do not interact: a ragged batch equals the per-sample runs (the property the DP shard relies on)."""
    spec = dict(embed=128, nsample=4096, views=[3, 1, 8, 2], seed=21, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        full = head(feat, metas, rj)["all_coords_preds"].cpu()
    orc = run_oracle(cfg, w, consts, batch)["all_coords_preds"]
    assert float(torch.norm(full[-1] - orc[-1], dim=-1).mean()) < 1e-6
    offs = np.concatenate([[0], np.cumsum(spec["views"])])
    for i in (1, 2):
        m = dict(metas)
        m["cam_intr"] = metas["cam_intr"][offs[i]:offs[i + 1]].contiguous()
        m["cam_extr"] = metas["cam_extr"][offs[i]:offs[i + 1]].contiguous()
        m["cam_view_num"] = np.asarray([spec["views"][i]])
        m["master_id"] = [0]
        with torch.no_grad():
            one = head(feat[offs[i]:offs[i + 1]].contiguous(), m, rj[i:i + 1].contiguous())["all_coords_preds"].cpu()
        assert torch.equal(one[:, 0], full[:, i])


@pytest.mark.parametrize("embed", [128, 256])
def test_fused_sampling_every_view_count(embed):
    """Q1 makes the merge address memory by r = s * N + n across (view, channel, segment) planes, so every view count has its
    own pattern of rows with n == 0, of rows that straddle planes and of tiles that straddle samples: one sample of each
    N = 1 .. 10 in one batch, fused front end against the oracle's sampling stage and against the operator sequence."""
    spec = dict(embed=embed, nsample=4096, views=[7, 1, 10, 3, 2, 9, 5, 4, 8, 6], seed=77, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    taps = {}
    # (the decoder behind the sampling stage is compared with the oracle at C = 128; at C = 256 the two front ends' vertices with
    #  each other -- the oracle's decoder on ten samples is 20 s of CPU time that other tests already spend on this width)
    orc = run_oracle(cfg, w, consts, batch, taps=taps, stop_after_sampling=embed != 128)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    bf, outs = {}, {}
    for mode in (1, 0):
        eng.set_option("fused_sampling", mode)
        with torch.no_grad():
            outs[mode] = got = head(feat, metas, rj)["all_coords_preds"].cpu()
        bf[mode] = eng.tap("bps_feat", (10, 4096, embed)).cpu()
        scale = max(1.0, float(taps["bps_feat"].abs().max()))
        per_sample = (bf[mode] - taps["bps_feat"]).abs().amax(dim=(1, 2))
        assert float(per_sample.max()) < 2e-5 * scale, (mode, per_sample)
        if orc is not None:
            assert float(torch.norm(got[-1] - orc["all_coords_preds"][-1], dim=-1).mean()) < 1e-6
    assert _md(bf[1], bf[0]) < 1e-5 * scale
    assert float(torch.norm(outs[1][-1] - outs[0][-1], dim=-1).mean()) < 1e-6


@pytest.mark.parametrize("embed", [128, 256, 512])
def test_grouped_sampling_kernel_is_bit_identical_to_the_two_kernel_form(embed):
    """sample_group_kernel (merge.hip): the samples whose view count divides 8 run the whole sampling stage in one kernel
    (merge_net[0]'s hidden rows stay on the chip); the others -- and every sample of a batch too small to fill the chip with
    the 8-tile units -- go through sample_merge_kernel + merge_tail_kernel.  Which form a sample takes depends on the batch,
    so the two must give the same bits: one sample of each N = 1 .. 10, forced through the grouped kernel
    (group_min_views = 1: N = 1, 2, 4, 8 grouped, the rest two-kernel, in one forward) vs never (-1), `bps_feat` and the
    vertices bit for bit; both against the oracle's sampling stage."""
    spec = dict(embed=embed, nsample=4096, views=[7, 1, 10, 3, 2, 9, 5, 4, 8, 6, 8, 1, 2, 4], seed=78, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    taps = {}
    run_oracle(cfg, w, consts, batch, taps=taps, stop_after_sampling=True)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    B = len(spec["views"])
    bf, out = {}, {}
    for mode in (1, -1, 0):
        eng.set_option("group_min_views", mode)
        with torch.no_grad():
            for _ in range(3):                      # plain launches, capture, replay
                out[mode] = head(feat, metas, rj)["all_coords_preds"].cpu()
        bf[mode] = eng.tap("bps_feat", (B, 4096, embed)).cpu()
    scale = max(1.0, float(taps["bps_feat"].abs().max()))
    assert _md(bf[1], taps["bps_feat"]) < 2e-5 * scale
    assert torch.equal(bf[1], bf[-1]) and torch.equal(out[1], out[-1])
    assert torch.equal(bf[0], bf[-1]) and torch.equal(out[0], out[-1])      # default threshold: whichever form it picks


@pytest.mark.parametrize("fused", [1, 0])
def test_projections_outside_the_image_and_behind_the_camera(fused):
    """grid_sample's zero padding and the |z| < 1e-7 clamp of the projection (ptEmb_head.py:880-883,900): views that see the
    hand only partly (taps with one, two or no valid corner), not at all, or from behind (negative depth), mixed with normal
    ones in a ragged batch -- both front ends (sample.hip's clamped taps with zero weights, merge.hip's weight / pixel table)
    against the oracle's F.grid_sample, at the sampling stage's output and at the final vertices."""
    spec = dict(embed=128, nsample=4096, views=[4, 1, 3], seed=33, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    m = batch["img_metas"]
    K, E = m["cam_intr"].clone(), m["cam_extr"].clone()
    K[1, 0, 2] += 200.0                       # principal point shifted: the hand straddles the right image border
    K[2, 1, 2] -= 150.0                       # ... and the top border
    K[3, 0, 2] += 2000.0                      # nothing inside the image: every tap is padding
    E[5] = E[5] @ torch.diag(torch.tensor([-1.0, 1.0, -1.0, 1.0]))      # camera turned round: the hand is behind it (z < 0)
    K[6, 0, 0] = K[6, 1, 1] = 2000.0          # long lens: a few pixels of the map cover the whole ball
    m["cam_intr"], m["cam_extr"] = K.contiguous(), E.contiguous()
    taps = {}
    orc = run_oracle(cfg, w, consts, batch, taps=taps)["all_coords_preds"]
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    eng = head._engine_for(torch.device(DEV))
    eng.enable_taps(True)
    eng.set_option("fused_sampling", fused)
    with torch.no_grad():
        got = head(feat, metas, rj)["all_coords_preds"].cpu()
    bf = eng.tap("bps_feat", (3, 4096, 128)).cpu()
    assert torch.isfinite(bf).all() and torch.isfinite(got).all()
    assert _md(bf, taps["bps_feat"]) < 2e-5 * max(1.0, float(taps["bps_feat"].abs().max()))
    assert float(torch.norm(got[-1] - orc[-1], dim=-1).mean()) < 1e-6
    # the all-padding view really contributes zeros: its sampled planes vanish in the operator sequence's tensor
    if not fused:
        g = eng.tap("g", (8, 128, 4096)).cpu()
        assert float(g[3].abs().max()) == 0.0 and float(g[0].abs().max()) > 0.0


@pytest.mark.parametrize("embed,fh,fw,img,petr", [(128, 32, 32, 512, False), (256, 8, 24, (384, 128), False),
                                                  (128, 32, 32, 512, True), (256, 8, 24, (384, 128), True)])
def test_other_feature_map_and_image_sizes_through_the_whole_path(embed, fh, fw, img, petr):
    """The reference head is size-agnostic (ptEmb_head.py:831-838: inp_res from the batch, the feature map's own H x W in the
    positional table, grid_sample on whatever map arrives); PoemConfig.feat_h / feat_w carry it here.  512 x 512 images with
    32 x 32 features, and a non-square map / image: whole path against the oracle, fused and operator front ends, and a sample
    alone against the batch (bit for bit)."""
    views = [2, 3, 8]
    g = torch.Generator().manual_seed(91)
    iw, ih = (img, img) if isinstance(img, int) else img
    b = pk.inputs.synthetic_batch(views, seed=91)
    b["mlvl_feat"] = torch.randn(sum(views), 160, fh, fw, generator=g)
    K = b["img_metas"]["cam_intr"].clone()
    K[:, 0, 0] *= iw / 256.0; K[:, 0, 2] = iw / 2.0; K[:, 1, 1] *= ih / 256.0; K[:, 1, 2] = ih / 2.0
    b["img_metas"]["cam_intr"] = K
    b["img_metas"]["inp_img_shape"] = (iw, ih)
    spec = dict(embed=embed, nsample=4096, views=views, seed=91, parametric=False)
    if petr:      # the frustum grid follows the feature map and the (non-square) image: position_embeding's (h, w) naming, :115
        spec.update(petr=True, lid=True, depth_num=8, depth_start=0.05, depth_end=1.4)
    cfg, w, consts, _ = case_setup(spec)
    taps = {}
    orc = run_oracle(cfg, w, consts, b, taps=taps)["all_coords_preds"]
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(b, DEV)
    eng = None
    outs = {}
    for fused in (1, 0):
        head.set_option("fused_sampling", fused)
        with torch.no_grad():
            for _ in range(3):
                outs[fused] = head(feat, metas, rj)["all_coords_preds"].cpu()
        eng = head._engine
        assert (eng.cfg.feat_h, eng.cfg.feat_w) == (fh, fw)
        assert float(torch.norm(outs[fused][-1, :, 21:] - orc[-1, :, 21:], dim=-1).mean()) < 1e-6, fused
    head.set_option("fused_sampling", 1)
    m1 = dict(metas)
    m1["cam_intr"], m1["cam_extr"] = metas["cam_intr"][2:5].contiguous(), metas["cam_extr"][2:5].contiguous()
    m1["cam_view_num"], m1["master_id"] = np.asarray([3]), [0]
    with torch.no_grad():
        one = head(feat[2:5].contiguous(), m1, rj[1:2].contiguous())["all_coords_preds"].cpu()
    assert torch.equal(one[:, 0], outs[1][:, 1])


def test_petr_embedding_ragged_batch_both_front_ends_and_graph_replays():
    """PETR_EMBEDDING through everything the default path has: a ragged batch, the fused and the operator sampling front ends
    (bit-identical `x` either way: the same input_proj launch), graph replays (the per-view table is rebuilt from the caller's
    cameras in front of every replay: new cameras, same graph), a sample alone == the sample in its batch."""
    spec = dict(embed=256, nsample=4096, views=[8, 3, 1, 8], seed=77, parametric=False, petr=True)
    cfg, w, consts, batch = case_setup(spec)
    orc = run_oracle(cfg, w, consts, batch)["all_coords_preds"]
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    outs = {}
    with torch.no_grad():
        for fused in (1, 0):
            head.set_option("fused_sampling", fused)
            for _ in range(3):
                outs[fused] = head(feat, metas, rj)["all_coords_preds"].cpu()
            assert float(torch.norm(outs[fused][-1, :, 21:] - orc[-1, :, 21:], dim=-1).mean()) < 1e-6, fused
        head.set_option("fused_sampling", 1)
        assert head._engine.graph_stats()["replays"] >= 1
        # other cameras through the same engine and graph
        b2 = pk.inputs.synthetic_batch(spec["views"], seed=78)
        feat2, metas2, rj2 = batch_to(b2, DEV)
        got2 = head(feat2, metas2, rj2)["all_coords_preds"].cpu()
        orc2 = run_oracle(cfg, w, consts, b2)["all_coords_preds"]
        assert float(torch.norm(got2[-1, :, 21:] - orc2[-1, :, 21:], dim=-1).mean()) < 1e-6
        m1 = dict(metas)
        m1["cam_intr"], m1["cam_extr"] = metas["cam_intr"][8:11].contiguous(), metas["cam_extr"][8:11].contiguous()
        m1["cam_view_num"], m1["master_id"] = np.asarray([3]), [0]
        one = head(feat[8:11].contiguous(), m1, rj[1:2].contiguous())["all_coords_preds"].cpu()
    assert torch.equal(one[:, 0], outs[1][:, 1])
    # without the switch the same weights give another answer (the embedding is live)
    spec0 = dict(spec, petr=False)
    head0 = build_hip_head(spec0, DEV)
    with torch.no_grad():
        base = head0(feat, metas, rj)["all_coords_preds"].cpu()
    assert float((base - outs[1]).abs().max()) > 1e-4


def test_errors_are_loud():
    head = pk.build_head(__import__("util").head_cfg(128), data_preset=pk.CN({})).eval()
    b = pk.inputs.synthetic_batch([2], seed=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        head(b["mlvl_feat"], b["img_metas"], b["reference_joints"])      # CPU tensors: no fallback
    head.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        head(b["mlvl_feat"].to(DEV), b["img_metas"], b["reference_joints"].to(DEV))   # training through the HIP head: refused
    with pytest.raises(RuntimeError):
        hip.gemm(torch.zeros(4, 8), torch.zeros(8, dtype=torch.uint8), 4)


def test_full_size_batch_properties():
    """BASELINE configs[1] size (batch 32 x 8 views): size-independent properties instead of an oracle run --
    (i) every sample of the big batch equals its own single-sample run bit for bit (no cross-sample coupling, the
    property data-parallel sharding relies on); (ii) permuting the samples permutes the outputs; (iii) the 4 samples the
    oracle can afford agree to the MPVPE bar."""
    spec = dict(embed=256, nsample=4096, views=[8] * 32, seed=31, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        full = head(feat, metas, rj)["all_coords_preds"]
    assert torch.isfinite(full).all()

    def sub(idx):
        rows = torch.cat([torch.arange(8 * i, 8 * i + 8) for i in idx]).to(DEV)
        m = dict(metas)
        m["cam_intr"], m["cam_extr"] = metas["cam_intr"][rows].contiguous(), metas["cam_extr"][rows].contiguous()
        m["cam_view_num"] = np.asarray([8] * len(idx))
        m["master_id"] = [0] * len(idx)
        with torch.no_grad():
            return head(feat[rows].contiguous(), m, rj[torch.tensor(idx).to(DEV)].contiguous())["all_coords_preds"]

    one = sub([17])
    assert torch.equal(one[:, 0], full[:, 17])
    perm = [5, 30, 0, 11]
    assert torch.equal(sub(perm), full[:, perm])
    # the small-per-GPU-batch regime (the reference evaluates at batch 2, lib/opt.py:27-30 upstream): B = 2 and B = 8 take
    # other row-tile heights / grids than B = 32 (chain.hip chain_tile_p) -- same arithmetic per row, bit for bit
    assert torch.equal(sub([9, 20]), full[:, [9, 20]])
    eight = list(range(8, 16))
    assert torch.equal(sub(eight), full[:, eight])
    # A/B switches that must not change a bit: forced 32- / 64-row chain tiles, the block-0 anchor tables rebuilt per forward
    # (the round-2 behaviour) instead of read from the handle, where poem_create folded them
    eng = head._engine
    # ... and the hipGraph replay of the launch list against plain launches (first call of a layout captures, later calls replay)
    for name, val in (("chain_tile", 1), ("chain_tile", 2), ("chain_tile", 3), ("chain_tile", 0), ("tables_cached", 0), ("tables_cached", 1),
                      # (a key is captured at its second forward and replayed from the third: csrc/forward.cpp)
                      ("graphs", 0), ("graphs", 1), ("graphs", 1), ("graphs", 1),
                      # the cross attention's split-key partials merged by the chain kernel that consumes them (default), by a
                      # combine launch, inside the attention kernel: the same merge arithmetic in chunk order, three places
                      ("chain_combine", 0), ("xattn_merge", 1), ("xattn_merge", 0), ("chain_combine", 1),
                      # round 4: the launch-count / dependency shortcuts (one input launch, no dead embedding broadcast, block 0's F1 as
                      # two launches, anchor rows out of the chain's rows), the XCD-aware panel map, the split F1, one-query blocks
                      ("small_batch", 0), ("small_batch", 1), ("gemm_xcd_map", 0), ("gemm_xcd_map", 1), ("f1_split", 0), ("f1_split", 1),
                      ("va_p1", 1), ("va_p1", 2), ("va_p1", -1), ("bps_defer", 2), ("bps_defer", 0)):
        eng.set_option(name, val)
        with torch.no_grad():
            again = head(feat, metas, rj)["all_coords_preds"]
        assert torch.equal(again, full), (name, val)
    # the same on a batch of two, where the small-batch shortcuts are the ones in use
    two = sub([9, 20])
    for name, val in (("small_batch", 0), ("small_batch", 1), ("xattn_merge", 0), ("xattn_merge", -1), ("va_p1", 0), ("va_p1", -1),
                      ("chain_tile", 1), ("chain_tile", 0), ("graphs", 0), ("graphs", 1)):
        eng.set_option(name, val)
        assert torch.equal(sub([9, 20]), two), (name, val)
    # ... and on a single sample (the merged attention on channel-tile items)
    for name, val in (("xattn_half", 0), ("xattn_half", 1), ("xattn_merge", 0), ("xattn_merge", -1)):
        eng.set_option(name, val)
        assert torch.equal(sub([17]), one), (name, val)
    # oracle on 2 samples of the big batch
    b2 = dict(mlvl_feat=batch["mlvl_feat"][:16], reference_joints=batch["reference_joints"][:2],
              img_metas=dict(batch["img_metas"], cam_intr=batch["img_metas"]["cam_intr"][:16],
                             cam_extr=batch["img_metas"]["cam_extr"][:16], cam_view_num=np.asarray([8, 8]), master_id=[0, 0]))
    orc = run_oracle(cfg, w, consts, b2)["all_coords_preds"]
    mp = torch.norm(full[-1, :2, 21:].cpu() - orc[-1, :, 21:], dim=-1).mean(dim=1)
    assert float(mp.max()) < 1e-6, mp


def test_ragged_stream_fresh_layout_every_batch_replays_one_graph():
    """Row R1: the reference's collation hands the head a NEW view layout with every batch (collation_random_n_views,
    lib/utils/collation.py:7-25; view counts drawn per sample, lib/data_wds/multiview_wds.py:86-95; evaluation at batch 2,
    lib/opt.py:27-30 upstream).  300 distinct layouts through one head: every result equals the plain-launch path's bit for bit,
    the whole stream is served by ONE captured graph per batch size (no capture, no instantiate, no blocking upload per
    layout), and nothing accumulates: the process-wide count of parked graph execs does not move."""
    import gc
    import random
    # (the parked-exec count is process-wide: heads of EARLIER tests that the garbage collector has not freed yet would park
    #  their execs in the middle of this stream -- seen once as 43 -> 65 -- so they are collected before the count is taken)
    gc.collect()
    torch.cuda.synchronize()
    spec = dict(embed=128, nsample=4096, views=[2, 3], seed=71, parametric=False)
    head = build_hip_head(spec, DEV)
    rng = random.Random(7)
    layouts, seen = [], set()
    while len(layouts) < 300:
        B = (2, 3, 3, 3, 3, 4, 4, 4, 4, 4)[len(layouts) % 10]      # three batch sizes interleaved (30 / 120 / 150 layouts)
        v = tuple(min(max(1, int(round(rng.gauss(4, 2)))), 10) for _ in range(B))
        if v not in seen:
            seen.add(v)
            layouts.append(list(v))
    g = torch.Generator().manual_seed(5)
    pool = torch.randn(40, 160, 16, 16, generator=g).to(DEV)
    rj_all = (torch.tensor([0.0, 0.0, 0.6]) + 0.03 * torch.randn(4, 21, 3, generator=g)).to(DEV)
    K = torch.tensor([[300.0, 0, 128.0], [0, 300.0, 128.0], [0, 0, 1]])
    items = []
    for v in layouts:
        bn = sum(v)
        extr = torch.cat([pk.inputs.ring_extrinsics(n, ring=10) for n in v], 0)
        m = {"inp_img_shape": (256, 256), "cam_intr": K[None].repeat(bn, 1, 1).contiguous().to(DEV), "cam_extr": extr.contiguous().to(DEV),
             "master_id": [0] * len(v), "cam_view_num": np.asarray(v, dtype=np.int64)}
        items.append((pool[:bn], m, rj_all[:len(v)].contiguous()))
    with torch.no_grad():
        head(*max(items, key=lambda it: len(it[1]["cam_view_num"])))      # the grow-only workspace at its high-water mark (the
        head(*items[0])                                                   # workspace pointer is part of a graph's key)
        eng = head._engine
        gc.collect()
        before = eng.graph_stats()
        outs = [head(*it)["all_coords_preds"].clone() for it in items]
        torch.cuda.synchronize()
        after = eng.graph_stats()
        eng.set_option("graphs", 0)
        for it, want in zip(items, outs):
            assert torch.equal(head(*it)["all_coords_preds"], want), it[1]["cam_view_num"]
        eng.set_option("graphs", 1)
    d = {k: after[k] - before[k] for k in after}
    # three batch sizes -> at most three captures (each at the second sight of its key), at most three instantiations (fewer
    # when a parked exec of an earlier head of this process is taken over); everything else is a replay
    assert d["captures"] <= 3 and d["instantiations"] <= 3, d
    assert d["replays"] >= 300 - 6 and d["plain_forwards"] <= 6, d
    assert d["layout_uploads"] == 300 - 1 or d["layout_uploads"] == 300, d      # (items[0] was the layout already resident)
    assert after["cached_execs"] <= 3 and after["parked_execs"] <= before["parked_execs"], (before, after)
    assert after["exec_update_refusals"] == before["exec_update_refusals"], (before, after)


def test_head_on_two_streams_keeps_one_engine_per_stream():
    """Forwards issued alternately on two torch streams (a small-batch evaluation loop that wants consecutive forwards to
    overlap): the head keeps one engine -- workspace, layout arrays, side streams, graphs -- per stream, so the forwards never
    share scratch memory; results equal the single-stream run bit for bit."""
    spec = dict(embed=128, nsample=4096, views=[2, 3], seed=73, parametric=False)
    head = build_hip_head(spec, DEV)
    items = []
    for sd in range(6):
        cfg, w, consts, batch = case_setup(dict(spec, seed=100 + sd, views=[2 + sd % 3, 3]))
        items.append(batch_to(batch, DEV))
    with torch.no_grad():
        want = [head(*it)["all_coords_preds"].clone() for it in items]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        got = []
        for rep in range(3):
            got = []
            for i, it in enumerate(items):
                with torch.cuda.stream(streams[i % 2]):
                    got.append(head(*it)["all_coords_preds"])
        torch.cuda.synchronize()
    assert len(head._engines) == 3                      # the default stream's + one per side stream
    for g, w_ in zip(got, want):
        assert torch.equal(g, w_)


def test_view_layout_arrays_for_any_ragged_batch():
    """The CSR arrays of a ragged batch are built on the device from offsets carried in a kernel's argument segment
    (csrc/misc.hip view_layout_kernel): through the whole path, batches at the edges of that route -- one sample, one view each,
    max_views each, a batch larger than the head has seen -- give the same results as the same samples run one by one (the
    per-sample runs use the trivial layout [0, n])."""
    import random
    spec = dict(embed=128, nsample=4096, views=[1], seed=74, parametric=False)
    head = build_hip_head(spec, DEV)
    rng = random.Random(3)
    for views in ([1], [10], [1] * 7, [10] * 3, [rng.randint(1, 10) for _ in range(13)], [2, 1, 10, 1, 3]):
        cfg, w, consts, batch = case_setup(dict(spec, views=views, seed=200 + len(views)))
        feat, metas, rj = batch_to(batch, DEV)
        with torch.no_grad():
            full = head(feat, metas, rj)["all_coords_preds"]
            offs = np.concatenate([[0], np.cumsum(views)])
            for i in sorted({0, len(views) // 2, len(views) - 1}):
                rows = torch.arange(int(offs[i]), int(offs[i + 1])).to(DEV)
                m = dict(metas, cam_intr=metas["cam_intr"][rows].contiguous(), cam_extr=metas["cam_extr"][rows].contiguous(),
                         cam_view_num=np.asarray([views[i]]), master_id=[0])
                one = head(feat[rows].contiguous(), m, rj[i:i + 1].contiguous())["all_coords_preds"]
                assert torch.equal(one[:, 0], full[:, i]), (views, i)


def test_retired_graph_execs_are_reused_not_accumulated():
    """Heads come and go (engine rebuilds after load_state_dict, test suites, periodic evaluation): a destroyed handle parks its
    graph execs -- they cannot be destroyed on this runtime (csrc/handle.cpp) -- and the next capture of the same shape takes
    one over through hipGraphExecUpdate.  Ten head life cycles must not grow the parked list beyond what one cycle leaves, and
    the re-used execs must compute the same bits as fresh ones."""
    import gc
    gc.collect()                                   # (heads of earlier tests park their execs now, not in the middle of the count)
    torch.cuda.synchronize()
    spec = dict(embed=128, nsample=4096, views=[2, 3], seed=72, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    feat, metas, rj = batch_to(batch, DEV)
    want, parked, stats = None, [], None
    for cycle in range(10):
        head = build_hip_head(spec, DEV)
        with torch.no_grad():
            for _ in range(3):
                out = head(feat, metas, rj)["all_coords_preds"]
        torch.cuda.synchronize()
        stats = head._engine.graph_stats()
        assert stats["replays"] >= 1, stats
        want = out.clone() if want is None else want
        assert torch.equal(out, want), cycle
        del head, out
        gc.collect()
        parked.append(stats["parked_execs"])
    assert stats["exec_reuses"] >= 8 or stats["exec_update_refusals"] == 0, stats
    assert max(parked) <= parked[1] + 1, parked          # steady state after the first cycle


def test_parked_graph_exec_is_not_updated_while_its_last_launch_is_running():
    """A retired exec is rewritten in place by hipGraphExecUpdate when a later capture of the same shape takes it over: it must
    have FINISHED (and sit on the same device).  Head A replays its graph many times on stream 1 and is dropped with those
    launches still queued; head B on stream 2 -- whose work does not wait for stream 1, and whose engine already exists --
    captures the same shape at once: the parked exec is passed over (`exec_busy_skips`) and a fresh one instantiated.  Once
    everything has drained, a third head takes a parked exec over.  Every head computes the same bits."""
    import gc
    spec = dict(embed=256, nsample=4096, views=[8] * 24, seed=73, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    feat, metas, rj = batch_to(batch, DEV)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    head, head_b = build_hip_head(spec, DEV), build_hip_head(spec, DEV)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            for _ in range(3):
                want = head(feat, metas, rj)["all_coords_preds"].clone()
        with torch.cuda.stream(s2):
            out_b = head_b(feat, metas, rj)["all_coords_preds"]          # engine built, key seen once: the next forward captures
        torch.cuda.synchronize()
        base = head._engine.graph_stats()
        assert base["replays"] >= 1 and torch.equal(out_b, want)
        with torch.cuda.stream(s1):
            for _ in range(16):                     # ~0.25 s of queued replays of the exec that is about to be parked
                out_a = head(feat, metas, rj)["all_coords_preds"]
        del head
        gc.collect()                                 # poem_destroy: the exec is parked behind its last launch
        with torch.cuda.stream(s2):
            for _ in range(2):
                out_b = head_b(feat, metas, rj)["all_coords_preds"]      # capture (re-use attempt), replay
        st_b = head_b._engine.graph_stats()
    torch.cuda.synchronize()
    assert torch.equal(out_a, want) and torch.equal(out_b, want)
    assert st_b["replays"] >= 1, st_b
    # (the skip is the expected outcome; a host slow enough that stream 1 drained first re-uses the exec instead)
    assert st_b["exec_busy_skips"] > base["exec_busy_skips"] or st_b["exec_reuses"] > base["exec_reuses"], (base, st_b)
    del head_b
    gc.collect()
    torch.cuda.synchronize()
    head_c = build_hip_head(spec, DEV)
    with torch.no_grad():
        for _ in range(3):
            out_c = head_c(feat, metas, rj)["all_coords_preds"]
    torch.cuda.synchronize()
    st_c = head_c._engine.graph_stats()
    assert torch.equal(out_c, want)
    assert st_c["exec_reuses"] > st_b["exec_reuses"] or st_c["exec_update_refusals"] > st_b["exec_update_refusals"], (st_b, st_c)


def _full_size_properties(spec, n_oracle=1):
    """Size-independent properties at a BASELINE per-GPU load: finite; every probed sample of the big batch equals its own
    single-sample run bit for bit; a permuted sub-batch equals the permuted outputs; the first ``n_oracle`` samples meet the
    MPVPE bar against the oracle."""
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    views = spec["views"]
    offs = np.concatenate([[0], np.cumsum(views)])
    with torch.no_grad():
        full = head(feat, metas, rj)["all_coords_preds"]
    assert torch.isfinite(full).all()

    def sub(idx):
        rows = torch.cat([torch.arange(offs[i], offs[i + 1]) for i in idx]).to(DEV)
        m = dict(metas)
        m["cam_intr"], m["cam_extr"] = metas["cam_intr"][rows].contiguous(), metas["cam_extr"][rows].contiguous()
        m["cam_view_num"] = np.asarray([views[i] for i in idx])
        m["master_id"] = [0] * len(idx)
        with torch.no_grad():
            return head(feat[rows].contiguous(), m, rj[torch.tensor(idx).to(DEV)].contiguous())["all_coords_preds"]

    B = len(views)
    assert torch.equal(sub([B - 1])[:, 0], full[:, B - 1])
    perm = [B // 2, 0, B - 2]
    assert torch.equal(sub(perm), full[:, perm])
    nv = int(offs[n_oracle])
    b2 = dict(mlvl_feat=batch["mlvl_feat"][:nv], reference_joints=batch["reference_joints"][:n_oracle],
              img_metas=dict(batch["img_metas"], cam_intr=batch["img_metas"]["cam_intr"][:nv],
                             cam_extr=batch["img_metas"]["cam_extr"][:nv], cam_view_num=np.asarray(views[:n_oracle]),
                             master_id=[0] * n_oracle))
    orc = run_oracle(cfg, w, consts, b2)["all_coords_preds"]
    mp = torch.norm(full[-1, :n_oracle, 21:].cpu() - orc[-1, :, 21:], dim=-1).mean(dim=1)
    assert float(mp.max()) < 1e-6, mp


def test_full_size_config_c4_large_10_views_batch_16():
    """BASELINE configs[3]: POEM-large, 10 views, batch 16 on one GPU (the feature-sampling stress case)."""
    _full_size_properties(dict(embed=512, nsample=4096, views=[10] * 16, seed=41, parametric=False), n_oracle=1)


def test_full_size_config_c5_ragged_2_to_10_views_batch_64():
    """BASELINE configs[4]'s per-GPU load: POEM-medium, 64 samples with 2..10 views each (seed 5)."""
    views = np.random.RandomState(5).randint(2, 11, size=64).tolist()
    assert min(views) == 2 and max(views) == 10
    _full_size_properties(dict(embed=256, nsample=4096, views=views, seed=51, parametric=False), n_oracle=2)


@pytest.mark.parametrize("name", ["tinynan", "tinynan2", "smallnan"])
def test_nan_fixtures_vs_reference(name):
    """The reference's own outputs for a batch with a NaN sample (round-6 fixtures, tests/golden/make_golden.py): the HIP head
    returns the poisoned sample's centre bit for bit and the clean samples within the path's bar (operator front end at C = 32,
    fused front end and chains at C = 128)."""
    z, meta = load_golden(name)
    spec = meta["spec"]
    ref = torch.from_numpy(z["all_coords_preds"])
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(case_setup(spec)[3], DEV)
    with torch.no_grad():
        outs = [head(feat, metas, rj)["all_coords_preds"].cpu() for _ in range(3)]      # plain launches, capture, replay
    for out in outs:
        assert torch.isfinite(out).all()
        assert torch.equal(out[:, 1], ref[:, 1])                       # the poisoned sample: its centre, exactly
        for b in range(ref.shape[1]):
            assert float(torch.norm(out[-1, b, 21:] - ref[-1, b, 21:], dim=-1).mean()) < 1e-6, b


def test_nan_features_of_one_sample_give_its_centre_and_leave_the_others_alone():
    """SURVEY a20, whole path: ``interm_ref_pts = torch.nan_to_num(interm_ref_pts)`` (ptEmb_head.py:944) in front of the
    de-normalisation -- a sample whose backbone features are NaN comes out as its hand centre in every layer (0 * radius + c), and,
    because no kernel mixes samples, every other sample of the batch keeps the bits of its clean run.  The oracle agrees."""
    spec = dict(embed=128, nsample=4096, views=[2, 3, 2], seed=17, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        clean = head(feat, metas, rj)["all_coords_preds"]
    # (i) every view of sample 1; (ii) only a NON-master view of it (the NaN then enters through the cross-view dot products of
    # merge_features_mv, not through the master rows' residual); (iii) one pixel of one channel of its master view
    for case, sl in (("all views", (slice(2, 5),)), ("second view", (slice(3, 4),)), ("one pixel", (2, 7, 8, 8))):
        bad = feat.clone()
        bad[sl] = float("nan")
        with torch.no_grad():
            outs = [head(bad, metas, rj)["all_coords_preds"] for _ in range(3)]      # plain launches, capture, graph replay
        # the oracle (torch: ReLU and softmax propagate NaN, nan_to_num at the end) decides the single-pixel case -- a pixel poisons
        # the sample iff a basis point taps it; for whole NaN views the answer is the centre (checked against the oracle on the
        # CPU in tests/test_oracle_golden.py::test_nan_views_give_the_centre)
        ref = run_oracle(cfg, w, consts, dict(batch, mlvl_feat=bad.cpu()))["all_coords_preds"] if case == "one pixel" else None
        poisoned = ref is None or bool(torch.equal(ref[:, 1], batch["reference_joints"][1, 9].expand(3, 799, 3)))
        for out in outs:
            assert torch.isfinite(out).all(), case
            if poisoned:
                assert torch.equal(out[:, 1], rj[1, 9].expand(3, 799, 3)), case
            assert torch.equal(out[:, 0], clean[:, 0]) and torch.equal(out[:, 2], clean[:, 2]), case
            if ref is not None:
                assert float((out.cpu() - ref).abs().max()) < 1e-6, case


def test_full_size_config_c3_medium_mano_8_views_batch_32():
    """BASELINE configs[2]'s per-GPU load: POEM-medium_MANO, 8 views, batch 32, the parametric tail on the device (Q3 flatten,
    Linears, rot6d -> axis-angle, MANO linear blend skinning through the HIP ManoLayer on a synthetic asset set of MANO's
    shapes -- the real assets are licence-gated).  Properties: finite; a sample of the big batch equals its own
    single-sample run bit for bit (coordinates, pose, shape); the first two decoder layers equal the non-parametric head's
    (the tail only replaces the last layer); the last layer equals MANO(pred_pose, pred_shape) + centre re-evaluated by
    the oracle's LBS restatement."""
    import mano_oracle as mo
    spec = dict(embed=256, nsample=4096, views=[8] * 32, seed=61, parametric=True)
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    assets = pk.mano.synthetic_mano_assets(0)
    head.set_mano_layer(pk.ManoLayer(assets, center_idx=9, device=DEV))
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        out = head(feat, metas, rj)
    full, pose, shape = out["all_coords_preds"], out["pred_pose"], out["pred_shape"]
    assert full.shape == (3, 32, 799, 3) and pose.shape == (32, 16, 3) and shape.shape == (32, 10)
    assert torch.isfinite(full).all() and torch.isfinite(pose).all() and torch.isfinite(shape).all()
    rows = torch.arange(8 * 21, 8 * 22).to(DEV)
    m = dict(metas)
    m["cam_intr"], m["cam_extr"] = metas["cam_intr"][rows].contiguous(), metas["cam_extr"][rows].contiguous()
    m["cam_view_num"], m["master_id"] = np.asarray([8]), [0]
    with torch.no_grad():
        one = head(feat[rows].contiguous(), m, rj[21:22].contiguous())
    assert torch.equal(one["all_coords_preds"][:, 0], full[:, 21])
    assert torch.equal(one["pred_pose"][0], pose[21]) and torch.equal(one["pred_shape"][0], shape[21])
    # layers 0, 1 are untouched by the tail
    plain = build_hip_head(dict(spec, parametric=False), DEV)        # (seeded weights are keyed by tensor name: same common tensors)
    with torch.no_grad():
        base = plain(feat, metas, rj)["all_coords_preds"]
    assert torch.equal(base[:2], full[:2])
    # last layer = MANO(pose, shape) centred at joint 9, + the sample's centre (ptEmb_head.py:944-958 upstream: no radius scale)
    verts, joints = mo.mano_lbs(assets, pose.reshape(32, 48).cpu(), shape.cpu(), center_idx=9)
    want = torch.cat([joints, verts], dim=1) + batch["reference_joints"][:, 9:10].double()
    got = full[-1].cpu().double()
    assert float((got - want).abs().max()) < 5e-6
    # the bar of the path: last-layer MPVPE <= 1e-3 mm against the oracle's LBS on the regressed (pose, shape) -- with the centre
    # (|c| ~ 0.5 m: 3e-8 m of fp32 round-off per coordinate) and without it.  (Third-party legs -- manotorch, pytorch3d's
    # rot6d chain -- are unpinned upstream: the oracle restates the published MANO model.)
    mpvpe = float(torch.linalg.norm(got[:, 21:] - want[:, 21:], dim=-1).mean())
    assert mpvpe <= 1e-6, mpvpe
    layer = head.mano_layer(out["pred_pose"].reshape(32, 48), shape)
    mpvpe0 = float(torch.linalg.norm(layer.verts.cpu().double() - verts, dim=-1).mean())
    assert mpvpe0 <= 1e-6, mpvpe0
    # the layer ran INSIDE the forward's launch graph (poem_attach_mano) and gives the bits of the stand-alone call
    assert head._engine._mano is head.mano_layer.th_table
    assert torch.equal(full[-1, :, 21:], torch.nan_to_num(layer.verts) + rj[:, 9:10])
    assert torch.equal(full[-1, :, :21], torch.nan_to_num(layer.joints) + rj[:, 9:10])
    # ... and the Python-callable route (any other layer object: forward -> callable -> poem_finalize_parametric) agrees bit for bit
    head.mano_in_python = True
    head.set_mano_layer(head.mano_layer)
    assert head._engine._mano is None
    with torch.no_grad():
        again = head(feat, metas, rj)
    assert torch.equal(again["all_coords_preds"], full) and torch.equal(again["pred_pose"], pose)


def test_bench_gpus_2_launches_two_ranks_or_refuses():
    """bench.py --gpus 2 without a launcher: on a box with one GPU it refuses (no line claiming 2 GPUs); with
    POEM_SINGLE_DEVICE=1 it rehearses the 2-rank code path itself (torch.distributed.run, gloo, both ranks on cuda:0) and the
    line reports the ranks that actually joined and the process group's own world size."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "POEM_SINGLE_DEVICE", "POEM_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--rotate", "2", "--cpu-samples", "0", "--no-e2e", "--no-extra-configs"]
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
        env["POEM_SINGLE_DEVICE"] = "1"
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["ranks_joined"] == 2 and "world_size=2" in res["config"]["process_group"]
    assert res["value"] > 0 and res["scaling"] == "weak"


def test_eval_single_script_runs_the_path(tmp_path):
    """The eval_single.py counterpart end to end on the GPU (tiny synthetic epoch), cfg written back as upstream does."""
    import json
    import os
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgp = tmp_path / "cfg.yaml"
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "eval_single.py"), "--cfg", str(cfgp), "--dataset",
                          "DexYCB", "--view_min", "2", "--view_max", "4", "--model", "small", "-g", "0", "--epoch_size", "6"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["exp_id"] == "DexYCB_view_2_4_small" and res["samples"] == 6 and res["embed"] == 128
    assert np.isfinite(res["MPVPE_mm_vs_synthetic_gt"])
    y = yaml.safe_load(open(cfgp))
    assert y["MODEL"]["HEAD"]["EMBED_DIMS"] == 128 and y["DATASET"]["TEST"]["TARGET"]["VIEW_RANGE"] == [2, 4]
    # one stage earlier: backbone pyramid -> feat_decode / heatmap_stage (HIP) -> DLT -> head
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "eval_single.py"), "--cfg", str(cfgp), "--dataset",
                          "DexYCB", "--view_min", "2", "--view_max", "4", "--model", "small", "-g", "0", "--epoch_size", "4",
                          "--pyramid"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["scope"] == "pyramid->verts" and res["samples"] == 4 and np.isfinite(res["MPVPE_mm_vs_synthetic_gt"])
    # from record shards: tar records -> device transform -> full model (HRNet on PyTorch-ROCm) -> metrics  (N4)
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "eval_single.py"), "--cfg", str(cfgp), "--dataset",
                          "DexYCB", "--view_min", "2", "--view_max", "4", "--model", "small", "-g", "0", "--epoch_size", "8",
                          "--shards", str(tmp_path / "shards")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["scope"] == "shards->images->verts" and res["samples"] == 8 and np.isfinite(res["MPVPE_mm_vs_record_gt"])
    assert sorted(os.listdir(tmp_path / "shards")) == [f"DexYCB_mv_test-00000{i}.tar" for i in range(4)]


@pytest.mark.gpu
def test_neighbour_counts_below_32_are_part_of_the_handle_and_split_precision_refuses_them():
    """N_NEIGHBOR / N_NEIGHBOR_QUERY < 32 (round 6: the masked vector attention, vecattn.hip MODE 3): the counts reach the engine
    (`poem_config_t.knn`, option "knn_query"), a head built with 32 / 32 on the same weights lands elsewhere, graph replay and plain
    launches agree bit for bit, and the split-precision kernels -- which have no masked form -- refuse instead of ignoring the keys."""
    z, meta = load_golden("smallk")
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    feat, metas, rj = batch_to(batch, DEV)
    head = build_hip_head(spec, DEV)
    with torch.no_grad():
        a = head(feat, metas, rj)["all_coords_preds"].clone()
        b = head(feat, metas, rj)["all_coords_preds"].clone()            # replayed graph
        head.set_option("graphs", 0)
        c = head(feat, metas, rj)["all_coords_preds"].clone()            # plain launches
    assert torch.equal(a, b) and torch.equal(a, c)
    ref = torch.from_numpy(z["all_coords_preds"])
    # hot weights: the neighbour sets decide.  Every query whose first-k set differs from the reference's is a near-tie at rank k
    # of the reference's own distances (fp32 round-off), and there are few of them; the mesh stays within 1e-2 mm.
    eng = head._engine
    eng.enable_taps(True)
    with torch.no_grad():
        head(feat, metas, rj)
    B, Q = len(spec["views"]), 799
    pt_xyz = eng.tap("pt_xyz", (B, spec["nsample"], 3)).cpu()
    for blk in (1, 2):
        xyz = torch.from_numpy(z[f"tap.b{blk - 1}.xyz"])
        for which, k in (("self", spec["knn_query"]), ("cross", spec["knn"])):
            want = torch.from_numpy(z[f"tap.b{blk}.idx_{which}"].astype(np.int64))
            assert want.shape == (B, Q, k)
            got = eng.tap(f"b{blk}.idx_{which}", (B, Q, 32), torch.int32).cpu().long()[..., :k]
            same = (torch.sort(got, dim=-1).values == torch.sort(want, dim=-1).values).all(-1)
            assert float(same.float().mean()) > 0.995, (blk, which)
            for bb, q in torch.nonzero(~same).tolist():
                src = xyz if which == "self" else pt_xyz
                d = xyz[bb, q][None] - src[bb]
                d = d * d
                sd = torch.sort((d[:, 0] + d[:, 1]) + d[:, 2]).values
                assert float((sd[k] - sd[k - 1]) / sd[k - 1]) < 1e-5, (blk, which, bb, q)
    mpvpe = torch.norm(a.cpu()[-1, :, 21:] - ref[-1, :, 21:], dim=-1).mean(dim=1)
    assert float(mpvpe.max()) < 1e-5, mpvpe
    eng.enable_taps(False)
    full = build_hip_head(dict(spec, knn=32, knn_query=32), DEV)
    with torch.no_grad():
        d = full(feat, metas, rj)["all_coords_preds"]
    assert _md(d.cpu(), ref) > 1e-3
    head2 = build_hip_head(spec, DEV)
    with pytest.raises(RuntimeError):
        head2.set_precision("split_f16x3")            # (no engine yet: refused when the first forward creates it)
        with torch.no_grad():
            head2(feat, metas, rj)
    with pytest.raises(RuntimeError):
        head.set_precision("split_f16x3")             # a live engine refuses at once


def _random_constructor_specs(n, seed=2026):
    """Seeded draws over the constructor keys the path reads (widths with their legal head counts, block counts, both neighbour
    counts, basis-point counts (the anchor ids of the shipped asset reach 765: at least 1024), NORMALIZE, the parametric tail) and ragged view layouts."""
    g = np.random.default_rng(seed)
    specs = []
    for i in range(n):
        C = int(g.choice([32, 64, 128]))
        heads = int(g.choice([h for h in (1, 2, 4, 8, 16) if C % h == 0 and C // h in (8, 16, 32, 64)]))
        B = int(g.integers(1, 4))
        specs.append(dict(embed=C, heads=heads, nblocks=int(g.integers(1, 5)), nsample=int(g.choice([1024, 2048])),
                          knn=int(g.integers(1, 33)), knn_query=int(g.integers(1, 33)), views=[int(v) for v in g.integers(1, 5, size=B)],
                          seed=100 + i, parametric=bool(g.integers(0, 4) == 0), pe_normalize=bool(g.integers(0, 2))))
    return specs


@pytest.mark.gpu
@pytest.mark.parametrize("spec", _random_constructor_specs(12), ids=lambda s: "C{embed}h{heads}b{nblocks}S{nsample}k{knn}q{knn_query}".format(**s))
def test_random_constructor_configs_vs_oracle(spec):
    """Whole path at constructor settings no fixture holds (1-4 decoder blocks, 1-16 heads of 8-64 channels, neighbour counts 1-32,
    1024 / 2048 basis points, ragged 1-4 views): HIP head vs the oracle on the same seeded weights and inputs, every layer."""
    cfg, w, consts, batch = case_setup(spec)
    assert (cfg.heads, cfg.nblocks, cfg.knn, cfg.knn_query) == (spec["heads"], spec["nblocks"], spec["knn"], spec["knn_query"])
    head = build_hip_head(spec, DEV)
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        res = head(feat, metas, rj)
        again = head(feat, metas, rj)["all_coords_preds"]                 # replayed graph
    orc = run_oracle(cfg, w, consts, batch)
    got = res["all_coords_preds"].cpu()
    assert got.shape == orc["all_coords_preds"].shape == (spec["nblocks"], len(spec["views"]), 799, 3)
    assert torch.equal(res["all_coords_preds"], again)
    assert _md(got, orc["all_coords_preds"]) < 5e-6                                                   # metres
    if spec["parametric"]:
        assert _md(res["pred_pose"], orc["pred_pose"]) < 2e-4
        assert _md(res["pred_shape"], orc["pred_shape"]) < 2e-5
