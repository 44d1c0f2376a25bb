"""Operator micro-benchmarks on the GPU box (development tool, not part of the product or the bench contract)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import poem_v2_amd as pk
from poem_v2_amd import hip

dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def gemm_cases():
    shapes = [(131072, 256, 256), (25568, 256, 256), (1048576, 256, 256), (25568, 1024, 256), (25568, 256, 1024),
              (1048576, 128, 256)]
    if os.environ.get("POEM_GEMM_SHAPES"):
        shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["POEM_GEMM_SHAPES"].split(",")]
    for (M, N, K) in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / math.sqrt(K)
        b = torch.randn(N, device=dev)
        wp = hip.pack_linear(w)
        xp = hip.pack_rows(x)
        ref = torch.nn.functional.linear(x[:4096].double(), w.double(), b.double())
        fl = 2.0 * M * N * K
        t0 = timeit(lambda: hip.gemm(x, wp, N, bias=b))
        line = f"M={M:8d} N={N:5d} K={K:5d}  v1 RM->RM {t0*1e3:8.1f}us {fl/t0/1e9:6.1f}TF"
        for ip in (False, True):
            for op in (False, True):
                xin = xp if ip else x
                y = hip.gemm_ex(xin, wp, M, N, K, bias=b, in_pa=ip, out_pa=op)
                yy = hip.unpack_rows(y, M, N) if op else y
                err = float((yy[:4096].double() - ref).abs().max())
                t = timeit(lambda: hip.gemm_ex(xin, wp, M, N, K, bias=b, in_pa=ip, out_pa=op))
                line += f" | {'PA' if ip else 'RM'}->{'PA' if op else 'RM'} {t*1e3:7.1f}us {fl/t/1e9:6.1f}TF e={err:.0e}"
        print(line, flush=True)


def kslab_cases():
    """POEM-huge's Linears (M = B * 799 query rows or B * 4096 basis-point rows, K = 1024 / 4096): the K-slab kernel (default
    dispatch of poem_gemm) against the operands-from-L2 kernel (poem_gemm_ex) it replaces there."""
    for (M, N, K) in [(6392, 1024, 1024), (6392, 2048, 1024), (6392, 3072, 1024), (6392, 5120, 1024), (6392, 1024, 4096),
                      (32768, 1024, 1024), (12784, 512, 2048)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / math.sqrt(K)
        b = torch.randn(N, device=dev)
        wp = hip.pack_linear(w)
        fl = 2.0 * M * N * K
        t0 = timeit(lambda: hip.gemm(x, wp, N, bias=b))
        t1 = timeit(lambda: hip.gemm_ex(x, wp, M, N, K, bias=b))
        print(f"M={M:6d} N={N:5d} K={K:5d}  kslab {t0*1e3:8.1f}us {fl/t0/1e9:6.1f}TF ({fl/t0/1e9/157.3:.2f})  |  gemm2 {t1*1e3:8.1f}us "
              f"{fl/t1/1e9:6.1f}TF ({fl/t1/1e9/157.3:.2f})", flush=True)


def vecattn_case(B=32, Q=799, NS=4096, C=256):
    import poem_oracle as po
    g = torch.Generator().manual_seed(0)
    qxyz = (torch.rand(B, Q, 3, generator=g) * 2 - 1).to(dev)
    sxyz = (torch.rand(B, NS, 3, generator=g) * 2 - 1).to(dev)
    q, k, v = (torch.randn(B, n, C, generator=g).to(dev) for n in (Q, NS, NS))
    idx = hip.knn(qxyz, sxyz)
    w = lambda *shp: (torch.randn(*shp, generator=g) / math.sqrt(shp[-1])).to(dev)
    wd1, bd1 = w(C, 3), w(C) * 0.1
    packs = [hip.pack_linear(w(C, C)) for _ in range(3)]
    bs = [w(C) * 0.1 for _ in range(3)]
    fn = lambda: hip.vector_attention(qxyz, sxyz, None, idx, q, k, v, wd1, bd1, packs[0], bs[0], packs[1], bs[1], packs[2], bs[2])
    t = timeit(fn, 10)
    fl = B * Q * 32 * (6.0 * C * C + 6.0 * C)
    print(f"vecattn B={B} Q={Q} NS={NS} C={C}: {t*1e3:8.1f} us  {fl/t/1e9:6.1f} TF", flush=True)


def attn_case(B=32, Q=799, NS=4096, C=256, heads=4):
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, n, C, generator=g).to(dev) for n in (Q, NS, NS))
    t = timeit(lambda: hip.cross_attention(q, k, v, heads), 10)
    fl = 4.0 * B * Q * NS * C
    print(f"cross_attn B={B} Q={Q} NS={NS} C={C}: {t*1e3:8.1f} us  {fl/t/1e9:6.1f} TF", flush=True)


def knn_case(B=32, Q=799, NS=4096):
    g = torch.Generator().manual_seed(0)
    qxyz = (torch.rand(B, Q, 3, generator=g) * 2 - 1).to(dev)
    sxyz = (torch.rand(B, NS, 3, generator=g) * 2 - 1).to(dev)
    t = timeit(lambda: hip.knn(qxyz, sxyz), 10)
    print(f"knn B={B} Q={Q} NS={NS}: {t*1e3:8.1f} us", flush=True)
    t = timeit(lambda: hip.knn(qxyz, qxyz), 10)
    print(f"knn B={B} Q={Q} NS={Q}: {t*1e3:8.1f} us", flush=True)


def decode_case(views=256):
    """N1 stage at the bench's view count: feat_decode + heatmap_stage (+ eager PyTorch-ROCm of the same restatement)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import decode_oracle as do
    sd = do.seeded_decoder_state(0)
    feats = [f.to(dev) for f in do.synthetic_mlvl_feats(views, 0)]
    dec = pk.decode.FeatureDecoders(sd, dev)
    if os.environ.get("POEM_FUSE_UPCAT"):
        dec.fuse_upcat = [c == "1" for c in os.environ["POEM_FUSE_UPCAT"]]
    from poem_v2_amd import hip
    for name in os.environ.get("POEM_DECODE_AB", "").split(","):
        if name:
            hip.lib().poem_set_decode_option(name.encode(), 0)
            print(f"   {name}=0: feat_decode {timeit(lambda: dec.feat_decode(feats), 10)*1e3:8.1f} us  heatmap_stage "
                  f"{timeit(lambda: dec.heatmap_stage(feats, 256, 256), 10)*1e3:8.1f} us", flush=True)
            hip.lib().poem_set_decode_option(name.encode(), 3 if name == "row_stager" else 1)
    t1 = timeit(lambda: dec.feat_decode(feats), 10)
    t2 = timeit(lambda: dec.heatmap_stage(feats, 256, 256), 10)
    fl1 = views * 2.0 * (9 * 40 * 80 * 1024 + 9 * 80 * 160 * 256 + 9 * 160 * 320 * 64 + 320 * 160 * 256)
    fl2 = views * 2.0 * 9 * (480 * 160 * 256 + 240 * 80 * 1024 + 120 * 40 * 4096)
    print(f"decode views={views}: feat_decode {t1*1e3:8.1f} us ({fl1/t1/1e9:6.1f} TF)  heatmap_stage {t2*1e3:8.1f} us ({fl2/t2/1e9:6.1f} TF)", flush=True)
    if os.environ.get("POEM_NO_EAGER"):
        return
    sdd = {k: v.to(dev) for k, v in sd.items()}
    with torch.no_grad():
        e1 = timeit(lambda: do.feat_decode(feats, sdd), 5)
        e2 = timeit(lambda: do.uv_decode(feats, sdd), 5)
    print(f"   PyTorch-ROCm eager (MIOpen): feat_decode {e1*1e3:8.1f} us  uv_decode {e2*1e3:8.1f} us", flush=True)


def down2_case(views=256):
    """feat_decode's three stride-2 ConvBlocks alone, staging wave on / off (`s2_staging_wave`)."""
    from poem_v2_amd import hip
    for cin, cout, r in ((40, 80, 64), (80, 160, 32), (160, 320, 16)):
        g = torch.Generator().manual_seed(0)
        sd = {"c.conv.weight": torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5), "c.conv.bias": torch.zeros(cout),
              "c.norm.weight": torch.ones(cout), "c.norm.bias": torch.zeros(cout), "c.norm.running_mean": torch.zeros(cout),
              "c.norm.running_var": torch.ones(cout)}
        conv = pk.decode._Conv3x3(sd, "c", dev)
        x = torch.randn(views, cin, r, r, generator=g).to(dev)
        ro = r // 2
        lat = torch.randn(views, cout, ro, ro, generator=g).to(dev)
        out = torch.empty(views, cout, ro, ro, device=dev)
        fl = views * 2.0 * 9 * cin * cout * ro * ro
        line = f"down2 {cin:3d}->{cout:3d} @{r:2d}^2 views={views}:"
        for on in (0, 1, 2, 3, 1):           # 0: round-3 kernel; 1: the default rule; 2 / 3: blocks per CU forced
            hip.lib().poem_set_decode_option(b"s2_staging_wave", on)
            t = timeit(lambda: conv.down2(x, r, r, out, pk.decode._plain_strides(cout, ro, ro), residual=lat), 20)
            line += f"  staging_wave={on} {t*1e3:7.1f} us ({fl/t/1e9:6.1f} TF)"
        print(line, flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    which = sys.argv[1:] or ["gemm"]
    for w in which:
        {"gemm": gemm_cases, "kslab": kslab_cases, "vecattn": vecattn_case, "attn": attn_case, "knn": knn_case, "decode": decode_case, "down2": down2_case}[w]()
