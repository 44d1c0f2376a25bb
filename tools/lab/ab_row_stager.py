import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo/oracle')
import poem_v2_amd as pk
from poem_v2_amd import hip
import decode_oracle as do
dev = torch.device('cuda:0')
sd = do.seeded_decoder_state(0)
feats = [f.to(dev) for f in do.synthetic_mlvl_feats(256, 0)]
dec = pk.decode.FeatureDecoders(sd, dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
outs = {}
for v in (0, 3, 7, 3, 7):               # bits 0 / 1: row stager at 64^2 / 32^2; bit 2: pinned taps at 16^2
    hip.lib().poem_set_decode_option(b"row_stager", v & 3)
    hip.lib().poem_set_decode_option(b"pin32", v >> 2)
    t = timeit(lambda: dec.heatmap_stage(feats, 256, 256))
    outs[v] = dec.heatmap_stage(feats, 256, 256)
    print(f"row_stager={v}: heatmap_stage {t*1e3:.1f} us", flush=True)
o = [x if isinstance(x, torch.Tensor) else x[0] for x in outs.values()]
print("bit-equal:", all(torch.equal(o[0], y) for y in o[1:]))
