"""CPU oracle for the convolutional glue in front of the head.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Restates ``PtEmbedMultiviewStereoV2.feat_decode`` / ``uv_decode`` / ``heatmap_stage`` (lib/models/POEM.py:167-222
upstream, HRNet branch) with ``ConvBlock`` = Conv2d -> BatchNorm2d(eval) -> ReLU (lib/models/bricks/conv.py:4-45) in
plain torch on CPU.  PINNED: tests/golden/decode.npz holds outputs of the reference's own methods on seeded inputs and
seeded weights (tests/golden/make_golden.py::run_decode)."""
import torch
import torch.nn.functional as F

from dlt_oracle import heatmap_to_uv

FEAT_SIZE = (40, 80, 160, 320)          # POEM.py:55-56 (HRNet)
NUM_JOINTS = 21


def decoder_key_shapes():
    """state_dict keys (relative to the model) and shapes of the modules on this path (POEM.py:84-112)."""
    f = FEAT_SIZE
    ks = {}

    def block(name, cin, cout, k, norm):
        ks[f"{name}.conv.weight"] = (cout, cin, k, k)
        ks[f"{name}.conv.bias"] = (cout,)
        if norm:
            for n in ("weight", "bias", "running_mean", "running_var"):
                ks[f"{name}.norm.{n}"] = (cout,)

    for i in range(3):
        block(f"feat_delayer.{i}", f[i], f[i + 1], 3, True)                       # :84-88
    block("feat_in", f[3], f[2], 1, False)                                        # :89-94
    block("uv_delayer.0", f[3] + f[2], f[2], 3, True)                             # :101-108
    block("uv_delayer.1", f[2] + f[1], f[1], 3, True)
    block("uv_delayer.2", f[1] + f[0], f[0], 3, True)
    block("uv_out", f[0], NUM_JOINTS, 1, False)                                   # :110
    return ks


def seeded_decoder_state(seed=0):
    """Deterministic weights for fixtures / benches: conv weights N(0, sqrt(2 / fan_out)) as ConvBlock's
    kaiming_normal_(mode='fan_out') draws them, small random biases, and *non-trivial* BatchNorm statistics so that the
    folded affine is exercised (the reference initialises gamma = 1, beta = 0, mean = 0, var = 1)."""
    g = torch.Generator().manual_seed(1000 + seed)
    sd = {}
    for k, shp in decoder_key_shapes().items():
        if k.endswith("conv.weight"):
            fan_out = shp[0] * shp[2] * shp[3]
            sd[k] = torch.randn(shp, generator=g) * (2.0 / fan_out) ** 0.5
        elif k.endswith("conv.bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("norm.weight"):
            sd[k] = 1.0 + 0.2 * torch.randn(shp, generator=g)
        elif k.endswith("norm.bias"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(shp, generator=g)
    return sd


def synthetic_mlvl_feats(views, seed=0):
    """HRNet-shaped multi-level features of a 256x256 image (POEM.py:240-246): (BN,40,64,64) ... (BN,320,8,8)."""
    g = torch.Generator().manual_seed(2000 + seed)
    return [torch.randn(views, c, r, r, generator=g) for c, r in zip(FEAT_SIZE, (64, 32, 16, 8))]


def conv_block(x, sd, name, stride=1, relu=True, eps=1e-5):
    w = sd[f"{name}.conv.weight"]
    x = F.conv2d(x, w, sd[f"{name}.conv.bias"], stride=stride, padding=w.shape[-1] // 2)   # conv.py:18-23
    if f"{name}.norm.weight" in sd:                                                       # conv.py:24-25, eval mode
        x = F.batch_norm(x, sd[f"{name}.norm.running_mean"], sd[f"{name}.norm.running_var"], sd[f"{name}.norm.weight"],
                         sd[f"{name}.norm.bias"], training=False, eps=eps)
    return F.relu(x) if relu else x


def feat_decode(mlvl_feats, sd):
    """POEM.py:183-193 (HRNet branch): stride-2 ConvBlocks with lateral adds, bilinear x2, 1x1 conv."""
    x = mlvl_feats[0]
    for i in range(3):
        x = conv_block(x, sd, f"feat_delayer.{i}", stride=2) + mlvl_feats[i + 1]
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    return conv_block(x, sd, "feat_in", relu=False)


def uv_decode(mlvl_feats, sd):
    """POEM.py:197-207 (the unused uv_feat branch, :208, is not computed)."""
    rev = list(reversed(mlvl_feats))
    x = rev[0]
    for i in range(3):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        x = torch.cat((x, rev[i + 1]), dim=1)
        x = conv_block(x, sd, f"uv_delayer.{i}")
    x = F.max_pool2d(x, kernel_size=2, stride=2)
    return torch.sigmoid(conv_block(x, sd, "uv_out", relu=False))


def heatmap_stage(mlvl_feats, sd, W, H):
    """POEM.py:213-222: heat maps -> normalised expectation -> pixel coordinates (BN,21,2)."""
    return heatmap_to_uv(uv_decode(mlvl_feats, sd), W, H)
