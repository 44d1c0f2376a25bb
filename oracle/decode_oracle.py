"""CPU oracle for the convolutional glue in front of the head.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Restates ``PtEmbedMultiviewStereoV2.feat_decode`` / ``uv_decode`` / ``heatmap_stage`` (lib/models/POEM.py:167-222
upstream, HRNet branch) with ``ConvBlock`` = Conv2d -> BatchNorm2d(eval) -> ReLU (lib/models/bricks/conv.py:4-45) in
plain torch on CPU.  PINNED: tests/golden/decode.npz holds outputs of the reference's own methods on seeded inputs and
seeded weights (tests/golden/make_golden.py::run_decode)."""
import torch
import torch.nn.functional as F

from dlt_oracle import heatmap_to_uv

from poem_v2_amd.decode import FEAT_SIZE, NUM_JOINTS  # noqa: F401
from poem_v2_amd.weights import decoder_key_shapes, seeded_decoder_state_dict as seeded_decoder_state  # noqa: F401
from poem_v2_amd.inputs import synthetic_pyramid as synthetic_mlvl_feats  # noqa: F401  (seeded generators shared with the product side)


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
