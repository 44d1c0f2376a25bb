"""Convolutional glue in front of the head (feat_decode / uv_decode / heatmap_stage): oracle pinned to the
reference model's own outputs (CPU), HIP kernels vs both (GPU, through the C ABI)."""
import json
import os

import numpy as np
import pytest
import torch

import decode_oracle as do
from util import GOLDEN

DEV = "cuda:0"


def _md(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def _golden():
    z = np.load(os.path.join(GOLDEN, "decode.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def test_oracle_matches_reference_model_outputs():
    z, meta = _golden()
    sd = do.seeded_decoder_state(meta["seed"])
    feats = do.synthetic_mlvl_feats(meta["views"], meta["seed"])
    mlvl = do.feat_decode(feats, sd)
    assert tuple(mlvl.shape) == (meta["views"], 160, 16, 16)
    assert _md(mlvl[:, ::4], torch.from_numpy(z["mlvl_feat_s4"])) < 1e-5 * float(z["mlvl_feat_absmax"])
    hm = do.uv_decode(feats, sd)
    assert tuple(hm.shape) == (meta["views"], 21, 32, 32)
    assert _md(hm[:, ::3], torch.from_numpy(z["uv_hmap_s3"])) < 1e-6
    assert _md(do.heatmap_stage(feats, sd, 256, 256), torch.from_numpy(z["uv"])) < 1e-4      # pixels


def test_decoder_key_table_matches_what_the_fixture_loaded():
    ks = do.decoder_key_shapes()
    assert ks["uv_delayer.0.conv.weight"] == (160, 480, 3, 3) and ks["feat_delayer.2.conv.weight"] == (320, 160, 3, 3)
    assert ks["feat_in.conv.weight"] == (160, 320, 1, 1) and ks["uv_out.conv.weight"] == (21, 40, 1, 1)
    import poem_v2_amd as pk
    assert set(pk.decode.FeatureDecoders.live_keys()) == set(ks)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,r,stride,relu,res", [(40, 80, 64, 2, True, True), (480, 160, 16, 1, True, False),
                                                         (120, 40, 64, 1, True, False), (16, 33, 8, 1, False, False),
                                                         (160, 320, 16, 2, True, True)])
def test_conv3x3_operator(cin, cout, r, stride, relu, res):
    import torch.nn.functional as F
    import poem_v2_amd as pk
    from poem_v2_amd import hip
    g = torch.Generator().manual_seed(cin + cout)
    views = 2
    x = torch.randn(views, cin, r, r, generator=g)
    sd = {"c.conv.weight": torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5),
          "c.conv.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.weight": 1 + 0.2 * torch.randn(cout, generator=g),
          "c.norm.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.running_mean": 0.1 * torch.randn(cout, generator=g),
          "c.norm.running_var": 0.5 + torch.rand(cout, generator=g)}
    ro = r // stride
    lateral = torch.randn(views, cout, ro, ro, generator=g) if res else None
    ref = do.conv_block(x.double(), {k: v.double() for k, v in sd.items()}, "c", stride=stride, relu=relu)
    if res:
        ref = ref + lateral.double()
    conv = pk.decode._Conv3x3(sd, "c", torch.device(DEV))
    xp = pk.decode.upsample2_concat_pad(None, x.to(DEV), r, r, 1)
    assert _md(xp[:, :, 1:-1, 1:-1], x) == 0.0 and float(xp[:, :, 0].abs().max()) == 0.0
    # plain output
    out = torch.empty(views, cout, ro, ro, device=DEV)
    conv(xp, r, r, stride, out, pk.decode._plain_strides(cout, ro, ro), residual=None if lateral is None else lateral.to(DEV),
         relu=relu)
    assert _md(out, ref) < 2e-5
    # output written into a zero-bordered buffer (the next conv's input)
    outp = torch.zeros(views, cout, ro + 2, ro + 2, device=DEV)
    conv(xp, r, r, stride, outp, pk.decode._padded_strides(cout, ro, ro), residual=None if lateral is None else lateral.to(DEV),
         relu=relu)
    assert _md(outp[:, :, 1:-1, 1:-1], ref) < 2e-5 and float(outp[:, :, :, 0].abs().max()) == 0.0
    hip.lib()


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,r", [(40, 80, 64), (80, 160, 32), (160, 320, 16), (24, 80, 64), (40, 96, 64)])
def test_conv3x3_down2_operator(cin, cout, r):
    """poem_conv3x3_down2 (a stride-2 ConvBlock of feat_decode from the unbordered input, LDS-staged, 16-channel tiles)
    against the ConvBlock in fp64 and against the direct kernel over a zero-bordered copy -- HRNet-W40's three shapes with
    lateral add, a second input width, and a shape it does not take (the entry point says so)."""
    import poem_v2_amd as pk
    g = torch.Generator().manual_seed(cin + cout + r)
    views = 3
    x = torch.randn(views, cin, r, r, generator=g)
    sd = {"c.conv.weight": torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5),
          "c.conv.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.weight": 1 + 0.2 * torch.randn(cout, generator=g),
          "c.norm.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.running_mean": 0.1 * torch.randn(cout, generator=g),
          "c.norm.running_var": 0.5 + torch.rand(cout, generator=g)}
    ro = r // 2
    lateral = torch.randn(views, cout, ro, ro, generator=g)
    ref = do.conv_block(x.double(), {k: v.double() for k, v in sd.items()}, "c", stride=2, relu=True) + lateral.double()
    conv = pk.decode._Conv3x3(sd, "c", torch.device(DEV))
    out = torch.full((views, cout, ro, ro), float("nan"), device=DEV)
    taken = conv.down2(x.to(DEV), r, r, out, pk.decode._plain_strides(cout, ro, ro), residual=lateral.to(DEV))
    assert taken == (cout != 96)
    direct = torch.empty(views, cout, ro, ro, device=DEV)
    conv(pk.decode.upsample2_concat_pad(None, x.to(DEV), r, r, 1), r, r, 2, direct, pk.decode._plain_strides(cout, ro, ro),
         residual=lateral.to(DEV))
    assert _md(direct, ref) < 2e-5
    if taken:
        assert _md(out, ref) < 2e-5 and _md(out, direct) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,r,views", [(40, 80, 64, 200), (80, 160, 32, 300), (160, 320, 16, 500), (24, 80, 64, 3)])
def test_conv3x3_down2_staging_wave_is_bit_identical_to_the_round3_kernel(cin, cout, r, views):
    """poem_conv3x3_down2's default kernel (persistent blocks of four MFMA waves + a staging wave, four-row tiles) against the
    round-3 kernel (`s2_staging_wave` 0: every wave stages and multiplies, eight-row tiles) -- same summation order, so
    bit-identical -- at view counts where a block walks several tiles (more tiles than 3 x 256 block slots) and where the
    last blocks have one tile fewer, with and without the lateral add."""
    import poem_v2_amd as pk
    from poem_v2_amd import hip
    g = torch.Generator().manual_seed(cin + cout + r + views)
    x = torch.randn(views, cin, r, r, generator=g).to(DEV)
    sd = {"c.conv.weight": torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5),
          "c.conv.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.weight": 1 + 0.2 * torch.randn(cout, generator=g),
          "c.norm.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.running_mean": 0.1 * torch.randn(cout, generator=g),
          "c.norm.running_var": 0.5 + torch.rand(cout, generator=g)}
    ro = r // 2
    lateral = torch.randn(views, cout, ro, ro, generator=g).to(DEV)
    conv = pk.decode._Conv3x3(sd, "c", torch.device(DEV))
    try:
        for res in (lateral, None):
            outs = []
            for on in (1, 0):
                hip.check(hip.lib().poem_set_decode_option(b"s2_staging_wave", on), "poem_set_decode_option")
                out = torch.full((views, cout, ro, ro), float("nan"), device=DEV)
                assert conv.down2(x, r, r, out, pk.decode._plain_strides(cout, ro, ro), residual=res)
                outs.append(out)
            assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())
            if views <= 8:
                # the same result written into a zero-bordered buffer (view / channel / row strides and offset of the entry
                # point): interior == the plain result, border untouched
                hip.check(hip.lib().poem_set_decode_option(b"s2_staging_wave", 1), "poem_set_decode_option")
                outp = torch.zeros(views, cout, ro + 2, ro + 2, device=DEV)
                assert conv.down2(x, r, r, outp, pk.decode._padded_strides(cout, ro, ro), residual=res)
                assert torch.equal(outp[:, :, 1:-1, 1:-1], outs[0])
                assert float(outp[:, :, 0].abs().max()) == 0.0 and float(outp[:, :, :, 0].abs().max()) == 0.0
                assert float(outp[:, :, -1].abs().max()) == 0.0 and float(outp[:, :, :, -1].abs().max()) == 0.0
    finally:
        hip.lib().poem_set_decode_option(b"s2_staging_wave", 1)
    assert hip.lib().poem_set_decode_option(b"no_such_switch", 1) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,r", [(320, 160, 8), (64, 32, 8), (40, 32, 16)])
def test_conv1x1_upsample2_operator(cin, cout, r):
    """poem_conv1x1_upsample2 (feat_decode's tail in one launch: feat_in applied at the low resolution, result upsampled
    from LDS) against the reference's order -- F.interpolate x2 THEN the 1x1 convolution, POEM.py:190-193 -- in fp64, and
    against the two-launch sequence (poem_input_proj + poem_upsample2_concat_pad); a size it does not take says so."""
    import torch.nn.functional as F
    import poem_v2_amd as pk
    from poem_v2_amd import hip
    g = torch.Generator().manual_seed(cin + r)
    views = 5
    x = torch.randn(views, cin, r, r, generator=g)
    w, b = torch.randn(cout, cin, generator=g) / cin ** 0.5, 0.1 * torch.randn(cout, generator=g)
    up = F.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=False)
    ref = torch.einsum("oc,vchw->vohw", w.double(), up) + b.double()[None, :, None, None]
    wp = hip.pack_linear(w.to(DEV))
    y8 = torch.empty(views, cout, r, r, device=DEV)
    xd, bd = x.to(DEV), b.to(DEV)       # (named: a temporary's block goes back to the allocator while the launch still reads it)
    hip.check(hip.lib().poem_input_proj(hip.ptr(xd), wp.data_ptr(), hip.ptr(bd), None, None, hip.ptr(y8), views, cin,
                                        cout, r * r, hip.stream()), "poem_input_proj")
    two = pk.decode.upsample2_concat_pad(y8, None, 2 * r, 2 * r, 0)
    assert _md(two, ref) < 2e-5
    one = torch.full((views, cout, 2 * r, 2 * r), float("nan"), device=DEV)
    rc = hip.lib().poem_conv1x1_upsample2(hip.ptr(xd), wp.data_ptr(), hip.ptr(bd), hip.ptr(one), views, cin, cout, r, r, hip.stream())
    if r != 8:
        assert rc == hip.POEM_E_UNSUPPORTED
        return
    hip.check(rc, "poem_conv1x1_upsample2")
    assert _md(one, ref) < 2e-5 and _md(one, two) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("ca,cb,cout,r", [(320, 160, 160, 16), (160, 80, 80, 32), (80, 40, 40, 64), (16, 8, 33, 16), (24, 16, 200, 16)])
def test_upcat_conv3x3_matches_the_two_launch_sequence_and_torch(ca, cb, cout, r):
    """poem_upcat_conv3x3 (one uv_decode stage in one launch: bilinear x2 | concat | zero border staged straight into the
    LDS-staged convolution) against F.interpolate + torch.cat + the ConvBlock in fp64, and against the two-launch sequence
    (poem_upsample2_concat_pad + poem_conv3x3) it replaces -- the three uv_decode shapes, a channel-tail shape, and a shape
    the fused kernel does not take (cout > 160: the entry point says so and the caller falls back)."""
    import torch.nn.functional as F
    import poem_v2_amd as pk
    g = torch.Generator().manual_seed(ca + cout)
    views = 3
    a, b = torch.randn(views, ca, r // 2, r // 2, generator=g), torch.randn(views, cb, r, r, generator=g)
    cin = ca + cb
    sd = {"c.conv.weight": torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5),
          "c.conv.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.weight": 1 + 0.2 * torch.randn(cout, generator=g),
          "c.norm.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.running_mean": 0.1 * torch.randn(cout, generator=g),
          "c.norm.running_var": 0.5 + torch.rand(cout, generator=g)}
    x = torch.cat((F.interpolate(a.double(), scale_factor=2, mode="bilinear", align_corners=False), b.double()), dim=1)
    ref = do.conv_block(x, {k: v.double() for k, v in sd.items()}, "c", stride=1, relu=True)
    conv = pk.decode._Conv3x3(sd, "c", torch.device(DEV))
    two = torch.empty(views, cout, r, r, device=DEV)
    conv(pk.decode.upsample2_concat_pad(a.to(DEV), b.to(DEV), r, r, 1), r, r, 1, two, pk.decode._plain_strides(cout, r, r))
    one = torch.full((views, cout, r, r), float("nan"), device=DEV)
    taken = conv.upcat(a.to(DEV), b.to(DEV), r, r, one, pk.decode._plain_strides(cout, r, r))
    assert taken == (cout <= 160)
    assert _md(two, ref) < 2e-5
    if taken:
        assert _md(one, ref) < 2e-5 and _md(one, two) < 2e-5
        # ... and into a zero-bordered buffer, as the direct kernel can
        onep = torch.zeros(views, cout, r + 2, r + 2, device=DEV)
        assert conv.upcat(a.to(DEV), b.to(DEV), r, r, onep, pk.decode._padded_strides(cout, r, r))
        assert _md(onep[:, :, 1:-1, 1:-1], ref) < 2e-5 and float(onep[:, :, 0].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("ca,cb,cout,views,r", [(80, 40, 40, 5, 64), (16, 8, 40, 2, 64), (8, 16, 80, 2, 64), (80, 40, 40, 40, 64),
                                                (160, 80, 80, 5, 32), (8, 8, 40, 3, 32), (160, 80, 80, 70, 32),
                                                (320, 160, 160, 9, 16), (16, 8, 160, 3, 16)])
def test_upcat_conv3x3_row_stager_is_bit_identical_to_the_per_float_stager(ca, cb, cout, views, r):
    """The fused-input convolutions at 64 x 64 and 32 x 32 (uv_decode's last two stages, 120 -> 40 and 240 -> 80) staged by
    rows -- wave = channel, lane = (row group, column), the source rows of a chunk loaded once, two blocks per CU
    (`row_stager` 3, the default) -- against the per-float stager they replace (`row_stager` 0): every staged value is the
    same expression, so the results are bit-identical; top and bottom tiles (clamped source rows, rows outside the image), few
    and many views, other channel splits.  At 16 x 16 (the first stage, 480 -> 160 on 32-channel tiles) the switch under test
    is `pin32`: the pinned tap pipeline against the compiler-scheduled taps -- same summation order."""
    import poem_v2_amd as pk
    from poem_v2_amd import hip
    g = torch.Generator().manual_seed(ca + 7 * cb + cout + views + r)
    a, b = torch.randn(views, ca, r // 2, r // 2, generator=g).to(DEV), torch.randn(views, cb, r, r, generator=g).to(DEV)
    cin = ca + cb
    sd = {"c.conv.weight": torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5),
          "c.conv.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.weight": 1 + 0.2 * torch.randn(cout, generator=g),
          "c.norm.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.running_mean": 0.1 * torch.randn(cout, generator=g),
          "c.norm.running_var": 0.5 + torch.rand(cout, generator=g)}
    conv = pk.decode._Conv3x3(sd, "c", torch.device(DEV))
    outs = []
    try:
        for on in (1, 0):
            hip.check(hip.lib().poem_set_decode_option(b"row_stager", 3 * on), "poem_set_decode_option")
            hip.check(hip.lib().poem_set_decode_option(b"pin32", on), "poem_set_decode_option")
            out = torch.full((views, cout, r, r), float("nan"), device=DEV)
            assert conv.upcat(a, b, r, r, out, pk.decode._plain_strides(cout, r, r))
            outs.append(out)
    finally:
        hip.lib().poem_set_decode_option(b"row_stager", 3)
        hip.lib().poem_set_decode_option(b"pin32", 1)
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("ca,cb,views", [(80, 40, 5), (16, 8, 2), (80, 40, 37)])
def test_uv_decode_last_stage_with_the_read_out_head_fused_is_bit_identical(ca, cb, views):
    """poem_upcat_conv3x3_pool_head: uv_decode's last convolution (120 -> 40 at 64 x 64) with max_pool2d(2, 2) + uv_out +
    sigmoid in its epilogue (POEM.py:203-207) against the two launches it replaces -- the same affine / ReLU, the same maxima,
    the 40-term contraction as the same fma chain: bit-identical heat maps; the whole decoder with the switch on and off."""
    import poem_v2_amd as pk
    from poem_v2_amd import hip
    g = torch.Generator().manual_seed(3 * ca + cb + views)
    r, cout, J = 64, 40, 21
    a, b = torch.randn(views, ca, r // 2, r // 2, generator=g).to(DEV), torch.randn(views, cb, r, r, generator=g).to(DEV)
    cin = ca + cb
    sd = {"c.conv.weight": torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5),
          "c.conv.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.weight": 1 + 0.2 * torch.randn(cout, generator=g),
          "c.norm.bias": 0.1 * torch.randn(cout, generator=g), "c.norm.running_mean": 0.1 * torch.randn(cout, generator=g),
          "c.norm.running_var": 0.5 + torch.rand(cout, generator=g)}
    conv = pk.decode._Conv3x3(sd, "c", torch.device(DEV))
    hw, hb = (0.3 * torch.randn(J, cout, generator=g)).to(DEV), (0.1 * torch.randn(J, generator=g)).to(DEV)
    y = torch.empty(views, cout, r, r, device=DEV)
    assert conv.upcat(a, b, r, r, y, pk.decode._plain_strides(cout, r, r))
    two = torch.empty(views, J, r // 2, r // 2, device=DEV)
    hip.check(hip.lib().poem_pool_conv1x1_sigmoid(hip.ptr(y), hip.ptr(hw), hip.ptr(hb), hip.ptr(two), views, cout, J, r, r, hip.stream()))
    one = torch.full((views, J, r // 2, r // 2), float("nan"), device=DEV)
    hip.check(hip.lib().poem_upcat_conv3x3_pool_head(hip.ptr(a), ca, hip.ptr(b), cb, conv.packed.data_ptr(), hip.ptr(conv.scale),
                                                     hip.ptr(conv.shift), hip.ptr(hw), hip.ptr(hb), hip.ptr(one), views, cout, J, r, r, 1,
                                                     hip.stream()), "poem_upcat_conv3x3_pool_head")
    assert torch.equal(one, two) and bool(torch.isfinite(one).all())
    ref = torch.sigmoid(torch.nn.functional.conv2d(torch.nn.functional.max_pool2d(y, 2, 2), hw[:, :, None, None], hb))
    assert _md(one, ref) < 1e-6
    # shapes it does not take are refused, not mangled
    assert hip.lib().poem_upcat_conv3x3_pool_head(hip.ptr(a), ca, hip.ptr(b), cb, conv.packed.data_ptr(), hip.ptr(conv.scale),
                                                  hip.ptr(conv.shift), hip.ptr(hw), hip.ptr(hb), hip.ptr(one), views, cout, J, 32, 32, 1,
                                                  hip.stream()) == hip.POEM_E_UNSUPPORTED
    if (ca, cb) == (80, 40):
        sdd = do.seeded_decoder_state(3)
        feats = [f.to(DEV) for f in do.synthetic_mlvl_feats(views, 3)]
        dec = pk.decode.FeatureDecoders(sdd, DEV)
        hm1 = dec.uv_decode(feats)
        dec.fuse_pool_head = False
        assert torch.equal(hm1, dec.uv_decode(feats))
        try:
            hip.check(hip.lib().poem_set_decode_option(b"pool_fused", 0), "poem_set_decode_option")
            dec.fuse_pool_head = True
            assert torch.equal(hm1, dec.uv_decode(feats))          # refused -> the two-launch form
        finally:
            hip.lib().poem_set_decode_option(b"pool_fused", 1)


@pytest.mark.gpu
def test_upsample_concat_matches_torch():
    import torch.nn.functional as F
    import poem_v2_amd as pk
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(2, 24, 8, 8, generator=g), torch.randn(2, 16, 16, 16, generator=g)
    ref = torch.cat((F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=False), b), dim=1)
    got = pk.decode.upsample2_concat_pad(a.to(DEV), b.to(DEV), 16, 16, 1)
    assert _md(got[:, :, 1:-1, 1:-1], ref) < 1e-6
    got0 = pk.decode.upsample2_concat_pad(a.to(DEV), None, 16, 16, 0)
    assert _md(got0, ref[:, :24]) < 1e-6


@pytest.mark.gpu
def test_decoders_match_oracle_and_reference_fixture():
    import poem_v2_amd as pk
    z, meta = _golden()
    sd = do.seeded_decoder_state(meta["seed"])
    feats = do.synthetic_mlvl_feats(meta["views"], meta["seed"])
    dec = pk.decode.FeatureDecoders(sd, DEV)
    dfe = [f.to(DEV) for f in feats]
    mlvl = dec.feat_decode(dfe)
    scale = float(z["mlvl_feat_absmax"])
    assert _md(mlvl, do.feat_decode(feats, sd)) < 2e-5 * scale
    assert _md(mlvl[:, ::4], torch.from_numpy(z["mlvl_feat_s4"])) < 2e-5 * scale
    hm = dec.uv_decode(dfe)
    assert _md(hm, do.uv_decode(feats, sd)) < 2e-5          # three chained K <= 4320 contractions in front of a sigmoid
    assert _md(hm[:, ::3], torch.from_numpy(z["uv_hmap_s3"])) < 2e-5
    uv = dec.heatmap_stage(dfe, 256, 256)
    assert _md(uv, torch.from_numpy(z["uv"])) < 5e-4                                         # pixels
    # the four inputs are not modified (the reference methods do not mutate them either)
    for a, b in zip(dfe, feats):
        assert _md(a, b) == 0.0


@pytest.mark.gpu
def test_decoders_feed_the_triangulation():
    """heat maps -> uv -> ragged DLT: the chained stage in front of the head on one stream."""
    import poem_v2_amd as pk
    sd = do.seeded_decoder_state(1)
    views = [2, 3]
    feats = [f.to(DEV) for f in do.synthetic_mlvl_feats(sum(views), 1)]
    dec = pk.decode.FeatureDecoders(sd, DEV)
    b = pk.inputs.synthetic_batch(views, seed=1)
    m = b["img_metas"]
    uv = dec.heatmap_stage(feats, 256, 256)
    rj = pk.triangulation.triangulate_reference_joints(uv, m["cam_intr"].to(DEV), m["cam_extr"].to(DEV), views)
    import dlt_oracle as dl
    ref = dl.triangulate_reference_joints(do.heatmap_stage([f.cpu() for f in feats], sd, 256, 256), m["cam_intr"],
                                          m["cam_extr"], views)
    assert torch.isfinite(rj).all() and _md(rj, ref) < 1e-3      # metres; near-degenerate rays amplify the 1e-4 px


def test_errors_are_loud_on_cpu_tensors():
    import poem_v2_amd as pk
    with pytest.raises(RuntimeError):
        pk.decode.upsample2_concat_pad(None, torch.zeros(1, 8, 4, 4), 4, 4, 1)
