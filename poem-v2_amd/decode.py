"""The convolutional glue between the backbone's multi-level features and the head, MI355X-native:
``feat_decode`` / ``uv_decode`` / ``heatmap_stage`` of the reference model (lib/models/POEM.py:167-222 upstream, HRNet
branch) on the HIP kernels of csrc/decode.hip.  No CPU fallback.

``FeatureDecoders`` holds what the reference model holds for this stage -- ``feat_delayer``, ``feat_in``,
``uv_delayer``, ``uv_out`` -- under the same state_dict key names, so a reference checkpoint's tensors load by key
(``load_reference_state_dict``; the unused ``uv_in`` and everything outside this stage are ignored).  Weights are packed
into MFMA fragment order once; conv bias and eval-mode BatchNorm fold into one per-channel affine applied in the conv
epilogue."""
import torch

from . import hip

FEAT_SIZE = (40, 80, 160, 320)
NUM_JOINTS = 21
BN_EPS = 1e-5


def _pad32(t):
    n = (t.numel() + 31) // 32 * 32
    out = torch.zeros(n, dtype=torch.float32, device=t.device)
    out[:t.numel()] = t
    return out


class _Conv3x3:
    def __init__(self, sd, name, device):
        w = sd[f"{name}.conv.weight"].to(device=device, dtype=torch.float32).contiguous()
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        nbytes = hip.lib().poem_conv3x3_packed_bytes(self.cout, self.cin)
        if nbytes == 0:
            raise RuntimeError(f"{name}: in_channels must be a multiple of 8")
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
        hip.check(hip.lib().poem_pack_conv3x3(hip.ptr(w), self.cout, self.cin, self.packed.data_ptr(), hip.stream()),
                  "poem_pack_conv3x3")
        f64 = lambda k: sd[k].to(device=device, dtype=torch.float64)   # noqa: E731
        bias = f64(f"{name}.conv.bias")
        if f"{name}.norm.weight" in sd:
            inv = f64(f"{name}.norm.weight") / torch.sqrt(f64(f"{name}.norm.running_var") + BN_EPS)
            shift = (bias - f64(f"{name}.norm.running_mean")) * inv + f64(f"{name}.norm.bias")
        else:
            inv, shift = torch.ones_like(bias), bias
        self.scale, self.shift = _pad32(inv.float()), _pad32(shift.float())

    def __call__(self, x_padded, h, w, stride, out, out_strides, residual=None, relu=True):
        views = x_padded.shape[0]
        ns, cs, rs, off = out_strides
        hip.check(hip.lib().poem_conv3x3(hip.ptr(x_padded), self.packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                                         hip.ptr(residual), hip.ptr(out), views, self.cin, self.cout, h, w, stride,
                                         int(relu), ns, cs, rs, off, hip.stream()), "poem_conv3x3")


def _conv_down2(self, x, h, w, out, out_strides, residual=None, relu=True):
    """stride-2 conv3x3 of the unbordered ``x`` in one LDS-staged launch (poem_conv3x3_down2); False when the shape is not taken."""
    ns, cs, rs, off = out_strides
    rc = hip.lib().poem_conv3x3_down2(hip.ptr(x), self.packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                                      hip.ptr(residual), hip.ptr(out), x.shape[0], self.cin, self.cout, h, w, int(relu), ns, cs, rs,
                                      off, hip.stream())
    if rc == hip.POEM_E_UNSUPPORTED:
        return False
    hip.check(rc, "poem_conv3x3_down2")
    return True


def _conv_upcat(self, a, b, h, w, out, out_strides, relu=True):
    """conv3x3 of [bilinear x2 of a | b] in one launch (poem_upcat_conv3x3); False when the shape is not taken."""
    views = b.shape[0]
    ns, cs, rs, off = out_strides
    rc = hip.lib().poem_upcat_conv3x3(hip.ptr(a), int(a.shape[1]), hip.ptr(b), int(b.shape[1]), self.packed.data_ptr(),
                                      hip.ptr(self.scale), hip.ptr(self.shift), hip.ptr(out), views, self.cout, h, w, int(relu),
                                      ns, cs, rs, off, hip.stream())
    if rc == hip.POEM_E_UNSUPPORTED:
        return False
    hip.check(rc, "poem_upcat_conv3x3")
    return True


_Conv3x3.upcat = _conv_upcat
_Conv3x3.down2 = _conv_down2


def _padded_strides(c, h, w):
    """strides of an (n, c, h+2, w+2) zero-bordered tensor addressed by interior (y, x)."""
    return (c * (h + 2) * (w + 2), (h + 2) * (w + 2), w + 2, (w + 2) + 1)


def _plain_strides(c, h, w):
    return (c * h * w, h * w, w, 0)


def upsample2_concat_pad(a, b, h, w, pad):
    """[bilinear x2 of a | b] with a zero border; a or b may be None."""
    ref = a if a is not None else b
    views = ref.shape[0]
    ca = 0 if a is None else int(a.shape[1])
    cb = 0 if b is None else int(b.shape[1])
    out = torch.empty(views, ca + cb, h + 2 * pad, w + 2 * pad, dtype=torch.float32, device=ref.device)
    hip.check(hip.lib().poem_upsample2_concat_pad(hip.ptr(a), ca, hip.ptr(b), cb, hip.ptr(out), views, h, w, pad,
                                                  hip.stream()), "poem_upsample2_concat_pad")
    return out


class FeatureDecoders:
    """feat_decode / uv_decode / heatmap_stage of the reference model for the HRNet feature pyramid."""

    def __init__(self, state_dict, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("FeatureDecoders runs on the MI355X HIP path only (no CPU fallback)")
        self.device = torch.device(device)
        self.fuse_upcat = [True, True, True]     # per uv_decode stage: poem_upcat_conv3x3 (one launch) vs upsample/concat + conv
        self.fuse_feat_in = True                 # feat_in + bilinear x2 in one launch (poem_conv1x1_upsample2)
        self.fuse_pool_head = True               # uv_decode's last stage + max-pool + uv_out + sigmoid in one launch
        sd = state_dict
        with torch.cuda.device(self.device):
            self.feat_delayer = [_Conv3x3(sd, f"feat_delayer.{i}", self.device) for i in range(3)]
            self.uv_delayer = [_Conv3x3(sd, f"uv_delayer.{i}", self.device) for i in range(3)]
            w = sd["feat_in.conv.weight"].to(device=self.device, dtype=torch.float32)
            self.feat_in_w = hip.pack_linear(w.reshape(w.shape[0], w.shape[1]).contiguous())
            self.feat_in_b = sd["feat_in.conv.bias"].to(device=self.device, dtype=torch.float32).contiguous()
            self.feat_in_out = int(w.shape[0])
            w = sd["uv_out.conv.weight"].to(device=self.device, dtype=torch.float32)
            self.uv_out_w = w.reshape(w.shape[0], w.shape[1]).contiguous()
            self.uv_out_b = sd["uv_out.conv.bias"].to(device=self.device, dtype=torch.float32).contiguous()

    @staticmethod
    def live_keys():
        keys = []
        for blk in [f"feat_delayer.{i}" for i in range(3)] + [f"uv_delayer.{i}" for i in range(3)]:
            keys += [f"{blk}.conv.weight", f"{blk}.conv.bias"] + [f"{blk}.norm.{n}" for n in
                                                                   ("weight", "bias", "running_mean", "running_var")]
        return keys + ["feat_in.conv.weight", "feat_in.conv.bias", "uv_out.conv.weight", "uv_out.conv.bias"]

    @classmethod
    def load_reference_state_dict(cls, state_dict, device="cuda:0", prefix=""):
        """Pick this stage's tensors out of a full-model reference checkpoint (everything else is ignored)."""
        missing = [k for k in cls.live_keys() if prefix + k not in state_dict]
        if missing:
            raise KeyError(f"checkpoint lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        return cls({k: state_dict[prefix + k] for k in cls.live_keys()}, device)

    def _check(self, mlvl_feats):
        if len(mlvl_feats) != 4:
            raise ValueError("expected the four HRNet levels")
        out = []
        for f, c in zip(mlvl_feats, FEAT_SIZE):
            if not f.is_cuda:
                raise RuntimeError("FeatureDecoders operates on device tensors only (no CPU path)")
            if f.shape[1] != c:
                raise ValueError(f"level with {f.shape[1]} channels, expected {c}")
            out.append(f.to(dtype=torch.float32).contiguous())
        return out

    def feat_decode(self, mlvl_feats, backbone="HRNet"):
        """(BN,40,64,64),(BN,80,32,32),(BN,160,16,16),(BN,320,8,8) -> mlvl_feat (BN,160,16,16)   [POEM.py:183-193]"""
        assert backbone == "HRNet", "only the HRNet branch is built"
        f = self._check(mlvl_feats)
        views, r = f[0].shape[0], f[0].shape[-1]
        with torch.cuda.device(self.device):
            x, bordered = f[0], False
            for i, conv in enumerate(self.feat_delayer):
                ro = r // 2
                out = torch.empty(views, conv.cout, ro, ro, dtype=torch.float32, device=self.device)
                if bordered or not conv.down2(x, r, r, out, _plain_strides(conv.cout, ro, ro), residual=f[i + 1]):
                    # shapes the LDS-staged stride-2 kernel does not take: the direct kernel over zero-bordered tensors
                    if not bordered:
                        x = upsample2_concat_pad(None, x, r, r, 1)
                    if i == 2:
                        strides = _plain_strides(conv.cout, ro, ro)
                    else:                                                       # lands inside the next conv's input
                        out = torch.zeros(views, conv.cout, ro + 2, ro + 2, dtype=torch.float32, device=self.device)
                        strides = _padded_strides(conv.cout, ro, ro)
                    conv(x, r, r, 2, out, strides, residual=f[i + 1])
                    bordered = i < 2
                x, r = out, ro
            # feat_in is a 1x1 convolution: it commutes with the bilinear x2 in front of it (both linear, the interpolation
            # weights sum to one, so the bias passes through) -- applied at the low resolution it is a quarter of the FLOPs
            # and the (BN,320,16,16) intermediate never exists (POEM.py:190-193 upstream: interpolate, then feat_in)
            hw = r * r
            y = torch.empty(views, self.feat_in_out, 2 * r, 2 * r, dtype=torch.float32, device=self.device)
            rc = hip.lib().poem_conv1x1_upsample2(hip.ptr(x), self.feat_in_w.data_ptr(), hip.ptr(self.feat_in_b), hip.ptr(y),
                                                  views, int(x.shape[1]), self.feat_in_out, r, r, hip.stream()) \
                if self.fuse_feat_in else hip.POEM_E_UNSUPPORTED
            if rc == hip.POEM_E_UNSUPPORTED:                                     # other pyramid sizes: two launches
                y8 = torch.empty(views, self.feat_in_out, r, r, dtype=torch.float32, device=self.device)
                hip.check(hip.lib().poem_input_proj(hip.ptr(x), self.feat_in_w.data_ptr(), hip.ptr(self.feat_in_b), None, None,
                                                    hip.ptr(y8), views, int(x.shape[1]), self.feat_in_out, hw, hip.stream()),
                          "poem_input_proj")
                y = upsample2_concat_pad(y8, None, 2 * r, 2 * r, 0)              # (BN,160,16,16)
            else:
                hip.check(rc, "poem_conv1x1_upsample2")
        return y

    def uv_decode(self, mlvl_feats):
        """-> uv_hmap (BN,21,32,32)   [POEM.py:197-207; the unused uv_feat of :208 is not computed]"""
        f = self._check(mlvl_feats)
        rev = list(reversed(f))
        views = f[0].shape[0]
        with torch.cuda.device(self.device):
            x, r = rev[0], rev[0].shape[-1]
            for i, conv in enumerate(self.uv_delayer):
                r *= 2
                if i == 2 and self.fuse_pool_head and self.fuse_upcat[i]:
                    # the last stage and the read-out head in one launch: its (BN,40,64,64) output is read by nothing else
                    hm = torch.empty(views, NUM_JOINTS, r // 2, r // 2, dtype=torch.float32, device=self.device)
                    rc = hip.lib().poem_upcat_conv3x3_pool_head(
                        hip.ptr(x), int(x.shape[1]), hip.ptr(rev[i + 1]), int(rev[i + 1].shape[1]), conv.packed.data_ptr(),
                        hip.ptr(conv.scale), hip.ptr(conv.shift), hip.ptr(self.uv_out_w), hip.ptr(self.uv_out_b), hip.ptr(hm), views,
                        conv.cout, NUM_JOINTS, r, r, 1, hip.stream())
                    if rc != hip.POEM_E_UNSUPPORTED:
                        hip.check(rc, "poem_upcat_conv3x3_pool_head")
                        return hm
                y = torch.empty(views, conv.cout, r, r, dtype=torch.float32, device=self.device)
                if not (self.fuse_upcat[i] and conv.upcat(x, rev[i + 1], r, r, y, _plain_strides(conv.cout, r, r))):   # one launch where it pays
                    conv(upsample2_concat_pad(x, rev[i + 1], r, r, 1), r, r, 1, y, _plain_strides(conv.cout, r, r))
                x = y
            hm = torch.empty(views, NUM_JOINTS, r // 2, r // 2, dtype=torch.float32, device=self.device)
            hip.check(hip.lib().poem_pool_conv1x1_sigmoid(hip.ptr(x), hip.ptr(self.uv_out_w), hip.ptr(self.uv_out_b),
                                                          hip.ptr(hm), views, int(x.shape[1]), NUM_JOINTS, r, r,
                                                          hip.stream()), "poem_pool_conv1x1_sigmoid")
        return hm

    def heatmap_stage(self, img_feats, W, H):
        """-> uv_coord_im (BN,21,2) pixels   [POEM.py:213-222]"""
        from .triangulation import heatmap_to_uv
        return heatmap_to_uv(self.uv_decode(img_feats), W, H)
