"""HRNet-W40 feature pyramid on plain PyTorch-ROCm -- plumbing for the END-TO-END timing scope only (SURVEY.md 8d:
"E2E: 256x256 images -> verts, HRNet on PyTorch-ROCm").  The backbone is NOT part of the hot path and gets no HIP
kernels of its own; it exists so that ``bench.py --e2e`` and ``PtEmbedMultiviewStereoV2`` can
start from images the way the reference model does (lib/models/POEM.py:225-266 upstream).

Behaviour follows the reference's trimmed classification HRNet (lib/models/backbones/hrnet.py:240-420: stem of two
stride-2 3x3 convs, a Bottleneck stage, then 1 / 4 / 3 HighResolutionModules with 2 / 3 / 4 branches, the
classification head constructed but never run) with the widths of config/backbone/cls_hrnet_w40_*.yaml.  The state_dict
key names are the reference's (``conv1``, ``bn1``, ``layer1.N.*``, ``transitionS.I.*``, ``stageS.M.branches.B.K.*``,
``stageS.M.fuse_layers.I.J.*``), so a checkpoint's ``img_backbone.*`` tensors load by key; the dead classification
head (``incre_modules`` / ``downsamp_modules`` / ``final_layer`` / ``classifier``) is ignored.

Written as a flat list of folded convolutions instead of an nn.Module tree: every eval-mode BatchNorm is folded into
the convolution in front of it once at load time (fp64), so a forward is conv (+bias) -> [add] -> [ReLU] calls only.
Runs on whatever device its tensors live on (CPU for the parity test, MIOpen on the GPU)."""
import torch
import torch.nn.functional as F

from .builder import BACKBONE

BN_EPS = 1e-5
WIDTHS = (40, 80, 160, 320)                     # cls_hrnet_w40 NUM_CHANNELS of stage 4
STAGES = ((2, 1), (3, 4), (4, 3))               # (branches, modules) of stages 2..4; 4 BasicBlocks per branch
BLOCKS_PER_BRANCH = 4


def _conv_specs(widths=WIDTHS):
    """(conv key, bn key, cout, cin, kernel, stride) of every live conv -> BatchNorm pair, in the reference's key names."""
    specs = []

    def conv_bn(conv, bn, cout, cin, k, stride=1):
        specs.append((conv, bn, cout, cin, k, stride))

    conv_bn("conv1", "bn1", 64, 3, 3, 2)
    conv_bn("conv2", "bn2", 64, 64, 3, 2)
    cin = 64
    for i in range(4):                                             # layer1: Bottleneck x4, planes 64, expansion 4
        p = f"layer1.{i}"
        conv_bn(f"{p}.conv1", f"{p}.bn1", 64, cin, 1)
        conv_bn(f"{p}.conv2", f"{p}.bn2", 64, 64, 3)
        conv_bn(f"{p}.conv3", f"{p}.bn3", 256, 64, 1)
        if i == 0:
            conv_bn(f"{p}.downsample.0", f"{p}.downsample.1", 256, cin, 1)
        cin = 256
    pre = [256]
    for s, (nb, nm) in enumerate(STAGES, start=1):
        cur = list(widths[:nb])
        for i in range(nb):                                        # transition s (hrnet.py:319-344)
            if i < len(pre):
                if cur[i] != pre[i]:
                    conv_bn(f"transition{s}.{i}.0", f"transition{s}.{i}.1", cur[i], pre[i], 3)
            else:
                for j in range(i + 1 - len(pre)):
                    cout = cur[i] if j == i - len(pre) else pre[-1]
                    conv_bn(f"transition{s}.{i}.{j}.0", f"transition{s}.{i}.{j}.1", cout, pre[-1], 3, 2)
        for m in range(nm):
            p = f"stage{s + 1}.{m}"
            for b in range(nb):
                for k in range(BLOCKS_PER_BRANCH):
                    q = f"{p}.branches.{b}.{k}"
                    conv_bn(f"{q}.conv1", f"{q}.bn1", cur[b], cur[b], 3)
                    conv_bn(f"{q}.conv2", f"{q}.bn2", cur[b], cur[b], 3)
            for i in range(nb):                                    # fuse layers (hrnet.py:177-212)
                for j in range(nb):
                    if j > i:
                        conv_bn(f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", cur[i], cur[j], 1)
                    elif j < i:
                        for k in range(i - j):
                            cout = cur[i] if k == i - j - 1 else cur[j]
                            conv_bn(f"{p}.fuse_layers.{i}.{j}.{k}.0", f"{p}.fuse_layers.{i}.{j}.{k}.1", cout, cur[j], 3, 2)
        pre = cur
    return specs


def hrnet_param_shapes(widths=WIDTHS):
    """name -> shape of every live tensor (conv weights + BatchNorm affine / statistics)."""
    shapes = {}
    for conv, bn, cout, cin, k, _ in _conv_specs(widths):
        shapes[f"{conv}.weight"] = (cout, cin, k, k)
        for n in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{bn}.{n}"] = (cout,)
    return shapes


def seeded_hrnet_state_dict(seed=0, widths=WIDTHS):
    """Seeded weights with non-trivial BatchNorm statistics; the second conv of every residual block is small so that
    activations stay O(1) through the ~100 layers without trained statistics."""
    g = torch.Generator().manual_seed(7000 + seed)
    sd = {}
    for name, shape in hrnet_param_shapes(widths).items():
        if name.endswith("running_var"):
            sd[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1 and name.endswith(".weight"):
            sd[name] = 1.0 + 0.2 * (torch.rand(shape, generator=g) * 2 - 1)
        elif len(shape) == 1:
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if (".conv2.weight" in name and "branches" in name) or ".conv3.weight" in name:
                gain = 0.3
            elif "fuse_layers" in name:
                gain = 0.45
            sd[name] = torch.randn(shape, generator=g) * (gain * (1.6 / fan_in) ** 0.5)
    return sd


class _FoldedConv:
    """conv -> eval BatchNorm as one convolution with bias (fold in fp64, kept in fp32)."""

    def __init__(self, sd, conv, bn, stride, device):
        w = sd[f"{conv}.weight"].to(dtype=torch.float64)
        inv = sd[f"{bn}.weight"].to(torch.float64) / torch.sqrt(sd[f"{bn}.running_var"].to(torch.float64) + BN_EPS)
        self.weight = (w * inv.view(-1, 1, 1, 1)).float().to(device).contiguous()
        self.bias = (sd[f"{bn}.bias"].to(torch.float64) - sd[f"{bn}.running_mean"].to(torch.float64) * inv).float().to(device)
        self.stride = stride
        self.pad = w.shape[-1] // 2

    def __call__(self, x, relu=False):
        y = F.conv2d(x, self.weight, self.bias, stride=self.stride, padding=self.pad)
        return F.relu_(y) if relu else y


@BACKBONE.register_module()
class HRNet:
    """``HRNet(cfg)`` as the reference registers it (hrnet.py:444-454); weights arrive through ``load_state_dict``."""

    def __init__(self, cfg=None, state_dict=None, device="cpu"):
        self.name = type(self).__name__
        self.device = torch.device(device)
        self._sd_keys = list(hrnet_param_shapes())
        self._built = False
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # -- weights ------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, prefix="", strict=False):
        missing = [k for k in self._sd_keys if prefix + k not in state_dict]
        if missing:
            raise KeyError(f"HRNet checkpoint lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        sd = {k: state_dict[prefix + k].detach().cpu() for k in self._sd_keys}
        for k, shape in hrnet_param_shapes().items():
            if tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)}, expected {shape}")
        self._fold(sd)
        return [k for k in state_dict if k.startswith(prefix) and k[len(prefix):] not in sd]    # ignored (dead head)

    def to(self, device):
        self.device = torch.device(device)
        if self._built:
            for c in self._convs.values():
                c.weight, c.bias = c.weight.to(self.device), c.bias.to(self.device)
        return self

    def eval(self):
        return self

    def _fold(self, sd):
        self._convs = {conv: _FoldedConv(sd, conv, bn, stride, self.device) for conv, bn, _, _, _, stride in _conv_specs()}
        self._built = True

    # -- forward ------------------------------------------------------------------------------------------------
    def _basic(self, p, x):
        c = self._convs
        out = c[f"{p}.conv1"](x, relu=True)
        out = c[f"{p}.conv2"](out)
        return F.relu_(out.add_(x))

    def _bottleneck(self, p, x):
        c = self._convs
        out = c[f"{p}.conv1"](x, relu=True)
        out = c[f"{p}.conv2"](out, relu=True)
        out = c[f"{p}.conv3"](out)
        res = c[f"{p}.downsample.0"](x) if f"{p}.downsample.0" in c else x
        return F.relu_(out.add_(res))

    def _module(self, p, xs):
        c, nb = self._convs, len(xs)
        xs = list(xs)
        for b in range(nb):
            for k in range(BLOCKS_PER_BRANCH):
                xs[b] = self._basic(f"{p}.branches.{b}.{k}", xs[b])
        fused = []
        for i in range(nb):                                              # hrnet.py:226-233: y = ((t0 + t1) + t2) + t3
            y = None
            for j in range(nb):
                if j == i:
                    t = xs[j]
                elif j > i:
                    t = F.interpolate(c[f"{p}.fuse_layers.{i}.{j}.0"](xs[j]), scale_factor=2 ** (j - i), mode="nearest")
                else:
                    t = xs[j]
                    for k in range(i - j):
                        t = c[f"{p}.fuse_layers.{i}.{j}.{k}.0"](t, relu=k != i - j - 1)
                y = t if y is None else y + t
            fused.append(F.relu(y))
        return fused

    def _transition(self, s, ys, nb):
        c, out = self._convs, []
        for i in range(nb):
            if i < len(ys):
                name = f"transition{s}.{i}.0"
                out.append(c[name](ys[-1], relu=True) if name in c else ys[i])      # hrnet.py:397-410: ys[-1], as upstream
            else:
                t = ys[-1]
                for j in range(i + 1 - len(ys)):
                    t = c[f"transition{s}.{i}.{j}.0"](t, relu=True)
                out.append(t)
        return out

    @torch.no_grad()
    def forward(self, x):
        """x (BN,3,H,W) -> [(BN,40,H/4,W/4), (BN,80,H/8,W/8), (BN,160,H/16,W/16), (BN,320,H/32,W/32)]"""
        if not self._built:
            raise RuntimeError("HRNet has no weights: call load_state_dict first")
        c = self._convs
        x = x.to(device=self.device, dtype=torch.float32)
        x = c["conv1"](x, relu=True)
        x = c["conv2"](x, relu=True)
        for i in range(4):
            x = self._bottleneck(f"layer1.{i}", x)
        ys = [x]
        for s, (nb, nm) in enumerate(STAGES, start=1):
            ys = self._transition(s, ys, nb)
            for m in range(nm):
                ys = self._module(f"stage{s + 1}.{m}", ys)
        return ys

    __call__ = forward
