"""ctypes binding of libpoem_hip.so (include/poem_hip.h).  PyTorch is used only as the owner of device memory and
streams: every call passes raw ``data_ptr()`` values and the current HIP stream.  There is NO fallback: if the shared
library is missing or a call fails, a RuntimeError is raised."""
import ctypes
import os
import subprocess

import numpy as np
import torch  # noqa: F401  (must be imported before the library so that both share one libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("POEM_HIP_LIB") or os.path.join(CSRC, "libpoem_hip.so")   # override: A/B runs of a kept build
ASSETS = os.path.join(_HERE, "assets")

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
PRECISIONS = {"fp32": 0, "split_f16x3": 1, "split_f16x3_all": 2}            # include/poem_hip.h POEM_PRECISION_*


class PoemConfig(ctypes.Structure):
    _fields_ = [("embed", ctypes.c_int32), ("in_channels", ctypes.c_int32), ("nsample", ctypes.c_int32),
                ("nquery", ctypes.c_int32), ("heads", ctypes.c_int32), ("nblocks", ctypes.c_int32),
                ("knn", ctypes.c_int32), ("parametric", ctypes.c_int32), ("feat_h", ctypes.c_int32),
                ("feat_w", ctypes.c_int32), ("max_views", ctypes.c_int32), ("radius", ctypes.c_float),
                ("ln_eps", ctypes.c_float),
                # ABI 2 (round 5): the positional-encoding switches of the reference's constructor (include/poem_hip.h)
                ("pe_normalize", ctypes.c_int32), ("petr_embedding", ctypes.c_int32), ("depth_num", ctypes.c_int32),
                ("lid", ctypes.c_int32), ("reserved0", ctypes.c_int32), ("depth_start", ctypes.c_double),
                ("depth_end", ctypes.c_double), ("position_range", ctypes.c_double * 6)]


_vp, _i, _f, _sz, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64
_cfgp = ctypes.POINTER(PoemConfig)

# name -> (restype, argtypes); mirrors include/poem_hip.h one to one (tests check the symbol list against the header)
SIGNATURES = {
    "poem_abi_version": (_i, []),
    "poem_last_hip_error": (_i, []),
    "poem_error_string": (ctypes.c_char_p, [_i]),
    "poem_num_weight_tensors": (_i, [_cfgp]),
    "poem_weight_tensor_numel": (_i64, [_cfgp, _i]),
    "poem_packed_bytes": (_sz, [_cfgp]),
    "poem_create": (_i, [_cfgp, ctypes.POINTER(_vp), _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, ctypes.POINTER(_vp)]),
    "poem_destroy": (None, [_vp]),
    "poem_workspace_bytes": (_sz, [_vp, _i, _i]),
    "poem_graph_stats": (_i, [_vp, ctypes.POINTER(ctypes.c_int64), _i]),
    "poem_head_forward": (_i, [_vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int32), _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                               _sz, _vp]),
    "poem_decoder_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "poem_finalize_parametric": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "poem_tap": (_i64, [_vp, ctypes.c_char_p, _vp, _i64, _vp]),
    "poem_enable_taps": (_i, [_vp, _i]),
    "poem_profile_enable": (_i, [_vp, _i]),
    "poem_profile_read": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_f), _i]),
    "poem_profile_read_anchored": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_f)]),
    "poem_packed_linear_bytes": (_sz, [_i, _i]),
    "poem_pack_linear": (_i, [_vp, _i, _i, _vp, _vp]),
    "poem_set_overlap": (_i, [_vp, _i]),
    "poem_set_anchor_tables": (_i, [_vp, _i]),
    "poem_set_chains": (_i, [_vp, _i]),
    "poem_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "poem_set_precision": (_i, [_vp, _i]),
    "poem_pack_split_linear": (_i, [_vp, _i, _vp, _vp, _vp]),
    "poem_pack_split_gemm": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "poem_gemm_split": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "poem_vector_attention_split": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _i, _i, _i, _vp]),
    "poem_gemm": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "poem_gemm_ex": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "poem_pack_rows": (_i, [_vp, _i, _i, _vp, _vp]),
    "poem_unpack_rows": (_i, [_vp, _i, _i, _vp, _vp]),
    "poem_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "poem_pe_table": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "poem_pe_table_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "poem_frustum_features": (_i, [_cfgp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "poem_input_proj": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "poem_project_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "poem_merge_reduce": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "poem_merge_finalize": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "poem_cross_attention_scratch_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "poem_cross_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "poem_cross_attention_merged": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "poem_cross_attention_split_f16x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "poem_triangulate_dlt": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "poem_heatmap_uv": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "poem_conv3x3_packed_bytes": (_sz, [_i, _i]),
    "poem_pack_conv3x3": (_i, [_vp, _i, _i, _vp, _vp]),
    "poem_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _i, _i, _i, _vp]),
    "poem_set_decode_option": (_i, [ctypes.c_char_p, _i]),
    "poem_conv3x3_down2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i64, _i, _i, _i, _vp]),
    "poem_upsample2_concat_pad": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "poem_conv1x1_upsample2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "poem_upcat_conv3x3": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _i, _i, _i, _vp]),
    "poem_pool_conv1x1_sigmoid": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "poem_upcat_conv3x3_pool_head": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "poem_pa_epe": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "poem_mano_to_openpose": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "poem_rot6d_to_axis_angle": (_i, [_vp, _vp, _vp, _i, _vp]),
    "poem_mano_table_bytes": (_sz, []),
    "poem_mano_prepare": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "poem_mano_lbs": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "poem_attach_mano": (_i, [_vp, _vp, _i]),
    "poem_profile_read_stage": (_i, [_vp, _i, ctypes.POINTER(_i), ctypes.POINTER(_f)]),
    "poem_warp_affine": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "poem_pck_accumulate": (_i, [_vp, _vp, _i, _i, ctypes.c_double, ctypes.c_double, _i, _vp, _vp, _vp, _vp, _vp]),
    "poem_knn": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "poem_knn_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "poem_vector_attention": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _i, _i, _i, _vp]),
    "poem_reg_update": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
}

_LIB = None


def build(verbose=False):
    """Compile libpoem_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode != 0:
        raise RuntimeError("building libpoem_hip.so failed")
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU/PyTorch fallback for this path)")
        L = ctypes.CDLL(LIB_PATH)
        # the ABI first: a stale .so would read PoemConfig with another layout -- and would lack symbols, so the "rebuild it"
        # message has to come before the symbol loop's AttributeError
        try:
            L.poem_abi_version.restype = _i
            abi = L.poem_abi_version()
        except AttributeError:
            abi = None
        if abi != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI {abi}, this package binds ABI {ABI_VERSION}: rebuild it "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


POEM_E_UNSUPPORTED = -4      # include/poem_hip.h
ABI_VERSION = 3              # poem_abi_version(): 2 = poem_config_t with the positional-encoding switches (round 5);
                             # 3 = poem_mano_lbs takes the prepared asset table, poem_attach_mano (round 6)


def check(rc, what=""):
    if rc is not None and rc < 0:
        L = lib()
        msg = L.poem_error_string(int(rc)).decode()
        raise RuntimeError(f"libpoem_hip {what} failed: {msg} (code {rc}, hipError {L.poem_last_hip_error()})")
    return rc


def ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libpoem_hip operates on device tensors only (no CPU path)")
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"expected contiguous {dtype} tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def make_config(embed, in_channels=160, nsample=4096, nquery=799, heads=4, nblocks=3, knn=32, parametric=False,
                feat_h=16, feat_w=16, max_views=10, radius=0.1, ln_eps=1e-12, pe_normalize=True, petr_embedding=False,
                depth_num=32, lid=False, depth_start=0.0, depth_end=1.2, position_range=(-0.6, -0.6, 0.0, 0.6, 0.6, 1.2)):
    return PoemConfig(embed, in_channels, nsample, nquery, heads, nblocks, knn, int(bool(parametric)), feat_h, feat_w,
                      max_views, radius, ln_eps, int(bool(pe_normalize)), int(bool(petr_embedding)), int(depth_num),
                      int(bool(lid)), 0, float(depth_start), float(depth_end), (ctypes.c_double * 6)(*[float(v) for v in position_range]))


def load_assets(nsample, root=None):
    """bps (S,3), anchor (32,3), anchor_idx (32,) -- from ``<root>/assets`` when given/present (the reference reads
    them relative to cwd: ptEmb_head.py:791, point_transformers.py:12-13), else the copies shipped in the package."""
    for d in ([os.path.join(root, "assets")] if root else []) + [os.path.join(os.getcwd(), "assets"), ASSETS]:
        if all(os.path.exists(os.path.join(d, f)) for f in ("bps.npy", "anchor.npy", "anchor_idx.npy")):
            bps = np.load(os.path.join(d, "bps.npy")).reshape(-1, 3)
            if bps.shape[0] < nsample:
                continue
            anchor = np.load(os.path.join(d, "anchor.npy")).reshape(-1, 3)
            aidx = np.load(os.path.join(d, "anchor_idx.npy")).reshape(-1)
            return (torch.from_numpy(bps[:nsample].astype(np.float32).copy()),
                    torch.from_numpy(anchor.astype(np.float32).copy()), torch.from_numpy(aidx.astype(np.int64).copy()))
    raise FileNotFoundError("bps.npy / anchor.npy / anchor_idx.npy not found")


class Engine:
    """Owns one ``poem_handle_t``: raw weights, the packed image, constant tables and a grow-only workspace."""

    def __init__(self, cfg: PoemConfig, weights, bps, anchor, anchor_idx, template, device):
        from .weights import live_key_shapes
        L = lib()
        self.cfg = cfg
        self.device = torch.device(device)
        shapes = live_key_shapes(cfg.embed, cfg.in_channels, cfg.nquery, cfg.nblocks, bool(cfg.parametric),
                                 petr=bool(cfg.petr_embedding), depth_num=cfg.depth_num)
        n = L.poem_num_weight_tensors(ctypes.byref(cfg))
        check(n, "poem_num_weight_tensors")
        if n != len(shapes):
            raise RuntimeError(f"tensor table mismatch: library {n} vs python {len(shapes)}")
        self.raw = []
        for i, (key, shape) in enumerate(shapes.items()):
            t = weights[key].detach().to(self.device, torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape) or t.numel() != L.poem_weight_tensor_numel(ctypes.byref(cfg), i):
                raise RuntimeError(f"{key}: shape {tuple(t.shape)} does not match the library's tensor table")
            self.raw.append(t)
        self.bps = bps.to(self.device, torch.float32).contiguous()
        self.anchor = anchor.to(self.device, torch.float32).contiguous()
        self.anchor_idx = anchor_idx.to(self.device, torch.int32).contiguous()
        self.template = template.to(self.device, torch.float32).contiguous()
        assert self.bps.shape == (cfg.nsample, 3) and self.anchor.shape == (32, 3) and self.template.shape == (cfg.nquery, 3)
        # quirk Q2: the anchor ids index the query rows (vector self attention) AND the basis-point rows (cross attention)
        if anchor_idx.numel() != 32 or int(anchor_idx.min()) < 0 or int(anchor_idx.max()) >= min(cfg.nquery, cfg.nsample):
            raise RuntimeError(f"anchor_idx must hold 32 ids in [0, {min(cfg.nquery, cfg.nsample)}) "
                               f"(got {anchor_idx.numel()} ids, range [{int(anchor_idx.min())}, {int(anchor_idx.max())}])")
        nbytes = L.poem_packed_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            raise RuntimeError("unsupported configuration for libpoem_hip")
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        arr = (_vp * n)(*[t.data_ptr() for t in self.raw])
        h = _vp()
        with torch.cuda.device(self.device):
            check(L.poem_create(ctypes.byref(cfg), arr, n, self.bps.data_ptr(), self.anchor.data_ptr(),
                                self.anchor_idx.data_ptr(), self.template.data_ptr(), self.packed.data_ptr(), nbytes,
                                stream(), ctypes.byref(h)), "poem_create")
        self.handle = h
        self.workspace = None
        self._taps = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib().poem_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _ws(self, batch, views):
        # head path: sized for the batch size's worst case (batch * max_views): the library then lays the workspace out for
        # that capacity and one launch graph per batch size serves every view layout (include/poem_hip.h poem_workspace_bytes)
        need = lib().poem_workspace_bytes(self.handle, batch, views)
        if need == 0:
            raise RuntimeError("poem_workspace_bytes failed")
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.workspace, self.workspace.numel()

    GRAPH_STAT_NAMES = ("cached_execs", "captures", "instantiations", "replays", "plain_forwards", "layout_uploads",
                        "parked_execs", "exec_reuses", "exec_update_refusals", "exec_busy_skips")

    def graph_stats(self):
        """Counters of the launch-graph cache (include/poem_hip.h poem_graph_stats)."""
        out = (ctypes.c_int64 * 10)()
        check(lib().poem_graph_stats(self.handle, out, 10), "poem_graph_stats")
        return dict(zip(self.GRAPH_STAT_NAMES, [int(v) for v in out]))

    def enable_taps(self, flag=True):
        self._taps = bool(flag)
        check(lib().poem_enable_taps(self.handle, int(flag)))

    def set_overlap(self, flag=True):
        check(lib().poem_set_overlap(self.handle, int(flag)), "poem_set_overlap")

    def set_anchor_tables(self, flag=True):
        """Block-0 positional products once per forward instead of per sample (include/poem_hip.h)."""
        check(lib().poem_set_anchor_tables(self.handle, int(flag)), "poem_set_anchor_tables")

    def set_option(self, name, value):
        check(lib().poem_set_option(self.handle, name.encode(), int(value)), f"poem_set_option({name})")

    def set_chains(self, flag=True):
        """Query-side row-tile chain kernels (default on) vs one launch per operator (include/poem_hip.h)."""
        check(lib().poem_set_chains(self.handle, int(flag)), "poem_set_chains")

    def set_precision(self, mode):
        check(lib().poem_set_precision(self.handle, PRECISIONS[mode] if isinstance(mode, str) else int(mode)),
              "poem_set_precision")

    def profile_enable(self, max_launches):
        check(lib().poem_profile_enable(self.handle, int(max_launches)), "poem_profile_enable")

    def profile_read(self, reset=True):
        n, ms = _i(0), _f(0.0)
        check(lib().poem_profile_read(self.handle, ctypes.byref(n), ctypes.byref(ms), int(reset)), "poem_profile_read")
        return n.value, ms.value

    def profile_read_stage(self, stage):
        """(launches, total ms) of one timed span kind: 0 full vector attention, 1 anchored block-0 form, 2 sampling front
        end (input_proj .. merge finalize)  (include/poem_hip.h POEM_PROF_*; other values read (0, 0))."""
        n, ms = _i(0), _f(0.0)
        check(lib().poem_profile_read_stage(self.handle, int(stage), ctypes.byref(n), ctypes.byref(ms)), "poem_profile_read_stage")
        return n.value, ms.value

    def profile_read_anchored(self):
        n, ms = _i(0), _f(0.0)
        check(lib().poem_profile_read_anchored(self.handle, ctypes.byref(n), ctypes.byref(ms)), "poem_profile_read_anchored")
        return n.value, ms.value

    def head_forward(self, mlvl_feat, cam_intr, cam_extr, cam_view_num, reference_joints, inp_img_shape):
        c = self.cfg
        views = [int(v) for v in cam_view_num]
        B, BN = len(views), int(sum(views))
        if tuple(mlvl_feat.shape) != (BN, c.in_channels, c.feat_h, c.feat_w):
            raise RuntimeError(f"mlvl_feat shape {tuple(mlvl_feat.shape)} != {(BN, c.in_channels, c.feat_h, c.feat_w)}")
        # the C ABI takes raw pointers: a wrong-shaped camera / joint tensor would be an out-of-bounds device read where the
        # reference raises a shape error
        for what, t, shape in (("cam_intr", cam_intr, (BN, 3, 3)), ("cam_extr", cam_extr, (BN, 4, 4)),
                               ("reference_joints", reference_joints, (B, 21, 3))):
            if tuple(t.shape) != shape:
                raise RuntimeError(f"{what} shape {tuple(t.shape)} != {shape} (cam_view_num sums to {BN} views, {B} samples)")
        offs = (ctypes.c_int32 * (B + 1))(*np.concatenate([[0], np.cumsum(views)]).astype(np.int32).tolist())
        ws, need = self._ws(B, max(BN, B * c.max_views))
        out = torch.empty(c.nblocks, B, c.nquery, 3, dtype=torch.float32, device=self.device)
        pose = torch.empty(B, 48, dtype=torch.float32, device=self.device) if c.parametric else None
        betas = torch.empty(B, 10, dtype=torch.float32, device=self.device) if c.parametric else None
        with torch.cuda.device(self.device):
            check(lib().poem_head_forward(self.handle, ptr(mlvl_feat), ptr(cam_intr), ptr(cam_extr), offs, B,
                                          ptr(reference_joints), int(inp_img_shape[0]), int(inp_img_shape[1]),
                                          ptr(out), ptr(pose), ptr(betas), ws.data_ptr(), need, stream()),
                  "poem_head_forward")
        return out, pose, betas

    def decoder_forward(self, query_xyz, query_feat, pt_xyz, pt_feats):
        c = self.cfg
        B = query_xyz.shape[0]
        ws, need = self._ws(B, B)
        out = torch.empty(c.nblocks, B, c.nquery, 3, dtype=torch.float32, device=self.device)
        pose = torch.empty(B, 48, dtype=torch.float32, device=self.device) if c.parametric else None
        betas = torch.empty(B, 10, dtype=torch.float32, device=self.device) if c.parametric else None
        with torch.cuda.device(self.device):
            check(lib().poem_decoder_forward(self.handle, ptr(query_xyz), ptr(query_feat), ptr(pt_xyz), ptr(pt_feats), B,
                                             ptr(out), ptr(pose), ptr(betas), ws.data_ptr(), need, stream()),
                  "poem_decoder_forward")
        return out, pose, betas

    def attach_mano(self, table, center_idx):
        """The MANO layer inside the forward (include/poem_hip.h poem_attach_mano); ``table`` = ManoLayer.th_table or None."""
        check(lib().poem_attach_mano(self.handle, None if table is None else table.data_ptr(), int(center_idx)), "poem_attach_mano")
        self._mano = table      # (keeps the caller's table alive while attached)

    def finalize_parametric(self, verts, joints, reference_joints, out):
        with torch.cuda.device(self.device):
            check(lib().poem_finalize_parametric(self.handle, ptr(verts), ptr(joints), ptr(reference_joints),
                                                 out.shape[1], ptr(out), stream()), "poem_finalize_parametric")
        return out

    def tap(self, name, shape, dtype=torch.float32):
        n = lib().poem_tap(self.handle, name.encode(), None, 0, None)
        check(n, f"poem_tap({name})")
        t = torch.empty(int(n), dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().poem_tap(self.handle, name.encode(), t.data_ptr(), int(n), stream()))
        return t.view(*shape)


# ---- thin operator wrappers (used by the operator-level parity tests) -------------------------------------------
def pack_linear(w):
    n, k = w.shape
    nbytes = lib().poem_packed_linear_bytes(n, k)
    if nbytes == 0:
        raise RuntimeError("in_features must be a multiple of 8")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(lib().poem_pack_linear(ptr(w), n, k, out.data_ptr(), stream()), "poem_pack_linear")
    return out


def gemm(x, w_packed, n_out, bias=None, residual=None, act=ACT_NONE):
    M, K = x.shape
    y = torch.empty(M, n_out, dtype=torch.float32, device=x.device)
    check(lib().poem_gemm(ptr(x), K, w_packed.data_ptr(), ptr(bias), ptr(residual), n_out, ptr(y), n_out, M, n_out, K,
                          act, stream()), "poem_gemm")
    return y


def pack_rows(x):
    """(rows, cols) row-major -> packed-activation image (uint8 buffer)."""
    return pack_linear(x)


def unpack_rows(pa, rows, cols):
    x = torch.empty(rows, cols, dtype=torch.float32, device=pa.device)
    check(lib().poem_unpack_rows(pa.data_ptr(), rows, cols, ptr(x), stream()), "poem_unpack_rows")
    return x


def gemm_ex(x, w_packed, M, n_out, K, bias=None, residual=None, act=ACT_NONE, in_pa=False, out_pa=False):
    """x / residual / result are row-major tensors or PA byte buffers according to the layout flags."""
    if out_pa:
        y = torch.empty(lib().poem_packed_linear_bytes(M, n_out), dtype=torch.uint8, device=x.device)
    else:
        y = torch.empty(M, n_out, dtype=torch.float32, device=x.device)
    rptr = None if residual is None else residual.data_ptr()
    check(lib().poem_gemm_ex(x.data_ptr(), K, w_packed.data_ptr(), ptr(bias), rptr, n_out, y.data_ptr(), n_out, M, n_out,
                             K, act, int(in_pa), int(out_pa), stream()), "poem_gemm_ex")
    return y


def layernorm(x, g, b, eps):
    y = torch.empty_like(x)
    check(lib().poem_layernorm(ptr(x), ptr(g), ptr(b), ptr(y), x.shape[0], x.shape[1], eps, stream()), "poem_layernorm")
    return y


def cross_attention(q, k, v, heads, split=False, merged=False):
    B, NQ, C = q.shape
    ctx = torch.empty_like(q)
    need = lib().poem_cross_attention_scratch_bytes(B, NQ, k.shape[1], C, heads)
    scratch = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    fn = lib().poem_cross_attention_split_f16x3 if split else (lib().poem_cross_attention_merged if merged else lib().poem_cross_attention)
    check(fn(ptr(q), ptr(k), ptr(v), ptr(ctx), B, NQ, k.shape[1], C, heads, scratch.data_ptr(), need, stream()),
          "poem_cross_attention")
    return ctx


def knn(query_xyz, src_xyz, fma=False):
    """fma=True: distances with the fma contraction of pytorch3d's CUDA kernel (include/poem_hip.h poem_knn_ex)."""
    B, NQ, _ = query_xyz.shape
    idx = torch.empty(B, NQ, 32, dtype=torch.int32, device=query_xyz.device)
    if fma:
        check(lib().poem_knn_ex(ptr(query_xyz), ptr(src_xyz), idx.data_ptr(), B, NQ, src_xyz.shape[1], 1, stream()), "poem_knn_ex")
    else:
        check(lib().poem_knn(ptr(query_xyz), ptr(src_xyz), idx.data_ptr(), B, NQ, src_xyz.shape[1], stream()), "poem_knn")
    return idx


def vector_attention(query_xyz, src_xyz, anchor_xyz, idx, q, k, v, wd1, bd1, wd2p, bd2, wg1p, bg1, wg2p, bg2):
    B, NQ, C = q.shape
    out = torch.empty_like(q)
    shared = 1 if idx.dim() == 1 else 0
    check(lib().poem_vector_attention(ptr(query_xyz), ptr(src_xyz), ptr(anchor_xyz), idx.data_ptr(), shared, ptr(q), ptr(k),
                                      ptr(v), k.shape[1], ptr(wd1), ptr(bd1), wd2p.data_ptr(), ptr(bd2), wg1p.data_ptr(),
                                      ptr(bg1), wg2p.data_ptr(), ptr(bg2), ptr(out), B, NQ, C, stream()),
          "poem_vector_attention")
    return out



def gemm_split(x, w, bias=None, residual=None, act=ACT_NONE):
    """y = act(x w^T + bias) + residual through the split-precision panel GEMM (w: (N,K) row-major fp32 device tensor)."""
    N, K = w.shape
    M = x.shape[0]
    nt = (N + 31) // 32
    img = torch.empty(nt * 32 * K * 4, dtype=torch.uint8, device=w.device)
    sc = torch.empty(nt, dtype=torch.float32, device=w.device)
    check(lib().poem_pack_split_gemm(ptr(w), N, K, img.data_ptr(), sc.data_ptr(), stream()), "poem_pack_split_gemm")
    y = torch.empty(M, N, dtype=torch.float32, device=w.device)
    check(lib().poem_gemm_split(ptr(x), x.stride(0), img.data_ptr(), sc.data_ptr(), ptr(bias), ptr(residual),
                                residual.stride(0) if residual is not None else 0, ptr(y), N, M, N, K, act, stream()),
          "poem_gemm_split")
    return y


def pack_split_linear(w):
    """(C,C) fp32 device tensor -> (image uint8 (C*C*4,), scale float32 (1,)) for the split-precision vector attention."""
    C = w.shape[0]
    img = torch.empty(C * C * 4, dtype=torch.uint8, device=w.device)
    sc = torch.empty(1, dtype=torch.float32, device=w.device)
    check(lib().poem_pack_split_linear(ptr(w), C, img.data_ptr(), sc.data_ptr(), stream()), "poem_pack_split_linear")
    return img, sc


def vector_attention_split(query_xyz, src_xyz, anchor_xyz, idx, qg, kg, v, wd1, bd1, wd2_img, bd2, wg1d2_img, wg2_img, scales):
    """Composed form on the f16 matrix cores (include/poem_hip.h poem_vector_attention_split)."""
    B, NQ, C = qg.shape
    out = torch.empty_like(qg)
    shared = 1 if idx.dim() == 1 else 0
    check(lib().poem_vector_attention_split(ptr(query_xyz), ptr(src_xyz), ptr(anchor_xyz), idx.data_ptr(), shared, ptr(qg),
                                            ptr(kg), ptr(v), kg.shape[1], ptr(wd1), ptr(bd1), wd2_img.data_ptr(), ptr(bd2),
                                            wg1d2_img.data_ptr(), wg2_img.data_ptr(), ptr(scales), ptr(out), B, NQ, C,
                                            stream()), "poem_vector_attention_split")
    return out

def rot6d_to_axis_angle(params):
    """params (B,106) device fp32 -> (pose_aa (B,48), betas (B,10))."""
    B = params.shape[0]
    pose = torch.empty(B, 48, dtype=torch.float32, device=params.device)
    betas = torch.empty(B, 10, dtype=torch.float32, device=params.device)
    check(lib().poem_rot6d_to_axis_angle(ptr(params), ptr(pose), ptr(betas), B, stream()), "poem_rot6d_to_axis_angle")
    return pose, betas


def project_sample(x, bps, centre, view_sample, cam_intr, cam_extr, img_shape):
    BN, C, fh, fw = x.shape
    S = bps.shape[0]
    uv = torch.empty(BN * S * 2 + BN * 16, dtype=torch.float32, device=x.device)
    g = torch.empty(BN, C, S, dtype=torch.float32, device=x.device)
    check(lib().poem_project_sample(ptr(x), ptr(bps), ptr(centre), view_sample.data_ptr(), ptr(cam_intr), ptr(cam_extr),
                                    ptr(uv), ptr(g), BN, C, fh, fw, S, int(img_shape[0]), int(img_shape[1]), stream()),
          "poem_project_sample")
    return g, uv[:BN * S * 2].view(BN, S, 2)
