"""INTEGRATION.md section 1, executed: the package's classes registered into the REFERENCE's own ``HEAD`` / ``TRANSFORMER``
registries replace the hot path for the reference's own ``build_head`` (lib/models/POEM.py:114-115, lib/utils/builder.py:9-47,
252-304).  Build container only (needs /root/reference; skipped elsewhere).  Runs in a subprocess: the harness mocks
absent third-party packages in ``sys.modules`` and changes the working directory."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SCRIPT = r'''
import os, sys
root = sys.argv[1]
sys.path[:0] = [root, os.path.join(root, "oracle"), os.path.join(root, "tests", "golden")]
sys.dont_write_bytecode = True
import ref_harness as rh
CN, ref_build_head = rh.setup()                                  # imports lib.models: the reference registers its own classes
import yaml
from lib.utils.builder import HEAD, TRANSFORMER
import lib.models.heads.ptEmb_head as ref_head_mod
with open(os.path.join(rh.REF_ROOT, "config/release/train_medium.yaml")) as f:
    y = yaml.safe_load(f)
cfg = CN(y["MODEL"]["HEAD"])
preset = CN(y["DATA_PRESET"])

# 1. the reference's own head, for its checkpoint surface (every live AND dead tensor: SURVEY a21)
ref = ref_build_head(cfg, data_preset=preset)
assert type(ref) is ref_head_mod.POEM_Generalized_Head
ref_sd = ref.state_dict()

# 2. the block INTEGRATION.md section 1 adds to lib/models/__init__.py, verbatim
import poem_v2_amd as _pk
HEAD.register_module(name="POEM_Generalized_Head", force=True, module=_pk.POEM_Generalized_Head)
TRANSFORMER.register_module(name="PtEmbedTRv4", force=True, module=_pk.PtEmbedTRv4)

# 3. the reference's build_head now returns this package's head, constructed from the reference's own yacs node + kwargs
head = ref_build_head(cfg, data_preset=preset)
assert type(head) is _pk.POEM_Generalized_Head, type(head)
assert type(head.transformer) is _pk.PtEmbedTRv4
assert head.num_preds == ref.num_preds == 3 and head.embed_dims == 256 and head.nsample == ref.nsample
assert head.cfg_transformer.N_BLOCKS == 3

# 4. a reference checkpoint loads: the full state_dict (dead tensors included), bare and under the full-model prefix
ignored = head.load_reference_state_dict(ref_sd)
live = set(head.state_dict())
assert live <= set(ref_sd), sorted(live - set(ref_sd))[:5]
assert len(live) == 199 and len(ignored) == len(ref_sd) - 199, (len(live), len(ignored), len(ref_sd))
assert any(k.startswith("center_shift_layer") or "word_embeddings" in k for k in ignored)
import torch
for k in ("input_proj.weight", "transformer.pt_metro_encoder.2.encoder.vec_attn.query_cross_attn.fc_gamma.2.weight",
          "merge_net_feature.1.2.bias", "query_feat_embedding.weight"):
    assert torch.equal(head.state_dict()[k], ref_sd[k]), k
ignored2 = head.load_reference_state_dict({"ptEmb_head." + k: v for k, v in ref_sd.items()})
assert ignored2 == ignored
print("DROPIN_OK", len(ref_sd), len(ignored))
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="needs the reference tree (build container only)")
def test_reference_registry_builds_this_head_and_loads_its_checkpoint(tmp_path):
    script = tmp_path / "dropin.py"
    script.write_text(_SCRIPT)
    out = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "DROPIN_OK" in out.stdout
