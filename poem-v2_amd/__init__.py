"""MI355X-native point-embedded decoder for POEM-v2 (hot path only).

Public surface mirrors the reference's plugin API for this path: ``HEAD`` / ``TRANSFORMER`` registries,
``build_head`` / ``build_transformer``, classes ``POEM_Generalized_Head`` and ``PtEmbedTRv4`` (importing the package
registers them, as ``import lib.models`` does upstream)."""
from .config import CN  # noqa: F401
from .builder import (HEAD, TRANSFORMER, BACKBONE, MODEL, Registry, build_from_cfg, build_head, build_transformer,  # noqa: F401
                      build_backbone, build_model)
from . import builder, weights, inputs, hip, configs, triangulation, decode, backbone, transform, wds, mano  # noqa: F401
from .transformer import PtEmbedTRv4  # noqa: F401
from .head import POEM_Generalized_Head  # noqa: F401
from .backbone import HRNet  # noqa: F401
from .model import PtEmbedMultiviewStereoV2  # noqa: F401
from .transform import TRANSFORM, SimpleTransform3DMultiView, build_transform  # noqa: F401
from .wds import MultiviewWebDataset, MixWebDataset, collation_random_n_views  # noqa: F401
from .mano import ManoLayer  # noqa: F401
