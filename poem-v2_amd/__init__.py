"""MI355X-native point-embedded decoder for POEM-v2 (hot path only).

Public surface mirrors the reference's plugin API for this path: ``HEAD`` / ``TRANSFORMER`` registries,
``build_head`` / ``build_transformer``, classes ``POEM_Generalized_Head`` and ``PtEmbedTRv4``."""
from .config import CN  # noqa: F401
from .builder import HEAD, TRANSFORMER, Registry, build_from_cfg, build_head, build_transformer  # noqa: F401
from . import weights  # noqa: F401


def __getattr__(name):
    # heavy modules (ctypes binding, nn.Modules) are imported lazily so that ``import poem_v2_amd`` stays cheap
    if name in ("hip", "head", "transformer", "dist", "metrics", "inputs"):
        import importlib
        return importlib.import_module("." + name, __name__)
    if name in ("POEM_Generalized_Head", "PtEmbedTRv4"):
        import importlib
        mod = importlib.import_module(".head" if name == "POEM_Generalized_Head" else ".transformer", __name__)
        return getattr(mod, name)
    raise AttributeError(name)
