"""Plugin boundary of the path: string-keyed registries + ``build_from_cfg``.

Mirrors the reference's discovery contract (lib/utils/builder.py:9-47 ``build_from_cfg``, :252-304
``Registry.register_module``, :307-320 the registry singletons; lib/models/heads/__init__.py:4-5 ``build_head``;
lib/models/bricks/transformer.py:19-21 ``build_transformer``): a class registers under its own name, a config
node names it in ``TYPE``, extra keyword arguments arrive merged into the node as UPPER-CASE keys, and the
class is constructed as ``cls(cfg)``."""
import inspect

from .config import CN


class Registry:
    def __init__(self, name):
        self._name = name
        self._classes = {}

    @property
    def name(self):
        return self._name

    def __len__(self):
        return len(self._classes)

    def __contains__(self, key):
        return key in self._classes

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._classes)})"

    def get(self, key):
        return self._classes.get(key)

    def _add(self, cls, names, force):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        for n in names:
            if n in self._classes and not force:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._classes[n] = cls

    def register_module(self, name=None, force=False, module=None):
        """Decorator ``@REG.register_module()`` / ``@REG.register_module(name="x")`` or call
        ``REG.register_module(module=Cls)``."""
        if not isinstance(force, bool):
            raise TypeError(f"force must be a boolean, but got {type(force)}")
        if inspect.isclass(name):          # bare ``@REG.register_module`` use
            self._add(name, [name.__name__], force)
            return name
        if not (name is None or isinstance(name, str) or
                (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f"name must be None, a str or a sequence of str, but got {type(name)}")

        def names_for(cls):
            if name is None:
                return [cls.__name__]
            return [name] if isinstance(name, str) else list(name)

        if module is not None:
            self._add(module, names_for(module), force)
            return module

        def deco(cls):
            self._add(cls, names_for(cls), force)
            return cls

        return deco


def build_from_cfg(cfg, registry, **kwargs):
    # a node of the reference's own config class (yacs ``CfgNode``, a dict subclass) arrives here when this package's
    # classes are registered in the reference's registries: the head then builds its
    # transformer from ``cfg.TRANSFORMER`` through this function -- adopt the node instead of refusing it
    if isinstance(cfg, dict) and not isinstance(cfg, CN):
        cfg = CN(dict(cfg))
    assert isinstance(cfg, CN) and cfg.get("TYPE") is not None, "cfg must be a config node with a TYPE"
    if kwargs:
        merged = cfg.clone()
        merged.defrost()
        merged.merge_from_other_cfg(CN({k.upper(): v for k, v in kwargs.items()}))
        cfg = merged
    kind = cfg.TYPE
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f"{kind} is not in the {registry.name} registry")
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(kind)}")
    return cls(cfg)


HEAD = Registry("head")
TRANSFORMER = Registry("Transformer")
BACKBONE = Registry("backbone")          # lib/utils/builder.py:307-320 upstream
MODEL = Registry("model")


def build_head(cfg, **kwargs):
    return build_from_cfg(cfg, HEAD, **kwargs)


def build_transformer(cfg, **kwargs):
    return build_from_cfg(cfg, TRANSFORMER, **kwargs)


def build_backbone(cfg, **kwargs):
    return build_from_cfg(cfg, BACKBONE, **kwargs)


def build_model(cfg, **kwargs):
    return build_from_cfg(cfg, MODEL, **kwargs)
