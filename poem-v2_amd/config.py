"""Config node with the access pattern the reference's heads use (lib/utils/config.py:8-43 wraps
yacs.CfgNode; yacs is not a dependency here): nested dict with attribute access, ``.get``, ``clone``,
``merge_from_other_cfg``, ``freeze``/``defrost``.  Nested plain dicts are converted on construction."""
import copy


class CN(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        object.__setattr__(self, CN._FROZEN, False)
        for k, v in (init_dict or {}).items():
            dict.__setitem__(self, k, self._wrap(v))

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, CN):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(e) for e in v]
        return v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __setitem__(self, key, value):
        if object.__getattribute__(self, CN._FROZEN):
            raise AttributeError(f"attempted to modify frozen CN at key {key}")
        dict.__setitem__(self, key, self._wrap(value))

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CN()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    def _set_frozen(self, flag):
        object.__setattr__(self, CN._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CN):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, CN._FROZEN)

    def set_new_allowed(self, flag):
        pass

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if k in self and isinstance(self[k], CN) and isinstance(v, dict):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = copy.deepcopy(v)

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self.merge_from_other_cfg(CN(yaml.safe_load(f)))

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CN) else v) for k, v in self.items()}

    def dump(self, **kw):
        import yaml
        return yaml.safe_dump(self.to_dict(), **kw)
