"""Import shim: ``import poem_v2_amd`` loads the package that lives in the directory ``poem-v2_amd/``
(the directory name carries a hyphen, so it cannot be imported by name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "poem-v2_amd")
_spec = importlib.util.spec_from_file_location("poem_v2_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["poem_v2_amd"] = _mod
_spec.loader.exec_module(_mod)
