"""Import shim: the package directory is ``model-based-rl_amd`` (not a valid identifier);
``import mbrl_amd`` loads it under that name."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model-based-rl_amd")
_spec = importlib.util.spec_from_file_location("mbrl_amd", os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mbrl_amd"] = _mod
_spec.loader.exec_module(_mod)
