"""Import shim: exposes the package directory `faldoi-ipol_amd/` (not a valid Python identifier)
under the importable name `faldoi_ipol_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "faldoi-ipol_amd")
_spec = importlib.util.spec_from_file_location("faldoi_ipol_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["faldoi_ipol_amd"] = _mod
_spec.loader.exec_module(_mod)
