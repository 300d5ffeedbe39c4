"""Import alias: the package directory is ``flux3d.jl_amd`` (a dot is not importable), so
``import flux3d_jl_amd`` loads that directory as the package ``flux3d_jl_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flux3d.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "flux3d_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["flux3d_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
