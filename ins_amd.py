"""Import alias: the package directory is named `incompressiblenavierstokes.jl_amd` (not a valid
dotted Python name), so `import ins_amd` loads it under this name, submodules included."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "incompressiblenavierstokes.jl_amd")
_spec = importlib.util.spec_from_file_location("ins_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ins_amd"] = _mod
_spec.loader.exec_module(_mod)
