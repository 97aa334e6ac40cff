"""Import alias: the product package directory is `gaussian-lic_amd/` (not a valid Python identifier);
`import gaussian_lic_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gaussian-lic_amd")
_spec = importlib.util.spec_from_file_location(
    "gaussian_lic_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gaussian_lic_amd"] = _mod
_spec.loader.exec_module(_mod)
