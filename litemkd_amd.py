"""Import shim: `import litemkd_amd` loads the package that lives in ./lite-mkd_amd/
(a hyphen cannot appear in a Python package name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lite-mkd_amd")
_spec = importlib.util.spec_from_file_location(
    "litemkd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["litemkd_amd"] = _mod
_spec.loader.exec_module(_mod)
