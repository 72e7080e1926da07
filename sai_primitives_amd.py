"""Import shim: the package directory is named `sai-primitives_amd/` (not a Python identifier), so
`import sai_primitives_amd` resolves here and loads that directory as the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sai-primitives_amd")
_spec = importlib.util.spec_from_file_location("sai_primitives_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sai_primitives_amd"] = _mod
_spec.loader.exec_module(_mod)
