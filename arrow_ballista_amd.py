"""Import shim: the package directory is `arrow-ballista_amd/` (not a valid Python identifier), so
`import arrow_ballista_amd` loads it from there under this importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arrow-ballista_amd")
_spec = importlib.util.spec_from_file_location(
    "arrow_ballista_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["arrow_ballista_amd"] = _mod
_spec.loader.exec_module(_mod)
