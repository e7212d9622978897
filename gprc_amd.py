"""Import alias: `import gprc_amd` loads the package directory `gaussian-process-regression_amd/`
(its mandated name contains hyphens, which Python cannot import directly)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gaussian-process-regression_amd")
_spec = importlib.util.spec_from_file_location("gprc_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gprc_amd"] = _mod
_spec.loader.exec_module(_mod)
