"""Import alias: `import scanerf_amd` loads the package whose directory name
(scanerf-..._amd) is not a valid Python identifier."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "scanerf-scalable-bundle-adjusting-neural-radiance-fields-for-large-scale-scene-rendering_amd")
_spec = importlib.util.spec_from_file_location("scanerf_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                               submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["scanerf_amd"] = _mod
_spec.loader.exec_module(_mod)
