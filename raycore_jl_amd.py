"""Import alias: the package directory is named `raycore.jl_amd/` (a dot cannot appear in a Python
package name), so `import raycore_jl_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raycore.jl_amd")
_spec = importlib.util.spec_from_file_location("raycore_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["raycore_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
