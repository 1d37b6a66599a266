"""Import helper: the package directory is named `monte-carlo-ray-tracer_amd`
(not a valid Python identifier), so it is loaded by path under the module name
`mcrt_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "monte-carlo-ray-tracer_amd")


def load():
    name = "mcrt_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
