"""Import helper: the product package directory is `exaadmm.jl_amd/` (the name the build
contract fixes); a dot cannot appear in a Python module name, so load it by path."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = "exaadmm_jl_amd"


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(_ROOT, "exaadmm.jl_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


def load_submodule(name):
    load_package()
    return importlib.import_module(_NAME + "." + name)
