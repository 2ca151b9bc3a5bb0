"""Loader for the package directory `xlab-ee-fortran_amd/` (its name is not a Python identifier).

    import xee_pkg; xee = xee_pkg.load()      # -> module `xlab_ee_fortran_amd`
"""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG_DIR = ROOT / "xlab-ee-fortran_amd"
NAME = "xlab_ee_fortran_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, PKG_DIR / "__init__.py",
                                                  submodule_search_locations=[str(PKG_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
