"""Loader for the product package. Its directory is named `u96-slam_amd` (not a Python identifier), so it is
imported under the module name `u96_slam_amd` through importlib."""
import importlib.util
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent
_NAME = "u96_slam_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkgdir = ROOT / "u96-slam_amd"
    spec = importlib.util.spec_from_file_location(_NAME, pkgdir / "__init__.py", submodule_search_locations=[str(pkgdir)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
