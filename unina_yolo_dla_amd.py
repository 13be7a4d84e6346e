"""Import shim: the package directory is named ``unina-yolo-dla_amd`` (a hyphen
is not a valid Python identifier), so ``import unina_yolo_dla_amd`` lands here
and this module replaces itself with the real package loaded from that path."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "unina-yolo-dla_amd")
_spec = importlib.util.spec_from_file_location(
    "unina_yolo_dla_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["unina_yolo_dla_amd"] = _mod
_spec.loader.exec_module(_mod)
