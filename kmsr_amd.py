"""Import shim: exposes the package directory `kernel-modeling-super-resolution_amd/` as `kmsr_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel-modeling-super-resolution_amd")
_spec = importlib.util.spec_from_file_location("kmsr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kmsr_amd"] = _mod
_spec.loader.exec_module(_mod)
