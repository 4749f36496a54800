"""Import alias: ``import mer_amd`` loads the package directory
``multimodal-emotion-recognition_amd/`` (its name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimodal-emotion-recognition_amd")
_spec = importlib.util.spec_from_file_location(
    "mer_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mer_amd"] = _mod
_spec.loader.exec_module(_mod)
