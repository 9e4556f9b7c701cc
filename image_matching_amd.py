"""Import alias: the package directory is ``image-matching_amd/`` (hyphenated, per the
project layout), which Python cannot import by name.  ``import image_matching_amd``
finds this file, which loads the real package from that directory and installs it in
``sys.modules`` under the importable name (sub-modules resolve through
``submodule_search_locations``)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image-matching_amd")
_spec = importlib.util.spec_from_file_location(
    "image_matching_amd", os.path.join(_dir, "__init__.py"),
    submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["image_matching_amd"] = _mod
_spec.loader.exec_module(_mod)
