"""Import alias for the package directory ``mac-vo_amd/`` (a hyphen is not a legal Python identifier).

``import macvo_amd`` executes this shim, which loads ``mac-vo_amd/__init__.py`` as the package
``macvo_amd`` (with ``mac-vo_amd/`` as its ``__path__``) and replaces itself in ``sys.modules``;
sub-modules then import normally (``import macvo_amd.ops``).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mac-vo_amd")
_spec = importlib.util.spec_from_file_location(
    "macvo_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["macvo_amd"] = _mod
_spec.loader.exec_module(_mod)
