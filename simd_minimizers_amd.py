"""Import shim: the package directory is ``simd-minimizers_amd/`` (not a valid Python identifier), so
this module loads that directory AS the package ``simd_minimizers_amd`` through the regular import
machinery (a module spec with the directory as its submodule search path) and puts it into
``sys.modules`` in its own place - no exec of source text, submodules import normally."""
import importlib.util as _util
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "simd-minimizers_amd")
_spec = _util.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = _util.module_from_spec(_spec)
_sys.modules[__name__] = _pkg
_spec.loader.exec_module(_pkg)
