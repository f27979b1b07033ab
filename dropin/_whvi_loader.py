"""Loads this repo's implementation WITHOUT putting the repo root on ``sys.path``.

The repo root carries an import-path alias package ``src/``; a regular package beats the
reference's ``src`` directory (a namespace package: it has no ``__init__.py``) wherever it sits on the path, so a
maintainer who wants the reference's OWN ``src`` logic with only the two native modules swapped must not have the
root on the path.  The modules in this directory therefore import ``whvi_amd`` and the root shims by file location."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _by_path(name, path, search=None):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=search)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[name]
        raise
    return mod


def implementation(shim):
    """The root shim module ``fwht_cuda`` / ``fwht_cpp`` under a private name, with ``whvi_amd`` importable."""
    pkg = os.path.join(ROOT, "whvi_amd")
    _by_path("whvi_amd", os.path.join(pkg, "__init__.py"), [pkg])
    return _by_path("_whvi_dropin_" + shim, os.path.join(ROOT, shim + ".py"))
