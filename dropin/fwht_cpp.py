"""``fwht_cpp`` for a reference checkout that keeps its own ``src`` package: put THIS directory on ``sys.path``
(INTEGRATION.md 1).  ``forward`` / ``backward`` with the contract of src/fwht/cpp/fwht.cpp:23-34, from the module of
the same name at the repo root, loaded by file location (see ``_whvi_loader``)."""
from _whvi_loader import implementation as _implementation

_impl = _implementation("fwht_cpp")
forward, backward = _impl.forward, _impl.backward
__all__ = ["forward", "backward"]
