"""``fwht_cuda`` for a reference checkout that keeps its own ``src`` package: put THIS directory on ``sys.path``
(INTEGRATION.md 1).  Same function as the module of the same name at the repo root -- ``fwht(X)``, the contract of
src/fwht/cuda/fwht_cuda.cpp:5-18 -- loaded by file location so that the repo root (and with it the ``src`` alias
package, which would shadow the reference's ``src``) stays off the path."""
from _whvi_loader import implementation as _implementation

fwht = _implementation("fwht_cuda").fwht
__all__ = ["fwht"]
