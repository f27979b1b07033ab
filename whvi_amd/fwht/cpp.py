"""Host front-end on the native OpenMP library; interface of src/fwht/cpp/fwht.py (``FWHTFunction``, ``FWHT``)."""
import fwht_cpp

from whvi_amd.fwht._frontends import make_fwht_function, make_fwht_module

__all__ = ["FWHTFunction", "FWHT"]

FWHTFunction = make_fwht_function(
    fwht_cpp.forward, "FWHTFunction",
    "Batched FWHT along dimension 1 of a host tensor through ``fwht_cpp.forward`` (libwhvi_cpu.so).")
FWHT = make_fwht_module(FWHTFunction, "Module form of the host FWHT (src/fwht/cpp/fwht.py:21-30).")
