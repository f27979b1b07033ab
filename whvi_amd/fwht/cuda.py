"""GPU front-end on the MI355X HIP kernels; interface of src/fwht/cuda/fwht.py (``FWHTFunction.apply(x)``)."""
import fwht_cuda

from whvi_amd.fwht._frontends import make_fwht_function

__all__ = ["FWHTFunction"]

FWHTFunction = make_fwht_function(
    fwht_cuda.fwht, "FWHTFunction",
    "Batched FWHT of the rows of a 2-D GPU tensor through ``fwht_cuda.fwht`` (libwhvi_hip.so); new tensor out, "
    "input untouched; differentiable to any order.")
