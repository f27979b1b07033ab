"""FWHT front-ends, one module per backend exactly as in the reference's src/fwht/ package:
``cuda`` (MI355X HIP kernels), ``cpp`` (native host library), ``python`` (torch ops)."""
