"""whvi_amd -- MI355X-native implementation of WHVI's data-parallel hot path.

Host-side mirror of the reference's interface for this path (same names, argument meaning and
error behaviour) on top of hand-written gfx950 HIP kernels reached through a C ABI
(``include/whvi_hip.h`` -> ``whvi_amd/libwhvi_hip.so``):

    whvi_amd.fwht.cuda / .cpp / .python   FWHTFunction (+ FWHT, WHT_matmul)  <- src/fwht/*/fwht.py
    whvi_amd.utils                        matmul_diag_*, kl_diag_normal, build_H  <- src/utils.py
    whvi_amd.weights / .layers            WHVISquarePow2Matrix ... WHVILinear     <- src/weights.py, src/layers.py
    whvi_amd.networks / .likelihoods      WHVIRegression, GaussianLikelihood      <- src/networks.py, src/likelihoods.py
    whvi_amd.parallel                     MC-sample / row sharding over RCCL      (new)
"""
__version__ = "0.1.0"
