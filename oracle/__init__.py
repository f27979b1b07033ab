"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's FWHT / WHVI weight-sample path, used ONLY as the
checker by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``whvi_amd/`` imports this package; the product path fails loudly when its
HIP library is missing instead of falling back to anything here.

Parity status: PINNED -- see tests/test_oracle.py (reference known-answer vectors, dense
Hadamard identity, golden fixtures from the live reference, and bit-equality with the
reference's own compiled C++ FWHT in oracle/_ref when that is present).
"""
from .binding import (  # noqa: F401
    build,
    fwht,
    fwht_descending,
    pipeline,
    dense_wht,
    hadamard,
    load_reference_cpp,
)
