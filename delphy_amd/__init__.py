"""delphy_amd -- MI355X-native engine for Delphy's EMAT local-move hot path.

The product is the HIP library `libemat_hip.so` (kernels + C-ABI, see include/emat_backend.h and
include/emat_host.h); this package is the Python host-side mirror of that interface (ctypes), used by
bench.py, __graft_entry__.py and the tests.  There is no CPU fallback: constructing an `EmatBackend`
without the built library or without a HIP device raises.
"""
from .engine import (  # noqa: F401
    EmatBackend,
    EmatError,
    EmatMultiRun,
    EmatRun,
    FlatTree,
    PopModel,
    SynthParams,
    TipDescs,
    build_library,
    hky_q_matrix,
    library_build_id,
    library_path,
    load_library,
    make_synthetic_emat,
    source_build_id,
)
