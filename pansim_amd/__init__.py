"""pansim_amd -- MI355X (gfx950) drop-in for the per-generation hot path of bacpop/Pansim.

The package is a thin host-side mirror of the reference's `Population` API and main() loop
over the C ABI of libpansim_hip.so (include/pansim_hip.h).  All compute runs in hand-written
HIP kernels; there is no CPU fallback and importing the API without the built library fails.
"""
from ._lib import LIB_PATH, PansimError, load  # noqa: F401
from .population import (Population, draw_parents, fmt_f64, hamming_bitwise_fast, init_vector,  # noqa: F401
                         int_to_base, jaccard_distance_fast, sample_weights, standard_deviation)
from .simulation import (DEFAULTS, MultiSimulation, Simulation, derive, make_params, sample_pairs,  # noqa: F401
                         selection_coefficients, validate)

__version__ = "0.1.0"
