"""
sparch_amd — MI355X (gfx950) native implementation of sparch's recurrent spiking-layer
training path.  `sparch_amd.snns` mirrors the reference's `sparch.models.snns` API; the
arithmetic runs in libsparch_hip.so (hand-written HIP), reached through `_capi` (ctypes).
There is no CPU fallback: importing this package without the built library fails.
"""
from . import _capi  # noqa: F401  (fails loudly when libsparch_hip.so is missing)
from . import optim  # noqa: F401
from .functional import bin_events, check_status, compute_dtype, fbank, set_compute_dtype  # noqa: F401
from .snns import (SNN, LIFLayer, RLIFLayer, RadLIFLayer, ReadoutLayer,  # noqa: F401
                   SpikeFunctionBoxcar, adLIFLayer)

__version__ = "0.1.0"
