"""`sparch.models.snns` -> sparch_amd.snns (same public names as the reference module)."""
from sparch_amd.snns import (SNN, LIFLayer, RLIFLayer, RadLIFLayer, ReadoutLayer,  # noqa: F401
                             SpikeFunctionBoxcar, adLIFLayer)
