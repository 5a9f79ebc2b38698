"""`sparch.models.anns` -> sparch_amd.anns (same public names as the reference module)."""
from sparch_amd.anns import (ANN, GRULayer, LiGRULayer, MLPLayer, ReadoutLayerANN,  # noqa: F401
                             RNNLayer)
