"""`sparch.exp` -> sparch_amd.exp (the reference's `from sparch.exp import Experiment`)."""
from sparch_amd.exp import Experiment  # noqa: F401
