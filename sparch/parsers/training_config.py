"""`sparch.parsers.training_config` -> sparch_amd.parsers."""
from sparch_amd.parsers import add_training_options, print_training_options  # noqa: F401
