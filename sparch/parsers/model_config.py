"""`sparch.parsers.model_config` -> sparch_amd.parsers."""
from sparch_amd.parsers import add_model_options, print_model_options  # noqa: F401
