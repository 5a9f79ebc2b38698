"""Import-path shim: `sparch.*` resolves to the MI355X implementation in `sparch_amd`, so code written
against the reference (`from sparch.models.snns import SNN`, `from sparch.exp import Experiment`) and
whole-module checkpoints pickled under those paths keep working."""
