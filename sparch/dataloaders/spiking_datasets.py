"""`sparch.dataloaders.spiking_datasets` -> sparch_amd.dataloaders.spiking_datasets."""
from sparch_amd.dataloaders.spiking_datasets import SpikingDataset, load_shd_or_ssc  # noqa: F401
