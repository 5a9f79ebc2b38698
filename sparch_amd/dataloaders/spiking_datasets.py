"""
SHD / SSC loader with the reference's API (sparch/dataloaders/spiking_datasets.py):
`SpikingDataset(dataset_name, data_folder, split, nb_steps=100)` and
`load_shd_or_ssc(dataset_name, data_folder, split, batch_size, nb_steps=100, shuffle=True, workers=0)`,
whose batches keep the `(x, xlens, y)` collate contract (spiking_datasets.py:80-87).

What differs is WHERE the event lists become dense spike counts: the reference bins every sample on the
CPU (np.digitize + sparse -> dense, lines 66-78) and the trainer then uploads 280 KB per sample; here
`__getitem__` hands the raw `(times, units, label)` of a sample to the collate function, which uploads the
event lists (8 bytes per event) and bins the whole batch on the device with `sparch_bin_events` — the same
arithmetic (bit-exact against np.digitize + accumulate, tests/test_hip_parity.py::test_bin_events_*).
`dense_sample(index)` returns the reference's per-sample dense tensor for callers that want it.

h5py is imported when a dataset is opened (it is not installed in the offline build image: the class then
raises ImportError naming the package — there is no other source for the files' contents).
"""
import logging

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from ..functional import bin_events

logger = logging.getLogger(__name__)


def _open_h5(filename):
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - depends on the environment
        raise ImportError("sparch_amd.dataloaders: reading SHD/SSC needs the h5py package "
                          f"(file {filename})") from e
    return h5py.File(filename, "r")


class SpikingDataset(Dataset):
    """spiking_datasets.py:24-87.  `h5_file` (test hook): any mapping with ["spikes"]["times"],
    ["spikes"]["units"] and ["labels"] laid out like the dataset files."""

    def __init__(self, dataset_name, data_folder, split, nb_steps=100, h5_file=None, device="cuda"):
        self.device = device
        self.nb_steps = nb_steps
        self.nb_units = 700
        self.max_time = 1.4
        self.time_bins = np.linspace(0, self.max_time, num=self.nb_steps)
        filename = f"{data_folder}/{dataset_name}_{split}.h5"
        self.h5py_file = h5_file if h5_file is not None else _open_h5(filename)
        self.firing_times = self.h5py_file["spikes"]["times"]
        self.units_fired = self.h5py_file["spikes"]["units"]
        self.labels = np.array(self.h5py_file["labels"], dtype=np.int64)  # np.int in the reference (line 61)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        """Raw events of one sample: (times float32[n], units int32[n], label)."""
        return (np.asarray(self.firing_times[index], np.float32), np.asarray(self.units_fired[index], np.int32),
                int(self.labels[index]))

    def dense_sample(self, index):
        """The reference's `__getitem__` result (lines 66-78): (dense (nb_steps, nb_units) float32 on the
        CPU, label) — host-side, for inspection; the training path does not use it."""
        times = np.digitize(self.firing_times[index], self.time_bins)
        units = np.asarray(self.units_fired[index], np.int64)
        x = torch.zeros(self.nb_steps, self.nb_units)
        x.index_put_((torch.from_numpy(times.astype(np.int64)), torch.from_numpy(units)),
                     torch.ones(len(times)), accumulate=True)
        return x, int(self.labels[index])

    def generateBatch(self, batch):
        """(xs (B, nb_steps, nb_units) on the device, xlens (B,), ys (B,)) — spiking_datasets.py:80-87 with
        the binning done once per batch on the device."""
        times, units, ys = zip(*batch)
        xs, _ = bin_events(times, units, self.nb_steps, self.nb_units, self.max_time, device=self.device)
        xlens = torch.tensor([self.nb_steps] * len(ys))
        return xs, xlens, torch.LongTensor(ys)


def load_shd_or_ssc(dataset_name, data_folder, split, batch_size, nb_steps=100, shuffle=True, workers=0,
                    h5_file=None, device="cuda", rank=0, world=1, seed=0):
    """spiking_datasets.py:90-140.  rank / world (data-parallel runs; not in the reference, which is single
    device): every rank reads the same file and draws a disjoint 1/world share of each epoch's (shuffled)
    sample order through a DistributedSampler — call `loader.sampler.set_epoch(e)` per epoch; `batch_size`
    is the PER-RANK batch."""
    if dataset_name not in ["shd", "ssc"]:
        raise ValueError(f"Invalid dataset name {dataset_name}")
    if split not in ["train", "valid", "test"]:
        raise ValueError(f"Invalid split name {split}")
    if dataset_name == "shd" and split == "valid":
        logging.info("SHD does not have a validation split. Using test split.")
        split = "test"
    if workers != 0:
        raise ValueError("sparch_amd.dataloaders: the collate function bins on the GPU; use workers=0 "
                         "(the reference's default)")
    dataset = SpikingDataset(dataset_name, data_folder, split, nb_steps, h5_file=h5_file, device=device)
    logging.info(f"Number of examples in {split} set: {len(dataset)}")
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler

        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed)
        return DataLoader(dataset, batch_size=batch_size, collate_fn=dataset.generateBatch, sampler=sampler,
                          num_workers=0, pin_memory=False)
    return DataLoader(dataset, batch_size=batch_size, collate_fn=dataset.generateBatch, shuffle=shuffle,
                      num_workers=0, pin_memory=False)
