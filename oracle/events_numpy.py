"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

NumPy restatement of the reference's event binning, SpikingDataset.__getitem__
(spiking_datasets.py:54, 66-78): times = np.digitize(firing_times, np.linspace(0, 1.4, nb_steps));
dense (nb_steps, nb_units) tensor with 1 added per (time bin, unit) event (torch.sparse -> to_dense sums
duplicates).  Parity: the statements are numpy calls the reference makes verbatim; the only torch step
(sparse -> dense) is restated as np.add.at.  The reference raises for events the sparse constructor cannot
place (bin == nb_steps, i.e. t >= max_time); here they are dropped and counted, like the HIP kernel does.
"""
import numpy as np


def bin_sample(times, units, nb_steps=100, nb_units=700, max_time=1.4):
    bins = np.linspace(0, max_time, num=nb_steps)            # spiking_datasets.py:54
    idx = np.digitize(np.asarray(times), bins)               # spiking_datasets.py:68
    units = np.asarray(units).astype(np.int64)
    ok = (np.asarray(times) >= 0) & (idx < nb_steps) & (units >= 0) & (units < nb_units)
    x = np.zeros((nb_steps, nb_units), np.float32)
    np.add.at(x, (idx[ok], units[ok]), 1.0)                  # sparse -> dense, duplicates add (72-78)
    return x, int((~ok).sum())
