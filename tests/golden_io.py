"""Loader for tests/golden/*.npz (written by tools/gen_golden.py from the real reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SNN_CASES = [
    "snn_cfg1_LIF",
    "snn_adLIF_bn",
    "snn_RLIF_nonorm_bias",
    "snn_RadLIF_bn",
    "snn_RadLIF_bidir_bn",
    "snn_adLIF_layernorm",
    "snn_LIF_noreadout",
]
CELL_KINDS = ["LIF", "adLIF", "RLIF", "RadLIF"]
# fully dyadic networks (tools/gen_golden.py gen_snn_dyadic): spikes reproducible bit for bit in any summation order
DYADIC_CASES = ["dyadic_RadLIF_none", "dyadic_RLIF_none_bias", "dyadic_RadLIF_bidir_none", "dyadic_RadLIF_bn"]
DYADIC_LONG = "dyadic_RadLIF_T1000"


def layer_spikes(z, k):
    """Hidden layer k's spike train of a dyadic fixture (stored bit-packed) as a float32 array."""
    shape = tuple(int(v) for v in z[f"spikes.{k}.shape"])
    n = int(np.prod(shape))
    return np.unpackbits(z[f"spikes.{k}"])[:n].reshape(shape).astype(np.float32)


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def snn_case(name):
    """Returns (cfg, x, y, params, init_states, z) with torch CPU tensors."""
    z = load(name)
    cfg = json.loads(str(z["cfg"]))
    x = torch.from_numpy(z["x"].astype(np.float32))
    y = torch.from_numpy(z["y"])
    params = {k[len("param."):]: torch.from_numpy(v) for k, v in z.items() if k.startswith("param.")}
    n = len(cfg["layer_sizes"])
    init = []
    for i in range(n):
        st = {}
        for k in ("u0", "w0", "s0"):
            key = f"init.{i}.{k}"
            if key in z:
                st[k] = torch.from_numpy(z[key])
        init.append(st)
    return cfg, x, y, params, init, z
