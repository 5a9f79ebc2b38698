"""CPU: the C-ABI library loads and exports every symbol include/sparch_hip.h declares, and the
Python binding has a prototype for each (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from sparch_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sparch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sparch_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _capi.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(_capi.PROTOTYPES) == names, "binding table and header disagree"


def test_prototype_arity_matches_header():
    text = open(os.path.join(ROOT, "include", "sparch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for m in re.finditer(r"\b(sparch_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert len(_capi.PROTOTYPES[name][1]) == n, (name, n, len(_capi.PROTOTYPES[name][1]))


def test_host_only_entry_points():
    lib = _capi.lib
    assert lib.sparch_abi_version() == 5
    assert _capi.strerror(0) == "ok" and "workspace" in _capi.strerror(-3)
    assert lib.sparch_fbank_frames(16000) == 98 and lib.sparch_fbank_frames(399) == 0
    assert lib.sparch_vpack_bytes(1024) == 1024 * 1024 * 6   # three bf16 planes per fp32 element
    assert lib.sparch_vpack_bytes(100) == 4 * 4 * 1024 * 6  # 4 column tiles x (4 waves x 1 k-group of 32)
    assert lib.sparch_vpack_bytes(2048) == 0                 # V slice would not fit the register file: step path
    # forward granules (T x row tiles x column tiles x 32 rows x 8 B) + one agreement word per workgroup
    assert lib.sparch_rec_chan_bytes(256, 250, 1024) == 250 * 8 * 32 * 32 * 8 + 8 * 32 * 4
    # one workgroup: ring of 4 x 6 KiB plane tiles + ONE 4-byte table word, rounded up to 16 bytes (a caller sizing
    # its buffer in 8-byte words must not lose the table word)
    assert lib.sparch_rec_chan_bytes(9, 14, 8) == 4 * 6144 + 16
    assert lib.sparch_gemm_tn_workspace_bytes(1024, 1024, 64000) == 16 * 1024 * 1024 * 4
    assert lib.sparch_bn_bwd_workspace_bytes(64000, 1024) == 2 * 250 * 1024 * 4


def test_argument_validation_returns_codes_without_launching():
    lib = _capi.lib
    assert lib.sparch_gemm_nt(0, 4, 4, None, 4, None, 4, None, 4, None, None, None) == -1
    assert lib.sparch_readout_fwd(2, 3, 257, 1, None, None, 1, 1, 1, None, None) == -1  # C > 256
    assert lib.sparch_cell_fwd(2, 1, 1, 1, 4, 16, None, None, 16, None, None, None, 16, None, 16,
                               1.0, 0.0, 0, 16, None, None, None, 0, None, None) == -1  # kind RLIF on non-recurrent entry


def test_operand_precision_is_a_per_call_argument():
    """ABI v5: the library keeps no precision state (rounds 1-2: a process-wide sparch_set_operand_precision).  Every
    entry point that multiplies takes `precision` as its last argument; an unknown value is SPARCH_EINVAL before
    anything is launched (0 bytes for the workspace queries); the Python module passes its own setting per call."""
    from sparch_amd import functional as Fn
    lib = _capi.lib
    assert not hasattr(lib, "sparch_set_operand_precision") or "sparch_set_operand_precision" not in _capi.PROTOTYPES
    assert Fn.compute_dtype() == "fp32" and Fn._prec() == 0
    assert lib.sparch_gemm_spike_tn_workspace_bytes(128, 128, 4096, 0) > 0
    assert lib.sparch_gemm_spike_tn_workspace_bytes(128, 128, 4096, 1) > 0
    assert lib.sparch_gemm_spike_tn_workspace_bytes(128, 128, 4096, 7) == 0
    assert lib.sparch_gemm6_nn(4, 4, 4, None, 4, None, 4, None, 4, None, 7) == -1
    assert lib.sparch_vpack(64, None, 0, None, None, None, 5) == -1
    prev = Fn.set_compute_dtype("bf16")
    try:
        assert prev == "fp32" and Fn.compute_dtype() == "bf16" and Fn._prec() == 1
        with pytest.raises(ValueError):
            Fn.set_compute_dtype("fp8")
    finally:
        Fn.set_compute_dtype("fp32")
