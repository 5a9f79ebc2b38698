"""
ctypes binding of libsparch_hip.so (include/sparch_hip.h).

This is the only place the Python host touches native code.  Signatures carry
plain pointers and sizes (no torch types); tensors are passed as
`tensor.data_ptr()` and the stream as `torch.cuda.current_stream().cuda_stream`.
There is NO CPU fallback: if the library is missing or a call fails, we raise.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_uint64, c_void_p

# PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's).  It must be in the
# process BEFORE libsparch_hip.so is dlopen'ed so that both bind ONE HIP runtime (one device
# context, shared streams and allocations); loaded the other way round our kernels would be
# registered with a second, device-less runtime ("no ROCm-capable device is detected").
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPARCH_HIP_LIB") or os.path.join(_HERE, "libsparch_hip.so")  # override: diagnostic builds

SPARCH_OK = 0
KIND = {"LIF": 0, "adLIF": 1, "RLIF": 2, "RadLIF": 3}

P = c_void_p  # device (or host-array) pointer

# name -> (restype, argtypes); kept in the header's order.  tests/test_capi.py checks this
# table against every `sparch_*` declaration in include/sparch_hip.h.
PROTOTYPES = {
    "sparch_abi_version": (c_int, []),
    "sparch_strerror": (c_char_p, [c_int]),
    "sparch_last_hip_error": (c_char_p, []),
    "sparch_device_cus": (c_int, []),
    "sparch_gemm_nt": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, P, P]),
    "sparch_gemm_nn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P]),
    "sparch_gemm_tn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sparch_gemm_tn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, P, c_size_t, P]),
    "sparch_gemm_spike_nt": (c_int, [c_int, c_int, c_int, P, c_int, c_float, P, c_int, P, c_int, P, P, P, c_int]),
    "sparch_gemm_spike_tn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sparch_gemm_spike16_nt": (c_int, [c_int, c_int, c_int, P, c_int, c_float, P, c_int, P, c_int, P, P, P, c_int]),
    "sparch_gemm_spike16_tn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_float, P, c_int, c_int,
                                       c_int, P, c_size_t, P, c_int]),
    "sparch_gemm_spike_tn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_float, P, c_int, c_int,
                                     c_int, P, c_size_t, P, c_int]),
    "sparch_gemm6_splitk_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sparch_gemm6_nt_splitk": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_size_t, P, c_int]),
    "sparch_gemm6_nn_splitk": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_size_t, P, c_int]),
    "sparch_gemm6_nt": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, P, P, c_int]),
    "sparch_gemm6_nn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int]),
    "sparch_split3": (c_int, [c_size_t, P, P, P]),
    "sparch_gemm_spike16_nt_wp": (c_int, [c_int, c_int, c_int, P, c_int, c_float, P, P, c_int, P, c_int, P, P, P, c_int]),
    "sparch_gemm6_nn_wp": (c_int, [c_int, c_int, c_int, P, c_int, P, P, c_int, P, c_int, P, c_int]),
    "sparch_gemm6_tn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, P, c_size_t, P, c_int]),
    "sparch_flag_bf16_exact": (c_int, [c_size_t, P, P, P]),
    "sparch_gemm_auto_nt": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, P, P, P, c_int]),
    "sparch_gemm_auto_tn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, P, P,
                                    c_size_t, P, c_int]),
    "sparch_plane_bf16_exact": (c_int, [c_int, c_int, P, c_int, P, c_int, P, P]),
    "sparch_gemm_auto16_nt": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, P, P, P, P, c_int]),
    "sparch_gemm_auto16_tn": (c_int, [c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int,
                                      P, P, c_size_t, P, c_int]),
    "sparch_bn_finalize": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, c_float, c_int,
                                   P, P, P, P, P, P, P]),
    "sparch_bn_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sparch_bn_bwd_reduce": (c_int, [c_int, c_int, P, P, P, P, P, P, P, c_size_t, P]),
    "sparch_bn_bwd_apply": (c_int, [c_int, c_int, P, P, P, P, P, P, P, P, P]),
    "sparch_layernorm_fwd": (c_int, [c_int, c_int, c_int, P, P, P, c_float, P, P, P, P]),
    "sparch_layernorm_bwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "sparch_cell_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P,
                                c_float, c_float, c_uint64, P, P, P, P, c_int, P, P]),
    "sparch_cell_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, P, P, P, P, P, P,
                                c_float, c_float, c_uint64, P, P, P, P, P, P]),
    "sparch_vpack_bytes": (c_size_t, [c_int]),
    "sparch_set_xcd_local": (c_int, [c_int]),
    "sparch_vpack": (c_int, [c_int, P, c_int, P, P, P, c_int]),
    "sparch_vpack_both": (c_int, [c_int, P, P, P, P, P, c_int]),
    "sparch_vmask": (c_int, [c_int, P, P, P]),
    "sparch_rec_chan_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sparch_rec_cell_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P,
                                    c_float, c_float, c_uint64, P, P, P, P, c_int, P, P, c_size_t, P, c_int, P, c_int]),
    "sparch_rec_cell_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, P, P, P, P, P, P, P,
                                    c_float, c_float, c_uint64, P, P, P, P, P, P, P, c_size_t, P, c_int, P, c_int]),
    "sparch_rec_cell_step_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P,
                                         c_float, c_float, c_uint64, P, P, P, P, P, P, P]),
    "sparch_rec_cell_step_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P,
                                         P, c_float, c_float, c_uint64, P, P, P, P, P, P, P, P]),
    "sparch_colsum_clamped": (c_int, [c_int, c_int, c_int, P, P, P, P, P]),
    "sparch_add_halves": (c_int, [c_size_t, P, P, P]),
    "sparch_colsum": (c_int, [c_int, c_int, P, P, P, c_size_t, P]),
    "sparch_readout_fwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P]),
    "sparch_readout_bwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P]),
    "sparch_fbank_frames": (c_int, [c_int]),
    "sparch_fbank_fwd": (c_int, [c_int, c_int, c_int, P, P, P]),
    "sparch_bin_events": (c_int, [c_longlong, P, P, P, c_int, c_int, c_int, c_double, P, P, P]),
    "sparch_act_fwd": (c_int, [c_int, c_size_t, c_int, P, P, P, c_float, c_uint64, P, P]),
    "sparch_act_bwd": (c_int, [c_int, c_size_t, c_int, P, P, P, P, c_float, c_uint64, P, P]),
    "sparch_softmax_sum_fwd": (c_int, [c_int, c_int, c_int, P, P, P]),
    "sparch_softmax_sum_bwd": (c_int, [c_int, c_int, c_int, P, P, P, P]),
    "sparch_ann_rec_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_float, c_uint64, P, P, P, c_size_t,
                                   P, c_int, P]),
    "sparch_ann_rec_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_float, c_uint64, P, P, P, c_size_t,
                                   P, c_int, P]),
    "sparch_ann_rec_step_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_float, c_uint64, P, P,
                                        P, P]),
    "sparch_ann_rec_step_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_float, c_uint64, P, P, P,
                                        P]),
    "sparch_ligru_vpack_bytes": (c_size_t, [c_int, c_int]),
    "sparch_ligru_vpack": (c_int, [c_int, P, P, c_int, P, P, c_int]),
    "sparch_ligru_chan_bytes": (c_size_t, [c_int, c_int]),
    "sparch_ligru_fwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_float, c_uint64, P, P, P, P, P,
                                 c_size_t, P, c_int, P]),
    "sparch_ligru_bwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, c_uint64, P, P, P, P, P,
                                 c_size_t, P, c_int, P]),
    "sparch_mt19937_uniform_f32": (c_int, [P, P, c_size_t, P]),
    "sparch_gru_vpack_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sparch_gru_vpack": (c_int, [c_int, P, P, P, c_int, P, P, P, c_int]),
    "sparch_gru_chan_bytes": (c_size_t, [c_int, c_int]),
    "sparch_gru_fwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, c_float, c_uint64, P, P, P,
                               P, P, P, c_size_t, P, c_int, P]),
    "sparch_gru_bwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_float, c_uint64, P, P, P, P, P, P,
                               P, c_size_t, P, c_int, P]),
    "sparch_gate_step": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float, c_uint64, P]),
    "sparch_bn_bwd_apply_planes": (c_int, [c_int, c_int, P, P, P, P, P, P, P, P, P, P, P]),
    "sparch_gemm6_nn_pp": (c_int, [c_int, c_int, c_int, P, P, c_int, P, P, c_int, P, c_int, P, c_int]),
    "sparch_gemm_spike16_tn_ap": (c_int, [c_int, c_int, c_int, P, P, c_int, P, c_int, c_float, P, c_int, c_int, c_int,
                                          P, c_size_t, P, c_int]),
    "sparch_expand_counts_u8": (c_int, [ctypes.c_longlong, c_int, P, P, c_int, P, c_int, P]),
    "sparch_ce_loss": (c_int, [c_int, c_int, P, P, P, P, P]),
    "sparch_adam_scalars": (c_int, [P, P, c_double, c_double, P, P]),
    "sparch_adam_step": (c_int, [c_int, P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, c_float, P, P, P]),
}


class SparchHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: sparch_amd has no CPU fallback. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C sparch_amd/csrc`."
        )
    lib = ctypes.CDLL(LIB_PATH)
    runtimes = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    runtimes.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    if len(runtimes) > 1:
        raise ImportError(f"two HIP runtimes are mapped ({sorted(runtimes)}); libsparch_hip.so and "
                          "PyTorch must share one libamdhip64")
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = library/header mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def strerror(code):
    return lib.sparch_strerror(int(code)).decode()


def check(code, what):
    """Map a negative return code onto the reference's Python error convention."""
    if code == SPARCH_OK:
        return
    msg = f"{what}: {strerror(code)} (code {code})"
    if code == -4:
        msg += f" [HIP: {lib.sparch_last_hip_error().decode()}]"
    if code in (-1, -2):
        raise ValueError(msg)
    raise SparchHipError(msg)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
