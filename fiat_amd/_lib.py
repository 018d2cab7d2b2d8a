"""ctypes binding of libfiat_amd.so (include/fiat_amd.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc,
gfx950).  There is no fallback: if the library is missing the import fails, and
if no GPU is present every compute entry point raises."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint, c_void_p

import numpy as np
# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  Import it
# BEFORE loading libfiat_amd.so so that both share ONE runtime instance (device
# memory and streams are handed over from torch); loading ours first would pull a
# second runtime from /opt/rocm that does not see torch's devices.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FIAT_AMD_LIB", os.path.join(_HERE, "csrc", "libfiat_amd.so"))  # override: A/B builds

FX_OK = 0
FX_EINVAL, FX_ENOTIMPL, FX_ESINGULAR, FX_EHIP, FX_ENOMEM = -1, -2, -3, -4, -5
VARIANTS = {None: 0, "bubble": 1, "dual": 2}
# kernel-selection policy bits (include/fiat_amd.h FX_POLICY_*)
POLICY = {"no_fixed": 1 << 0, "no_small": 1 << 1, "no_stacked": 1 << 2, "no_coop": 1 << 3, "stacked_small": 1 << 4,
          "no_stacked_mix": 1 << 5, "no_shared_wave": 1 << 6, "no_shared_reg": 1 << 7, "no_macro_small": 1 << 8,
          "kernel_image": 1 << 9, "kernel_stream": 1 << 10, "no_wg": 1 << 11, "wg_small": 1 << 12, "no_small_values": 1 << 13}


class FiatAmdError(RuntimeError):
    pass


class LinAlgError(np.linalg.LinAlgError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()'); fiat_amd has no CPU fallback")

lib = ctypes.CDLL(LIB_PATH)

_p_d = POINTER(c_double)
_p_i = POINTER(c_int)

_SIGS = {
    "fx_last_error": (c_char_p, []),
    "fx_abi_version": (c_int, []),
    "fx_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "fx_ctx_destroy": (c_int, [c_void_p]),
    "fx_ctx_info": (c_int, [c_void_p, _p_i, _p_i, c_char_p, c_int]),
    "fx_ctx_set_policy": (c_int, [c_void_p, c_uint]),
    "fx_ctx_get_policy": (c_int, [c_void_p, POINTER(c_uint)]),
    "fx_element_create": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_int, c_int, c_void_p,
                                  POINTER(c_void_p)]),
    "fx_element_destroy": (c_int, [c_void_p]),
    "fx_element_set_coeffs": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "fx_element_dims": (c_int, [c_void_p, _p_i, _p_i, _p_i, _p_i, _p_i]),
    "fx_num_tables": (c_int, [c_int, c_int]),
    "fx_tabulate_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_tabulate_batch_host": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_riesz_assemble": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_vandermonde_solve_batch": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p]),
    "fx_line_element_create": (c_int, [c_void_p, c_int, c_void_p, POINTER(c_void_p)]),
    "fx_line_element_destroy": (c_int, [c_void_p]),
    "fx_line_tabulate_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_tensor_tabulate_batch": (c_int, [c_void_p, c_int, POINTER(c_void_p), c_int, c_int64, c_int, c_void_p,
                                         c_void_p, c_void_p]),
    "fx_tensor_tabulate_grid_batch": (c_int, [c_void_p, c_int, POINTER(c_void_p), c_int, c_int64, c_int, c_void_p,
                                              c_void_p, c_void_p]),
    "fx_prism_tabulate_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_plan_steps": (c_int, [c_int, c_int, c_int, c_double, c_int, _p_i, _p_d, c_void_p, c_void_p]),
    "fx_plan_c0_transform": (c_int, [c_int, c_int, c_void_p]),
    "fx_plan_coop": (c_int, [c_int, c_int, c_int, c_double, c_int, _p_i, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fx_pushforward_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_tabulate_batch_mapped": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "fx_tabulate_batch_shared": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "fx_collapsed_quadrature": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_tables_squared_norm": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_classify_tables": (c_int, [c_void_p, c_int64, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "fx_tables_point_major": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_plan_kernel": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_char_p, c_int]),
    "fx_macro_element_create": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p, c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "fx_macro_element_destroy": (c_int, [c_void_p]),
    "fx_macro_element_set_coeffs": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "fx_macro_tabulate_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "fx_table_outer_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "fx_map_points": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "fx_jacobi_batch": (c_int, [c_void_p, c_double, c_double, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "fx_comm_available": (c_int, []),
    "fx_comm_unique_id": (c_int, [c_void_p]),
    "fx_comm_create": (c_int, [c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "fx_comm_destroy": (c_int, [c_void_p]),
    "fx_allgather_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "fx_ctx_check": (c_int, [c_void_p, c_void_p]),
    "fx_time_tabulate_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, POINTER(c_float)]),
}

EXPORTS = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = ABI mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc):
    """Map C status codes to the exception types the reference raises."""
    if rc >= 0:
        return rc
    msg = lib.fx_last_error().decode()
    if rc == FX_EINVAL:
        raise ValueError(msg)
    if rc == FX_ENOTIMPL:
        raise NotImplementedError(msg)
    if rc == FX_ESINGULAR:
        raise LinAlgError(msg)
    raise FiatAmdError(msg)


def host_ptr(arr):
    return arr.ctypes.data_as(c_void_p)


def plan_steps(sd, n, variant=None, scale=0.0):
    """Recurrence step table of one expansion set (host computation only)."""
    cap = 4096
    ints = np.zeros((cap, 4), dtype=np.int32)
    coefs = np.zeros((cap, 3), dtype=np.float64)
    nsteps = c_int(0)
    phi0 = c_double(0.0)
    check(lib.fx_plan_steps(sd, n, VARIANTS[variant], float(scale), cap, ctypes.byref(nsteps), ctypes.byref(phi0),
                            host_ptr(ints), host_ptr(coefs)))
    k = nsteps.value
    return phi0.value, ints[:k].copy(), coefs[:k].copy()


def plan_coop(sd, n, variant=None, scale=0.0):
    """Schedule of the cooperative kernel (host computation only): dict with KS, per-producer
    entries (level, seed, publish, member), kstart and kperm."""
    cap, cap_k = 512, 128
    ks = c_int(0)
    nent = np.zeros(4, dtype=np.int32)
    ints = np.zeros((4, cap, 5), dtype=np.int32)
    kstart = np.zeros((4, cap_k), dtype=np.int32)
    kperm = np.full(4 * cap_k, -2, dtype=np.int32)
    check(lib.fx_plan_coop(sd, n, VARIANTS[variant], float(scale), cap, ctypes.byref(ks), host_ptr(nent), host_ptr(ints),
                           host_ptr(kstart), cap_k, host_ptr(kperm)))
    K = ks.value
    return {"KS": K, "entries": [ints[w, :nent[w], :4].copy() for w in range(4)],
            "kstart": [kstart[w, :K + 1].copy() for w in range(4)], "kperm": kperm[:4 * K].copy()}


def plan_c0_transform(sd, n):
    import math
    nexp = math.comb(n + sd, sd)
    T = np.zeros((nexp, nexp))
    check(lib.fx_plan_c0_transform(sd, n, host_ptr(T)))
    return T
