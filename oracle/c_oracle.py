"""TEST INFRASTRUCTURE -- ctypes wrapper of the C restatement (oracle/fiat_oracle.c).
Used by tests/ (parity at full batch size) and by bench.py's cpu_baseline leg only."""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
VARIANTS = {None: 0, "bubble": 1, "dual": 2}


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def load():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    lib.fo_tabulate_batch.restype = ctypes.c_int
    lib.fo_tabulate_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.fo_max_threads.restype = ctypes.c_int
    return lib


def tabulate_batch(cell, n, coeffs, order, pts, verts=None, scale=None, variant=None, nthreads=0):
    """pts (nreq, npts, sd) -> (nreq, ntab, rows, npts); rows = prod(coeffs.shape[:-1])."""
    lib = load()
    cell = np.ascontiguousarray(cell, dtype=np.float64)
    sd = cell.shape[1]
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    nreq, npts = pts.shape[0], pts.shape[1]
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    nexp = coeffs.shape[-1]
    rows = coeffs.size // nexp
    if scale is None:
        vol = 1.0
        for i in range(1, sd + 1):
            vol *= 2.0 / i
        scale = math.sqrt(1.0 / vol)
        if n == 0 and sd > 1:
            scale = 1.0
    ntab = math.comb(sd + order, sd)
    out = np.empty((nreq, ntab, rows, npts))
    v = None if verts is None else np.ascontiguousarray(verts, dtype=np.float64)
    rc = lib.fo_tabulate_batch(sd, n, VARIANTS[variant], float(scale), cell.ctypes.data, coeffs.ctypes.data, rows, order,
                               nreq, npts, pts.ctypes.data, None if v is None else v.ctypes.data, out.ctypes.data,
                               int(nthreads))
    if rc != 0:
        raise ValueError("fo_tabulate_batch: bad arguments")
    return out


def max_threads():
    return load().fo_max_threads()
