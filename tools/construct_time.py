import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, fiat_amd
from fiat_amd import runtime
runtime.Context.get()
T = fiat_amd.ufc_simplex(3); Tr = fiat_amd.ufc_simplex(2)
cases = [("Lagrange P1 tri", lambda: fiat_amd.Lagrange(Tr, 1)), ("Lagrange P3 tet", lambda: fiat_amd.Lagrange(T, 3)),
         ("DG P6 tet", lambda: fiat_amd.DiscontinuousLagrange(T, 6)), ("N2 tet", lambda: fiat_amd.Nedelec(T, 2)),
         ("RT2 tet", lambda: fiat_amd.RaviartThomas(T, 2)), ("BDM2 tet", lambda: fiat_amd.BrezziDouglasMarini(T, 2)),
         ("Regge1 tet", lambda: fiat_amd.Regge(T, 1)), ("P4 line", lambda: fiat_amd.Lagrange(fiat_amd.ufc_simplex(1), 4))]
for name, mk in cases:
    mk(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); el = mk(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); tab = el.tabulate(1, np.full((1, el.get_reference_element().get_spatial_dimension()), 0.2)); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); tab = el.tabulate(1, np.full((1, el.get_reference_element().get_spatial_dimension()), 0.2)); t2 = time.perf_counter() - t0
    print(f"{name:18s} construct {min(ts)*1e3:7.2f} ms (median {sorted(ts)[2]*1e3:7.2f})   first tabulate {t1*1e3:7.2f} ms, second {t2*1e3:7.2f} ms")
