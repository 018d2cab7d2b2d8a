import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, fiat_amd, bench
from fiat_amd import runtime
runtime.Context.get()
els = [fiat_amd.Lagrange(fiat_amd.ufc_simplex(3), k) for k in (1, 2, 3)] + [fiat_amd.Nedelec(fiat_amd.ufc_simplex(3), 1), fiat_amd.RaviartThomas(fiat_amd.ufc_simplex(2), 1)]
work = []
for i in range(60):
    el = els[i % len(els)]
    sd = el.get_reference_element().get_spatial_dimension()
    npts = [4, 11, 23][i % 3]
    nreq = 500
    pts = torch.as_tensor(bench.synth_points(sd, nreq, npts, i)).cuda()
    out = torch.empty(el.device_polyset().out_shape(1, nreq, npts), dtype=torch.float64, device="cuda")
    work.append((el, pts, out))
def run(stream=None):
    for el, pts, out in work:
        el.tabulate_batch(1, pts, out=out, stream=stream)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize()
direct = (time.perf_counter() - t0) / 20
ref = [o.clone() for _, _, o in work]
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run(s)
torch.cuda.synchronize()
for _, _, o in work: o.zero_()
with torch.cuda.graph(g, stream=s):
    run(s)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
ok = all(torch.equal(a, o) for a, (_, _, o) in zip(ref, work))
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 20
print(f"60 launches of 500 requests: direct {direct*1e3:.3f} ms, graph replay {graph*1e3:.3f} ms, equal {ok}")
