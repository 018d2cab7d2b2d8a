import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, bench
el, sd, deg, order, npts, batch = bench.build_element("p3tet")
ps = el.device_polyset()
pts = torch.as_tensor(bench.synth_points(sd, batch, npts, 2)).cuda()
out = torch.empty(ps.out_shape(order, batch, npts), dtype=torch.float64, device="cuda")
for steps in (20, 50, 200):
    for _ in range(5): ps.tabulate_batch(order, pts, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(steps): ps.tabulate_batch(order, pts, out=out)
    t_enq = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"steps={steps}: wall {t/steps*1e6:.1f} us/step, enqueue {t_enq/steps*1e6:.1f} us/step, events {e0.elapsed_time(e1)/steps*1e3:.1f} us/step, C-loop {ps.time_tabulate_batch(order, pts, None, out, steps)*1e3:.1f} us")
