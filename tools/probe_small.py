import sys; sys.path.insert(0,'.')
import fiat_amd as fa
from fiat_amd import runtime
el = fa.Lagrange(fa.ufc_simplex(2), 5)
ps = el.device_polyset()
for pol in [(), ("no_stacked",), ("no_small",)]:
    runtime.Context.get().set_policy(*pol)
    print(pol, [ps.kernel_name(0, 1000, n, instance=True) for n in (12, 16, 25, 30, 33)])
