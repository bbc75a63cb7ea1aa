import sys, time, numpy as np, torch
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
pkg = g.load_package()
x = torch.randn(10, 4, 1000, dtype=torch.float64, device="cuda").permute(2, 1, 0)
for name, fn in (("bulk_tail sync", lambda: pkg.bulk_tail(x)), ("bulk_tail async", lambda: pkg.bulk_tail(x, sync=False)),
                 ("ess_rhat rank", lambda: pkg.ess_rhat(x)), ("rhat rank", lambda: pkg.rhat(x)), ("mcse mean", lambda: pkg.mcse(x))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    print(f"{name:18s} {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us per call (config 1: 1000 x 4 x 10)")
xh = x.cpu().numpy()
t0 = time.perf_counter()
for _ in range(50):
    pkg.bulk_tail(xh)
print(f"{'bulk_tail host':18s} {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us per call (host arrays, staged)")
