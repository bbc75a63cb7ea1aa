#!/usr/bin/env python3
"""Development check of the wave-per-parameter kernel (needs a GPU): parity against the general path and the oracle on a
few shapes with readable diagnostics, then timings.  python tools/dev_wave.py [quick]"""
import os, sys, time
os.environ.setdefault("MDT_TEST_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as g
from bench import synth, timed_ms
from oracle import c_oracle as co
pkg = g.load_package()

def dev(x):
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 1, 0)))).cuda()
    return t.permute(2, 1, 0)

def relerr(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    nn = np.isnan(a) != np.isnan(b)
    m = ~np.isnan(a) & ~np.isnan(b)
    e = np.zeros(a.shape)
    e[m] = np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300)
    e[nn] = np.inf
    return e

rng = np.random.default_rng(3)
from conftest import ar1
for (D, C, s) in [(1000, 4, 2), (500, 4, 2), (997, 4, 2), (250, 8, 2), (1000, 4, 3), (640, 6, 1)]:
    x = ar1(0.5, D, C, 12, rng)
    x[:, :, 1] = rng.standard_normal((D, C))
    x[:, :, 2] = ar1(0.95, D, C, 1, rng)[:, :, 0]
    x[:, :, 3] = np.round(x[:, :, 3], 3)
    x[::2, :, 4] = -x[1::2, :, 4][: x[::2, :, 4].shape[0]] if D % 2 == 0 else x[::2, :, 4]
    x[3, 1, 5] = np.inf
    xd = dev(x)
    os.environ["MDT_WAVE"] = "2"
    a = [o.cpu().numpy() for o in pkg.bulk_tail(xd, split_chains=s)]
    os.environ["MDT_WAVE"] = "0"
    b = [o.cpu().numpy() for o in pkg.bulk_tail(xd, split_chains=s)]
    del os.environ["MDT_WAVE"]
    want = co.bulk_tail(x, split_chains=s)
    names = ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")
    for i, nm in enumerate(names):
        e1, e2 = relerr(a[i], b[i]), relerr(a[i], want[i])
        flag = "" if (e1.max() <= (0 if "rhat" in nm else 1e-9) and e2.max() <= 1e-9) else "   <<<<<< MISMATCH"
        print(f"{D}x{C} s={s} {nm:10s} wave-vs-general max {e1.max():.2e} (arg {int(e1.argmax())})  wave-vs-oracle max {e2.max():.2e} (arg {int(e2.argmax())}){flag}")
        if flag:
            print("   wave   ", a[i]); print("   general", b[i]); print("   oracle ", want[i])
    sys.stdout.flush()

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
for (D, C, P) in [(1000, 4, 100000), (500, 4, 100000), (250, 4, 200000), (1000, 2, 100000), (600, 6, 60000)]:
    x = synth(D, C, P, torch.device("cuda", 0), 4321).permute(2, 1, 0)
    for wv in ("2", "0"):
        os.environ["MDT_WAVE"] = wv
        ms = timed_ms(lambda: pkg.bulk_tail(x, sync=False), 5, 2)
        print(f"{D}x{C}x{P} MDT_WAVE={wv}: bulk+tail {ms:.2f} ms  ({D*C*P*8/ms/1e6:.0f} GB/s of input)  status {pkg.status()}")
        sys.stdout.flush()
    del os.environ["MDT_WAVE"]
    del x
