#!/usr/bin/env python3
"""Diagnostics (needs a GPU): the folded ranking by merge (default) against the general ranking
(MDT_FOLD_MERGE=0) on the bench array -- bit-equality of all four outputs, and both timings.

    python tools/ab_fold.py [--params 10000] [--draws 10000] [--chains 4] [--ties]
"""
import argparse
import os
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
from bench import synth  # noqa: E402


def timed(pkg, x, reps):
    pkg.bulk_tail(x)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        out = pkg.bulk_tail(x, sync=False)
    ev1.record()
    torch.cuda.synchronize()
    return [o.clone() for o in out], ev0.elapsed_time(ev1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--params", type=int, default=10000)
    ap.add_argument("--draws", type=int, default=10000)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--ties", action="store_true", help="round a tenth of the parameters to a grid (heavy ties, symmetric pairs)")
    a = ap.parse_args()
    pkg = g.load_package()
    xs = synth(a.draws, a.chains, a.params, torch.device("cuda", 0), 1234)
    if a.ties:
        xs[::10] = torch.round(xs[::10] * 8) / 8
        xs[1::10, :, ::7] = -xs[1::10, :, 1::7][:, :, : xs[1::10, :, ::7].shape[2]]  # symmetric pairs: equal folds, distinct x
    x = xs.permute(2, 1, 0)
    os.environ["MDT_FOLD_MERGE"] = "1"
    o1, t1 = timed(pkg, x, a.reps)
    os.environ["MDT_FOLD_MERGE"] = "0"
    o0, t0 = timed(pkg, x, a.reps)
    names = ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")
    ok = True
    for nm, u, v in zip(names, o1, o0):
        same = torch.equal(u.view(torch.int64), v.view(torch.int64))
        nbad = int((u.view(torch.int64) != v.view(torch.int64)).sum())
        ok &= same
        print(f"{nm}: bit-equal={same} (differing {nbad})")
        if not same:
            idx = (u.view(torch.int64) != v.view(torch.int64)).nonzero()[:5, 0]
            print("   first:", idx.tolist(), u[idx].tolist(), v[idx].tolist())
    print(f"merge {t1:.3f} ms   general {t0:.3f} ms   ({a.draws}x{a.chains}x{a.params})")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
