#!/usr/bin/env python3
"""Timing (needs a GPU): (iterations, parameters, chains) storage -- what MCMCChains keeps -- consumed in place through
chain_stride / param_stride, against the out-of-place permutation pass (mdt_layout_from_dpc) followed by the dense call.

    python tools/time_strided.py [--params 10000]
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
from bench import synth, timed_ms  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--params", type=int, default=10000)
a = ap.parse_args()
pkg = g.load_package()
D, C, P = 10000, 4, a.params
dense = synth(D, C, P, torch.device("cuda", 0), 1234)            # memory (params, chains, draws)
store = dense.permute(1, 0, 2).contiguous()                        # memory (chains, params, draws) = Julia's (draws, params, chains)
view = store.permute(2, 0, 1)                                      # logical (draws, chains, params), strides (1, D*P, D)
xd = dense.permute(2, 1, 0)
t_dense = timed_ms(lambda: pkg.bulk_tail(xd, sync=False), 10, 2)
t_strided = timed_ms(lambda: pkg.bulk_tail(view, sync=False), 10, 2)
t_perm = timed_ms(lambda: pkg.from_mcmcchains_layout(store.permute(2, 1, 0)), 10, 2)
same = all(torch.equal(u, v) for u, v in zip(pkg.bulk_tail(view), pkg.bulk_tail(xd)))
print(f"{D}x{C}x{P} f64: dense {t_dense:.2f} ms | strided in place {t_strided:.2f} ms | permutation pass {t_perm:.2f} ms + dense "
      f"= {t_perm + t_dense:.2f} ms (and a second {D * C * P * 8 / 1e9:.1f} GB buffer) | bit-equal: {same}")
