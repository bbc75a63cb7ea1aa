#!/usr/bin/env python3
"""Diagnostics: where the fused kernel spends its cycles, by phase (needs a GPU).

    MDT_PROF=1 python tools/phase_profile.py [--params 2560] [--draws 10000] [--chains 4]

Reads the per-phase shader-clock totals the kernel accumulates when the context is created with
MDT_PROF=1 (mdt_debug_phase_cycles) and prints each phase's share.  Shares, not durations: the
stamps themselves perturb nothing but they are summed over all workgroups.
"""
import argparse
import ctypes
import os
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
import sys

os.environ.setdefault("MDT_PROF", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
from bench import synth  # noqa: E402

# What every slot covers -- a stamp charges the time since the previous stamp of thread 0 to its slot (MDT_PROF_MARK in
# csrc/mdt_device.hpp), so a slot is "whatever ran between two stamps", named here by the code between them.
# Slots 1-6 are stamped by the kernel body AFTER a phase returned: with the sub-slots below active they only hold the
# tail of the phase behind its last inner stamp.
NAMES = {1: "ranking of x: after its last round (shift K_j of every chain, return)",
         2: "bulk series: after its last sweep (outputs of the phase, constant-chain check)",
         3: "folded ranking |x - median| (fold merge, or the general ranking where no sorted copy exists)",
         4: "R-hat of the folded scores (combine)",
         5: "tail-ESS indicators (popcount autocovariance, Geyer)",
         6: "outputs",
         8: "  ranking: sample sort, splitters, grid LUT",
         9: "  ranking: first pass over x (tile-major partition of every chain; two-pass form: class histogram + tags)",
         10: "  ranking: class totals -> rounds",
         11: "  ranking: partition pass (two-pass form only) + per round: load, local bins, placement + score output",
         12: "  ranking: per round: count inside bins, sorted copy, score lookup"}
# z-series with a split chain resident in LDS (series_z_phase: the headline, configs 1, 4, 5)
NAMES_Z = {13: "  series: waiting for a chain's loads + scatter to time order in LDS + exact moments",
           14: "  series: barrier per chain",
           7: "  series: next chain's loads issued + centring terms of the chain (wave 0)",
           15: "  series: lag products (+ indicator ballots, first sweep)"}
# long chains (series_win_phase: config 3): a window = a partition tile of a chain, gathered run by run
NAMES_W = {7: "  series (windows): barrier behind the products of the window before (waves waiting for the slowest)",
           13: "  series (windows): waiting for the window's loads + placing it in LDS",
           14: "  series (windows): run tables + issuing the next window's loads",
           15: "  series (windows): lag products (+ indicator ballots, first sweep)"}


NAMES_Q = {13: "  ranking: partition pass (two-pass form only)", 14: "  ranking: round load + local histogram", 15: "  ranking: round prefix",
           7: "  ranking: round output (scores / moments)", 11: "  ranking: round placement",
           12: "  ranking: round scores: table lookup, back to partition order",
           0: "  ranking: round count loop (+ sorted value store)", 6: "  ranking: round sorted tags (+ outputs of the kernel)"}


NAMES_F = {13: "  fold merge: tile boundaries (one-chunk path: median place, tables, score loads)",
           14: "  fold merge: staging: loads, fold, LDS stores (one-chunk path: diagonal search)",
           15: "  fold merge: diagonal search + merge (one-chunk path: merge + moment atomics)",
           7: "  fold merge: moment atomics (one-chunk path: members with equals, sums -> moments)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f", action="store_true", help="the library was built with -DMDT_PROF_F: the z-series sub-slots are lent to fold_merge_phase")
    ap.add_argument("--q", action="store_true", help="the library was built with -DMDT_PROF_Q: the z-series sub-slots are lent to the rank phase")
    ap.add_argument("--params", type=int, default=2560)
    ap.add_argument("--draws", type=int, default=10000)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cfg5", action="store_true", help="BASELINE config 5's call: rhat_nested(kind = rank) on Float32, 16 chains in 4 superchains")
    a = ap.parse_args()
    pkg = g.load_package()
    if a.cfg5:
        import numpy as np
        from bench import synth_cfg5
        a.chains = 16
        x = synth_cfg5(a.draws, 16, a.params, torch.device("cuda", 0), 55).permute(2, 1, 0)
        ids = np.repeat(np.arange(1, 5), 4)
        call = lambda **kw: pkg.rhat_nested(x, ids, kind="rank")  # (synchronous: the entry point has no sync switch)
    else:
        x = synth(a.draws, a.chains, a.params, torch.device("cuda", 0), 1234).permute(2, 1, 0)
        call = lambda **kw: pkg.bulk_tail(x, **kw)
    call()
    lib, h = pkg._lib.load(), pkg._lib.ctx(0, torch.cuda.current_stream().cuda_stream)
    buf = (ctypes.c_uint64 * 16)()
    lib.mdt_debug_phase_cycles(h, buf, 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(a.reps):
        call(sync=False)
    ev1.record()
    torch.cuda.synchronize()
    rc = lib.mdt_debug_phase_cycles(h, buf, 1)
    assert rc == 0, rc
    ms = ev0.elapsed_time(ev1) / a.reps
    top = sum(buf[i] for i in range(16))  # every stamp advances the shared clock: the slots partition the time
    print(f"{a.draws}x{a.chains}x{a.params}: {ms:.3f} ms per call, {a.params / ms * 1e3:.0f} param-series/s")
    # which series phase serves the shape: a split chain (draws / 2) + its halo fits the LDS buffer of 6400 values, or windows
    series = NAMES_W if a.draws // 2 + 41 > 6400 else NAMES_Z
    names = {**NAMES, **NAMES_Q} if a.q else ({**NAMES, **NAMES_F} if a.f else {**NAMES, **series})
    for i in sorted(names):
        print(f"{100.0 * buf[i] / top:6.2f} %  {names[i]}")
    other = sum(buf[i] for i in range(16) if i not in names)
    if other:
        print(f"{100.0 * other / top:6.2f} %  (slots without a stamp in this build)")


if __name__ == "__main__":
    main()
