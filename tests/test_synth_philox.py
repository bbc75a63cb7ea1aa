"""The synthetic inputs of bench.py: Philox-4x32-10 keyed by (seed, draw, chain, global parameter index).
Known-answer vectors of the generator, the host twin against the torch implementation, and position independence: the
shard of any rank is a slice of the array the single-GPU run sees."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10"""
    z = np.zeros(1, dtype=np.int64)
    assert [int(v[0]) for v in bench.philox4x32(z, z, z, z, 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = np.full(1, 0xFFFFFFFF, dtype=np.int64)
    assert [int(v[0]) for v in bench.philox4x32(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    c = [np.full(1, v, dtype=np.int64) for v in (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)]
    assert [int(v[0]) for v in bench.philox4x32(*c, 0xa4093822, 0x299f31d0)] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_shards_are_slices_and_the_host_twin_agrees():
    import torch
    D, C = 600, 4
    full = bench.synth(D, C, 40, None, 77)
    for p0, cnt in ((0, 16), (16, 8), (24, 16)):
        assert np.array_equal(bench.synth(D, C, cnt, None, 77, p0=p0), full[p0:p0 + cnt])     # position independent
    assert np.array_equal(bench.synth(D, C, 40, None, 77, chunk=7), full)                       # ... and chunk independent
    t = bench.synth(D, C, 40, torch.device("cpu"), 77).numpy()
    assert np.max(np.abs(t - full)) < 1e-13                                                     # torch twin (libm differences only)
    assert not np.array_equal(bench.synth(D, C, 8, None, 78), full[:8])                         # the seed matters
    # half IID, half AR(1) in every block of 8 parameters; lag-1 autocorrelation tells them apart
    x = bench.synth(4000, C, 16, None, 5)
    r1 = np.array([np.mean([np.corrcoef(x[p, c, :-1], x[p, c, 1:])[0, 1] for c in range(C)]) for p in range(16)])
    want = np.array([0, 0, 0, 0, -0.3, 0.3, 0.7, 0.9] * 2)
    assert np.max(np.abs(r1 - want)) < 0.06
    assert np.all(np.abs(x.std(axis=2) - 1.0) < 0.2)
    x5 = bench.synth_cfg5(500, 16, 6, None, 55)
    assert x5.dtype == np.float32 and np.array_equal(bench.synth_cfg5(500, 16, 3, None, 55, p0=3), x5[3:])
    assert np.all(x5[:, 0:4, 0] == x5[:, 0:1, 0]) and np.all(x5[:, 4:8, 0] == x5[:, 4:5, 0])    # superchains share their initial value
