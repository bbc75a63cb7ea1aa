"""CPU, world_size 2, gloo: the params-axis sharding + final all-gather of the N>1 path.
The compute on each rank is injected (the oracle -- test infrastructure) because the HIP path
needs a GPU; what is under test is partitioning and the collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, P, path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    from oracle import c_oracle as co
    pkg = g.load_package()
    x = np.random.default_rng(42).standard_normal((200, 4, P))  # same full array on every rank
    lo, hi = pkg.shard_bounds(P, rank, world)

    def compute(local, **kw):
        return co.bulk_tail(np.asarray(local))

    outs = pkg.ess_rhat_sharded(x[:, :, lo:hi], P, compute=compute)
    full = co.bulk_tail(x)
    ok = all(np.array_equal(o.numpy(), f) for o, f in zip(outs, full))
    with open(f"{path}.{rank}", "w") as f:
        f.write("ok" if ok else "mismatch")
    dist.destroy_process_group()


@pytest.mark.parametrize("P", [8, 7])
def test_sharded_gather_world2(tmp_path, P):
    port = _free_port()
    path = str(tmp_path / "res")
    mp.spawn(_worker, args=(2, port, P, path), nprocs=2, join=True)
    for r in range(2):
        assert open(f"{path}.{r}").read() == "ok"


def test_one_partition_rule():
    """the C-ABI's block rule (mdt_shard_bounds: what a multi-GPU context cuts by, and what a caller with pre-sharded
    input must follow) is dist.shard_bounds (what the one-process-per-GPU path cuts by): first P mod N blocks one longer.
    Pure arithmetic in the library: runs without a GPU."""
    import ctypes
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    lib = pkg._lib.load()
    for P in (0, 1, 7, 8, 10_000, 1_000_003):
        for n in (1, 2, 3, 8):
            prev = 0
            for i in range(n):
                lo, hi = ctypes.c_int64(), ctypes.c_int64()
                assert lib.mdt_shard_bounds(P, i, n, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == pkg.shard_bounds(P, i, n)
                assert lo.value == prev and hi.value - lo.value in (P // n, P // n + 1)
                prev = hi.value
            assert prev == P
    assert lib.mdt_shard_bounds(5, 2, 2, ctypes.byref(lo), ctypes.byref(hi)) == pkg._lib.MDT_EARG
