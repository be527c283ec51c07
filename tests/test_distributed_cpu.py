"""world_size-2 gloo test of the row-sharded search exchange (all-gather + merge), on CPU.

The collective logic is the product's (`vod_amd.distributed.ShardedFlatIndex`); the per-rank local search
and the k-way merge are injected test doubles backed by the oracle, because the real ones are HIP kernels.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, nq, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.flat_ip import flat_ip_topk, merge_shard_topk
        from vod_amd.distributed import ShardedFlatIndex, shard_bounds

        rng = np.random.default_rng(5)
        x = rng.integers(-4, 5, size=(n, d)).astype(np.float32)
        q = rng.integers(-4, 5, size=(nq, d)).astype(np.float32)
        bounds = shard_bounds(n, world, align=16)
        lo, hi = bounds[rank], bounds[rank + 1]

        def local_search(queries, kk, base):
            s, i = flat_ip_topk(queries.numpy(), x[lo:hi], kk, id_base=base)
            return torch.from_numpy(s), torch.from_numpy(i)

        def merge(gs, gi):
            s, i = merge_shard_topk(list(gs.numpy()), list(gi.numpy()), gs.shape[-1])
            return torch.from_numpy(s), torch.from_numpy(i)

        index = ShardedFlatIndex(None, lo, local_search=local_search, merge=merge)
        s, i = index.search(torch.from_numpy(q), k)
        rs, ri = flat_ip_topk(q, x, k)
        ok = np.array_equal(i.numpy(), ri) and np.array_equal(s.numpy(), rs)
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([ok, lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(1000, 10), (37, 20)])
def test_sharded_search_two_ranks_gloo(tmp_path, n, k):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, 16, 5, k, str(tmp_path)), nprocs=world, join=True)
    spans = []
    for r in range(world):
        ok, lo, hi = np.load(tmp_path / f"ok_{r}.npy")
        assert ok == 1, f"rank {r}: merged result differs from the global oracle"
        spans.append((lo, hi))
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == n


def test_shard_bounds_are_contiguous_and_aligned():
    from vod_amd.distributed import shard_bounds

    b = shard_bounds(10_000_000, 8, align=250_000)
    assert b[0] == 0 and b[-1] == 10_000_000 and all(x % 250_000 == 0 for x in b)
    assert [b[i + 1] - b[i] for i in range(8)] == [1_250_000] * 8
    b = shard_bounds(1001, 4, align=256)
    assert b == sorted(b) and b[-1] == 1001
