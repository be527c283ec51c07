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


# ---- the multi-GPU server's dispatch (vod_amd.search.group), world_size 2 on gloo ----------------------------------------


def _group_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.flat_ip import flat_ip_topk, merge_shard_topk, topk_desc_tiebreak
        from vod_amd.distributed import ShardedFlatIndex, shard_bounds
        from vod_amd.search.group import GroupDispatcher

        rng = np.random.default_rng(11)
        n, d = 3000, 24
        x = rng.integers(-4, 5, size=(n, d)).astype(np.float32)
        labels = rng.integers(0, 5, size=n).astype(np.int32)
        bounds = shard_bounds(n, world, align=256)
        lo, hi = bounds[rank], bounds[rank + 1]

        def local_search(queries, kk, base, subset=None):
            if kk == 1999:  # stands for an argument error the library raises on EVERY rank before its collective
                raise ValueError("injected: every rank refuses k = 1999")
            q = queries.numpy()
            if subset is None:
                s, i = flat_ip_topk(q, x[lo:hi], kk, id_base=base)
            else:  # the shard's own labels decide eligibility: what the HIP index does with its slice of the row labels
                full = q.astype(np.float64) @ x[lo:hi].astype(np.float64).T
                sub = subset.numpy()
                for r in range(len(q)):
                    allowed = sub[r][sub[r] >= 0]
                    if allowed.size:
                        full[r, ~np.isin(labels[lo:hi], allowed)] = np.nan
                s, i = topk_desc_tiebreak(full, kk, id_base=base)
            return torch.from_numpy(s), torch.from_numpy(i)

        def merge(gs, gi):
            s, i = merge_shard_topk(list(gs.numpy()), list(gi.numpy()), gs.shape[-1])
            return torch.from_numpy(s), torch.from_numpy(i)

        disp = GroupDispatcher(ShardedFlatIndex(None, lo, local_search=local_search, merge=merge), rank, world, torch.device("cpu"), dim=d)
        if rank != 0:
            served = disp.worker_loop()
            np.save(os.path.join(out_dir, f"served_{rank}.npy"), np.array([served, disp.errors]))
            return
        ok = True
        # requests the library would refuse are stopped on rank 0 BEFORE any broadcast (round-2 advisor: one such request
        # killed every worker of the group server); the workers never see them and the next search is served normally
        q_ok = rng.integers(-4, 5, size=(3, d)).astype(np.float32)
        refused = 0
        for bad_q, bad_k, bad_sub in [(q_ok, 0, None), (q_ok, 5000, None), (q_ok, -3, None), (q_ok[0], 5, None), (q_ok[:, :5], 5, None),
                                      (q_ok, 5, np.zeros((3, 65), np.int32)), (q_ok, 5, np.zeros((2, 4), np.int32))]:
            try:
                disp.search(bad_q, bad_k, subset=bad_sub)
            except ValueError:
                refused += 1
        ok = ok and refused == 7
        es, ei = disp.search(q_ok[:0], 4)
        ok = ok and es.shape == (0, 4) and ei.shape == (0, 4)
        # an error that every rank raises inside its local search: rank 0 reports it, the workers log it and keep serving
        try:
            disp.search(q_ok, 1999)
            ok = False
        except ValueError:
            pass
        for nq, k, with_subset in [(7, 10, False), (1, 3, False), (33, 50, True), (5, 2000, False)]:
            q = rng.integers(-4, 5, size=(nq, d)).astype(np.float32)
            sub = None
            full = q.astype(np.float64) @ x.astype(np.float64).T
            if with_subset:
                sub = np.full((nq, 2), -1, dtype=np.int32)
                sub[::2, 0] = 3
                sub[1::4] = [1, 4]
                for r in range(nq):
                    allowed = sub[r][sub[r] >= 0]
                    if allowed.size:
                        full[r, ~np.isin(labels, allowed)] = np.nan
            s, i = disp.search(q, k, subset=sub)
            rs, ri = topk_desc_tiebreak(full, k)
            ok = ok and np.array_equal(i, ri) and np.array_equal(s, rs)
        disp.stop()
        np.save(os.path.join(out_dir, "ok_0.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()


def test_group_dispatcher_two_ranks_gloo(tmp_path):
    """Rank 0 drives, rank 1 serves: header / queries / subset labels are broadcast, both ranks run the same sharded
    search, rank 0's merged answers equal the oracle on the whole store, the stop word ends rank 1's loop."""
    mp.spawn(_group_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "ok_0.npy")[0] == 1
    assert np.load(tmp_path / "served_1.npy").tolist() == [5, 1]  # 4 good searches + the one every rank refused; 7 bad ones never left rank 0


def test_group_dispatcher_eight_ranks_gloo(tmp_path):
    """The same dispatch at the node's full width: 8 ranks (375 rows each), rank 0 drives, 7 workers serve."""
    mp.spawn(_group_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert np.load(tmp_path / "ok_0.npy")[0] == 1
    for r in range(1, 8):
        assert np.load(tmp_path / f"served_{r}.npy").tolist() == [5, 1]


def test_bench_launcher_eight_ranks():
    """`python bench.py --gpus 8 --launch-check`: the self-launcher at the driver's widest configuration (8 fresh children,
    gloo rendezvous on 127.0.0.1, only rank 0 reports)."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--launch-check"], capture_output=True, text=True,
                         timeout=600, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": "ok", "world": 8, "rank_sum": 28.0}


def _one_sided_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vod_amd.distributed import ShardedFlatIndex
    from vod_amd.search.group import GroupDispatcher

    def local_search(queries, kk, base, subset=None):
        if rank == 1:  # stands for a HIP out-of-memory / launch failure on ONE shard
            raise RuntimeError("injected: hipErrorOutOfMemory on this shard only")
        return torch.zeros((len(queries), kk)), torch.zeros((len(queries), kk), dtype=torch.int64)

    disp = GroupDispatcher(ShardedFlatIndex(None, 0, local_search=local_search, merge=lambda s, i: (s[0], i[0])), rank, world,
                           torch.device("cpu"), dim=4)
    if rank != 0:
        disp.worker_loop()  # must not return: the one-sided failure ends the process with a non-zero code
        os._exit(0)
    disp.search(np.zeros((2, 4), np.float32), 3)  # blocks in the all-gather rank 1 never joins - until the test kills it
    os._exit(0)


def test_group_worker_dies_on_a_one_sided_failure():
    """Round-3 advisor: a worker that swallows an error only IT raised skips the all-gather the other ranks are blocked in, and every
    later request hangs.  Such a failure must end the worker non-zero (the owner process then tears the group down and reports it);
    errors every rank raises alike before the collective are still survived (`test_group_dispatcher_two_ranks_gloo`)."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_one_sided_worker, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    procs[1].join(timeout=120)
    try:
        assert procs[1].exitcode == 3, f"the failing worker left with {procs[1].exitcode}"
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
            p.join(timeout=30)
