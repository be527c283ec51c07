#!/usr/bin/env python3
"""Benchmark of the north-star metric: queries/s of exact brute-force inner-product top-k.

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns one fresh process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the same under an external launcher

Workload (BASELINE.json, configs[2], the configuration `metric` is quoted on; it fits one GPU):
10 M sections x 768 fp16, batch = 1024 queries, top-100, corpus resident in HBM, row-sharded over the N
ranks (strong scaling: the corpus is fixed, each rank holds N_total / N rows).  One "step" = one batch of
1024 queries answered end to end: local fused score + top-k on every rank, RCCL all-gather of the
per-shard top-k, merge.  Inputs are synthetic N(0,1) embeddings generated on the device (corpus seed 1234,
query seed 4321); queries are resident in HBM when the timed region starts.  `--data clustered` sorts the
rows by topic cluster and draws the queries from the LAST clusters (documents ingested in topic order:
the row order a real corpus has, /root/reference/src/vod_search/faiss_search/build.py:65-73).

Prints ONE JSON line on rank 0 (see the keys below); `roofline` is for the dominant kernel
(`mips_filter16p_kernel`: MFMA-bound above ~312 queries per batch, HBM-bound below), `cpu_baseline` is the
oracle-side faiss-CPU restatement timed on the host cores of this box (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GEN_CHUNK = 250_000  # rows per generation chunk; shard boundaries are multiples of it so the corpus is the same for every N
RIDGE_NQ = 312       # 2.5 PFLOP/s / 8 TB/s: batches below this many queries are HBM-bound


def parse_args() -> argparse.Namespace:
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--nq", type=int, default=1024)
    p.add_argument("--k", type=int, default=100)
    p.add_argument("--dtype", choices=["f16", "bf16"], default="f16")
    p.add_argument("--data", choices=["iid", "clustered"], default="iid")
    p.add_argument("--tile", type=int, default=0)
    p.add_argument("--growth", type=int, default=0, help="stage growth factor x100 (0 = library default)")
    p.add_argument("--force-collective", action="store_true",
                   help="run the multi-GPU step (RCCL all-gather of the packed top-k + merge) even with one rank: exercises the N > 1 code on a 1-GPU box")
    p.add_argument("--param", action="append", default=[], metavar="KEY=VALUE", help="library tunable (vodhip_index_set_param), repeatable")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--verify-queries", type=int, default=64)
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--launch-check", action="store_true",
                   help="CPU-only check of the launcher: every rank joins a gloo group, all-reduces its rank, rank 0 prints one JSON line")
    return p.parse_args()


def _free_rendezvous_port(socket) -> int:
    """A free port BELOW the kernel's ephemeral range: ranks 1..N-1 retry the rendezvous while rank 0 is still importing
    torch, and a client retrying a not-yet-listening localhost port inside that range can be given the port itself as
    its source (TCP self-connect), after which rank 0's listen fails with EADDRINUSE."""
    import random

    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            lo = int(f.read().split()[0])
    except (OSError, ValueError):
        lo = 32768
    rng = random.SystemRandom()
    for _ in range(64):
        port = rng.randrange(20000, max(20001, min(lo, 32768)))
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh children (one per GPU) and wait for them.

    The parent never touches the GPU (it does not even import torch): a process that has initialised HIP must not be
    replaced or forked into ranks.  Rank 0 inherits stdout, so its JSON line is this command's output."""
    import socket
    import subprocess

    port = _free_rendezvous_port(socket)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(pathlib.Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for other in alive:  # a rank died: the others would wait in a collective for ever
                    other.terminate()
    return rc


def launch_check(rank: int, world: int) -> None:
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": "ok", "world": world, "rank_sum": float(t.item())}), flush=True)
    dist.destroy_process_group()


def make_rows(torch, dev, tdt, data: str, chunk: int, rows: int, d: int, n_total: int):
    """Rows [chunk * GEN_CHUNK, +rows) of the synthetic corpus (identical for every rank count)."""
    g = torch.Generator(device=dev).manual_seed(1234 + chunk)
    x = torch.randn((rows, d), generator=g, device=dev, dtype=torch.float32)
    if data == "clustered":
        n_clusters = max(16, n_total // 5000)
        centers = cluster_centers(torch, dev, n_clusters, d)
        ridx = torch.arange(chunk * GEN_CHUNK, chunk * GEN_CHUNK + rows, device=dev, dtype=torch.int64)
        x = 0.6 * x + 0.8 * centers[(ridx * n_clusters) // n_total]  # unit variance, sorted by cluster
    return x.to(tdt)


def cluster_centers(torch, dev, n_clusters: int, d: int):
    g = torch.Generator(device=dev).manual_seed(99)
    return torch.randn((n_clusters, d), generator=g, device=dev, dtype=torch.float32)


def make_queries(torch, dev, tdt, data: str, nq: int, d: int, n_total: int):
    g = torch.Generator(device=dev).manual_seed(4321)
    q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
    if data == "clustered":  # every query looks for the topics at the END of the store
        n_clusters = max(16, n_total // 5000)
        centers = cluster_centers(torch, dev, n_clusters, d)
        late = n_clusters - 1 - torch.randint(0, max(1, n_clusters // 10), (nq,), generator=g, device=dev)
        q = 0.6 * q + 0.8 * centers[late]
    return q.to(tdt)


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool (before HIP starts)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_check:
        return launch_check(rank, world)
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or args.force_collective  # the exchange step runs (with one rank it gathers from itself)
    if rank != 0:  # only rank 0 reports: keep the other ranks' library banners out of the launcher's merged stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if multi:
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from vod_amd.index import HipFlatIndex, PackedTopk, merge_topk

    tdt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    n_total, d, nq, k = args.rows, args.dim, args.nq, args.k
    # contiguous row sharding on generation-chunk boundaries
    n_chunks = (n_total + GEN_CHUNK - 1) // GEN_CHUNK
    c_lo = (n_chunks * rank) // world
    c_hi = (n_chunks * (rank + 1)) // world
    row_lo = min(n_total, c_lo * GEN_CHUNK)
    row_hi = min(n_total, c_hi * GEN_CHUNK)
    n_local = row_hi - row_lo

    index = HipFlatIndex(d, max(n_local, 1), dtype=tdt, device=local_rank)
    if args.tile:
        index.set_param("tile", args.tile)
    if args.growth:
        index.set_param("growth", args.growth)
    for kv in args.param:
        key, _, val = kv.partition("=")
        index.set_param(key, int(val))
    t_build0 = time.perf_counter()
    for c in range(c_lo, c_hi):
        index.add(make_rows(torch, dev, tdt, args.data, c, min(GEN_CHUNK, n_total - c * GEN_CHUNK), d, n_total))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build0
    assert index.ntotal == n_local
    queries = make_queries(torch, dev, tdt, args.data, nq, d, n_total)

    packed = PackedTopk(nq, k, dev)  # [scores | ids] record of this rank: the exchange is ONE all-gather of 12*nq*k bytes
    out_s, out_i = packed.scores, packed.ids
    if multi:
        gathered = torch.empty((world * packed.nbytes,), dtype=torch.uint8, device=dev)

    # One step = one batch through the hot path.  The host runs ONE step ahead of the device: step i+1 is enqueued
    # before step i's exactness flag is checked (`finish` waits for that search alone), so the device never idles
    # between batches.  Every step's check (and recovery, if a candidate list overflowed) happens inside the timed
    # region; a recovered step re-sends its (now complete) local result through the exchange.
    # (Overlapping the all-gather of batch i with the search of batch i+1 on RCCL's stream was measured and dropped: the
    # persistent filter kernel owns every CU, the collective's workgroups wait for one anyway and slow the stage kernels
    # by 11 %: 1.936 ms per batch against 1.929 for this plain sequence on a 1.25 M-row shard.)
    state = {"in_flight": 0, "ns": 0, "launches": 0, "recovery_passes": 0, "res": None}

    def exchange():
        dist.all_gather_into_tensor(gathered, packed.buffer)
        return packed.merge_gathered(gathered, world)

    def step():
        index.search_async(queries, k, id_base=row_lo, out=(out_s, out_i))
        state["in_flight"] += 1
        state["res"] = exchange() if multi else (out_s, out_i)
        while state["in_flight"] > 1:
            finish_one()

    def finish_one():
        index.finish()
        state["in_flight"] -= 1
        state["ns"] += index.get_stat("last_filter_ns")
        state["launches"] += index.get_stat("last_filter_launches")
        passes = index.get_stat("last_safe_reruns")
        if passes:
            state["recovery_passes"] += passes
            if multi:  # every step searches the same queries: the recovered local result replaces the exchanged one
                state["res"] = exchange()

    def drain():
        while state["in_flight"]:
            finish_one()

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    index.set_param("profile", 1)
    state["ns"] = state["launches"] = state["recovery_passes"] = 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    filter_ns, filter_launches = state["ns"], state["launches"]
    index.set_param("profile", 0)
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- post-run verification (outside the timed region): exactness on a query sample ----
    verify = None
    if not args.no_verify:
        fs, fi = state["res"]
        n_v = min(nq, max(1, args.verify_queries))
        sample = [int(round(j * (nq - 1) / max(1, n_v - 1))) for j in range(n_v)] if n_v > 1 else [0]
        sample = sorted(set(sample))
        # local brute force on this rank's shard with torch (fp32 matmul on the stored rows), merged over ranks
        kk = min(k, n_local)
        ls = torch.full((len(sample), kk), float("-inf"), device=dev)
        li = torch.full((len(sample), kk), -1, dtype=torch.int64, device=dev)
        qs = queries[sample].float()
        for lo in range(0, n_local, 1_000_000):
            blk = index.stored_rows(lo, min(1_000_000, n_local - lo)).float()
            ts, ti = torch.topk(qs @ blk.T, min(kk, blk.shape[0]), dim=1)
            cs, ci = torch.cat([ls, ts], dim=1), torch.cat([li, ti + (lo + row_lo)], dim=1)
            top = torch.topk(cs, kk, dim=1)
            ls, li = top.values, torch.gather(ci, 1, top.indices)
            del blk
        if multi:
            pad_s = torch.full((len(sample), k), float("-inf"), device=dev)
            pad_i = torch.full((len(sample), k), -1, dtype=torch.int64, device=dev)
            pad_s[:, : ls.shape[1]] = ls
            pad_i[:, : li.shape[1]] = li
            as_ = torch.empty((world * len(sample), k), device=dev)
            ai_ = torch.empty((world * len(sample), k), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(as_, pad_s)
            dist.all_gather_into_tensor(ai_, pad_i)
            ls, li = merge_topk(as_.view(world, len(sample), k), ai_.view(world, len(sample), k))
        got_i = fi[sample].cpu()
        ref_i = li.cpu()
        hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(got_i, ref_i))
        verify = {
            "recall_at_k_vs_torch_fp32": hits / float(ref_i.numel()),
            "max_abs_score_diff": float((fs[sample][:, : ls.shape[1]].cpu() - ls.cpu()).abs().max()),
            "queries_checked": len(sample),
        }

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = nq * args.steps / elapsed
        flops_per_step = 2.0 * nq * n_local * d          # algorithmic flops of this rank's filter launches per step
        bytes_per_step = n_local * d * 2.0 + nq * d * 2.0 + nq * k * 12.0
        filt_s = filter_ns * 1e-9
        mfma_bound = nq >= RIDGE_NQ
        if filt_s > 0:
            achieved = flops_per_step * args.steps / filt_s / 1e12 if mfma_bound else bytes_per_step * args.steps / filt_s / 1e9
        else:
            achieved = None
        peak = 2500.0 if mfma_bound else 8000.0
        traffic, traffic_src = None, None
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():  # HBM bytes per step from this round's rocprofv3 --pmc passes of the same command (tools/pmc.sh)
            try:
                ent = json.loads(tfile.read_text()).get(f"{n_total}x{d}x{nq}@{world}" + ("" if args.data == "iid" else "/clustered"))
                if isinstance(ent, dict):
                    traffic, traffic_src = ent.get("bytes"), ent.get("source")
                else:
                    traffic = ent
            except Exception:
                traffic = None

        def _m(v: int) -> str:
            return f"{v // 1_000_000}M" if v % 1_000_000 == 0 else (f"{v / 1e6:g}M" if v >= 1_000_000 else str(v))

        line = {
            "metric": f"queries/sec brute-force top-k ({_m(n_total)}x{d} {'fp16' if args.dtype == 'f16' else 'bf16'})",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic" if args.data == "iid" else "synthetic, rows sorted by topic cluster, queries from the last clusters",
            "config": {
                "workload": f"{n_total} sections x {d} {args.dtype}, batch {nq} queries, top-{k}, exact brute force",
                "rows_per_gpu": n_local,
                "parallelism": f"row-sharded x{world} + RCCL all-gather of per-shard top-k" if multi else "single GPU",
                "index_build_s": round(t_build, 3),
                "recovery_passes": state["recovery_passes"],
            },
            "roofline": {
                "bound": "mfma" if mfma_bound else "hbm",
                "kernel": "mips_filter16p_kernel" if (args.tile in (0, 8, 9) and nq > 128) else f"mips_filter_kernel[tile={args.tile}]",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": (achieved / peak) if achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches_per_step": filter_launches / args.steps,
                "kernel_ms_per_step": filt_s / args.steps * 1e3,
                "algorithmic_flops_per_step": flops_per_step,
                "algorithmic_bytes_per_step": bytes_per_step,
                "mfma_frac_of_2.5PF": (flops_per_step * args.steps / filt_s / 2.5e15) if filt_s > 0 else None,
                "hbm_frac_at_8TBps": (bytes_per_step * args.steps / filt_s / 8e12) if filt_s > 0 else None,
            },
        }
        if verify is not None:
            line["verify"] = verify
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_baseline import time_cpu_baseline  # the reported CPU baseline, never the product path

            line["cpu_baseline"] = time_cpu_baseline(d, nq, k, n_total, target_seconds=args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = qps / line["cpu_baseline"]["value"]
    index.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio (block-buffered when stdout is a pipe): flush it first so that
        # the JSON line is the LAST line of rank 0's stdout
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
