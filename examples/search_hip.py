#!/usr/bin/env python3
"""Counterpart of the reference's `examples/search/faiss.py` (/root/reference/examples/search/faiss.py:18-58) on the HIP engine.

Same steps, same CLI fields (`--dataset_size`, `--vector_size`, `--batch_size`, `--top_k`, `--n_trials`; the reference parses
them with its `arguantic` helper), same client calls:

    vectors -> index build (here: the vector store the server maps; no faiss file)  ->  master spawns the server  ->
    client.search(vector=..., top_k=3)  ->  timed loop of `n_trials` searches with fresh random queries, ms/batch.

    python examples/search_hip.py --dataset_size 100000 --vector_size 384 --batch_size 32 --top_k 10     # BASELINE configs[0]
"""
from __future__ import annotations

import argparse
import pathlib
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vod_amd import factory  # noqa: E402


def parse_args() -> argparse.Namespace:
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--dataset_size", type=int, default=10_000)
    p.add_argument("--vector_size", type=int, default=128)
    p.add_argument("--batch_size", type=int, default=10)
    p.add_argument("--top_k", type=int, default=100)
    p.add_argument("--n_trials", type=int, default=10)
    # extensions of this engine (absent in the reference's script)
    p.add_argument("--devices", type=str, default=None, help="comma-separated GPU ids: row-shard the store over them behind one address")
    p.add_argument("--binary", action="store_true", help="use the raw-bytes route instead of the reference's base64-in-JSON")
    return p.parse_args()


def run(args: argparse.Namespace) -> None:
    vectors = np.random.randn(args.dataset_size, args.vector_size).astype("float32")
    with tempfile.TemporaryDirectory() as tmpdir:
        # Build the index: the store file the server streams into HBM (the reference builds, writes and re-reads a faiss file here)
        devices = None if args.devices is None else [int(d) for d in args.devices.split(",")]
        master = factory.build_hip_mips_index(vectors, config={"logging_level": "warning"}, cache_dir=tmpdir, devices=devices)
        # Spin up the server
        with master:
            client = master.get_client()
            if args.binary:
                client = type(client)(host=client.host, port=client.port, binary=True)
            print(client)
            query_vecs = np.random.randn(args.batch_size, args.vector_size).astype("float32")
            results = client.search(vector=query_vecs, top_k=3)
            print({"search_results": results})
            # Benchmark
            print("Benchmarking...")
            start = time.perf_counter()
            for _ in range(args.n_trials):
                query_vecs = np.random.randn(args.batch_size, args.vector_size).astype("float32")
                results = client.search(vector=query_vecs, top_k=args.top_k)
            end = time.perf_counter()
            assert results.indices.shape == (args.batch_size, args.top_k)
            print(f"HIP MIPS: {1000 * (end - start) / args.n_trials:.3f} ms/batch")


if __name__ == "__main__":
    a = parse_args()
    print(a)
    run(a)
