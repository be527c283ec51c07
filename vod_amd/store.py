"""On-disk form of the corpus vectors handed from the predict/index pipeline to the search server.

The reference persists float32 vectors in a tensorstore/zarr array and then builds, writes and re-reads
a faiss file (/root/reference/src/vod_ops/workflows/predict/compute.py:119-138,
src/vod_search/factory.py:153-173, src/vod_search/faiss_search/server.py:42): three passes over
N*D*4 bytes per rebuild.  Here the hand-off is one raw `.npy` ([N, D], float16 or float32) that the
server memory-maps and streams to HBM in slices; the fp16/bf16 rounding happens on the device.
"""
from __future__ import annotations

import hashlib
import json
import pathlib

import numpy as np


def save_vectors(path: str | pathlib.Path, vectors, dtype=np.float16, chunk: int = 262144) -> pathlib.Path:
    """Write `vectors` ([N, D], any sliceable sequence of rows) as an `.npy` of `dtype`, slice by slice."""
    path = pathlib.Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    n = len(vectors)
    d = int(np.asarray(vectors[0]).shape[-1]) if n else 0
    out = np.lib.format.open_memmap(path, mode="w+", dtype=np.dtype(dtype), shape=(n, d))
    for lo in range(0, n, chunk):
        out[lo : lo + chunk] = np.asarray(vectors[lo : lo + chunk]).astype(dtype, copy=False)
    out.flush()
    del out
    return path


def open_vectors(path: str | pathlib.Path):
    """Memory-map a vector file written by `save_vectors` (or any 2-D float16/float32 `.npy`).

    A directory holding a zarr v2 array (the reference's tensorstore vector store, ts_factory.py:57-92) is opened
    as `zarr_store.ZarrVectors` instead: same `.shape` / row-slice interface, chunks decoded on access.
    """
    path = pathlib.Path(path)
    if path.is_dir() and (path / ".zarray").exists():
        from vod_amd.zarr_store import ZarrVectors

        return ZarrVectors(path)
    if path.is_dir():
        path = path / "vectors.npy"
    arr = np.load(path, mmap_mode="r", allow_pickle=False)
    if arr.ndim != 2:
        raise ValueError(f"expected a 2-D vector file, got shape {arr.shape}")
    if arr.dtype not in (np.float16, np.float32):
        raise ValueError(f"expected float16/float32 vectors, got {arr.dtype}")
    return arr


def fingerprint_vectors(vectors, config: dict | None = None, sample: int = 4096) -> str:
    """Cheap content fingerprint (shape + dtype + a strided row sample + config), used to name cached stores.

    Plays the role of `fingerprint(vectors)-config.fingerprint()` in the reference's index cache path
    (src/vod_search/factory.py:146-154).
    """
    h = hashlib.sha1()
    n = len(vectors)
    first = np.asarray(vectors[0]) if n else np.zeros((0,), dtype=np.float32)
    h.update(json.dumps({"n": n, "d": int(first.shape[-1]) if n else 0, "dtype": str(first.dtype), "config": config or {}},
                        sort_keys=True).encode())
    if n:
        step = max(1, n // sample)
        for i in range(0, n, step):
            h.update(np.ascontiguousarray(np.asarray(vectors[i])).tobytes())
    return h.hexdigest()[:16]
