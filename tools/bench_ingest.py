#!/usr/bin/env python3
"""Host -> HBM ingest rate of the corpus store (the index is rebuilt every training period:
/root/reference/src/vod_exps/recipes/periodic_training.py:53-96, src/vod_search/faiss_search/build.py:51-81).

Sources: (i) pinned float32 NumPy, (ii) pageable float32 NumPy, (iii) the `.npy` memory-mapped store (float16), (iv) the
reference's zarr hand-off layout (100-row chunks, raw and zlib).  Reports seconds, source GB/s, rows/s and the fraction of the
63 GB/s PCIe Gen5 x16 figure.      python tools/bench_ingest.py [--rows 10000000] [--dim 768] [--out profiles/r03_ingest.json]
"""
import argparse
import json
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PCIE = 63.0  # GB/s, PCIe Gen5 x16 per direction (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--zarr-rows", type=int, default=1_000_000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch

    from vod_amd.index import HipFlatIndex
    from vod_amd.search.server import HipEngine
    from vod_amd.zarr_store import write_zarr_vectors

    out = {"dim": a.dim, "pcie_spec_GBps": PCIE, "host_cpus": len(os.sched_getaffinity(0)), "cases": []}

    def record(name, n_rows, n_bytes, seconds, **extra):
        rec = {"source": name, "rows": n_rows, "source_GB": round(n_bytes / 1e9, 2), "seconds": round(seconds, 3),
               "GBps": round(n_bytes / 1e9 / seconds, 2), "rows_per_s": round(n_rows / seconds), "fraction_of_pcie_spec": round(n_bytes / 1e9 / seconds / PCIE, 3), **extra}
        out["cases"].append(rec)
        print(json.dumps(rec), flush=True)

    def fill(arr):  # cheap non-constant content, chunk by chunk (values do not matter for the rate)
        rng = np.random.default_rng(0)
        blk = rng.standard_normal((65536, arr.shape[1])).astype(arr.dtype)
        for lo in range(0, arr.shape[0], 65536):
            arr[lo : lo + 65536] = blk[: min(65536, arr.shape[0] - lo)]

    rows = a.rows
    # (i) pinned float32
    pinned = None
    while pinned is None and rows >= 500_000:
        try:
            pinned = torch.empty((rows, a.dim), dtype=torch.float32, pin_memory=True)
        except RuntimeError:
            rows //= 2
    x = pinned.numpy()
    fill(x)
    for name, src in (("pinned float32 NumPy", x),):
        ix = HipFlatIndex(a.dim, rows, device=0)
        ix.add(src[: 1 << 16])  # staging buffers, code load
        ix.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.add(src)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        record(name, rows, src.nbytes, dt, in_place_dma=bool(ix.get_stat("last_ingest_pinned_src")))
        ix.close()
    # (ii) pageable float32 (half the rows: the copy is the same code path)
    n2 = rows // 2
    page = np.empty((n2, a.dim), dtype=np.float32)
    fill(page)
    for threads in (1, 4, 8, 16):
        ix = HipFlatIndex(a.dim, n2, device=0)
        ix.set_param("ingest_threads", threads)
        ix.add(page[: 1 << 16])
        ix.reset()
        t0 = time.perf_counter()
        ix.add(page)
        torch.cuda.synchronize()
        record("pageable float32 NumPy", n2, page.nbytes, time.perf_counter() - t0, ingest_threads=threads)
        ix.close()
    del page, pinned, x
    # (iii) .npy memory map (float16 store file, as `vod_amd.store.save_vectors` writes it)
    tmp = pathlib.Path(tempfile.mkdtemp())
    mm = np.lib.format.open_memmap(tmp / "vectors.npy", mode="w+", dtype=np.float16, shape=(n2, a.dim))
    fill(mm)
    mm.flush()
    del mm
    t0 = time.perf_counter()
    eng = HipEngine(str(tmp / "vectors.npy"))
    torch.cuda.synchronize()
    record(".npy memory map (float16, page cache warm)", n2, n2 * a.dim * 2, time.perf_counter() - t0, via="HipEngine")
    eng.index.close()
    os.remove(tmp / "vectors.npy")
    # (iv) zarr v2, 100-row chunks (the reference's tensorstore layout), raw and zlib
    zr = min(a.zarr_rows, rows)
    src = np.random.default_rng(1).standard_normal((zr, a.dim)).astype(np.float32)
    for comp, nrows in ((None, zr), ({"id": "zlib", "level": 1}, zr // 4)):
        path = write_zarr_vectors(tmp / f"z_{'raw' if comp is None else 'zlib'}", src[:nrows], dtype=np.float32, chunk_size=100, compressor=comp)
        for workers in (1, 4, 8, 16):
            from vod_amd.zarr_store import ZarrVectors

            zv = ZarrVectors(path)
            ix = HipFlatIndex(a.dim, nrows, device=0)
            t0 = time.perf_counter()
            for _lo, blk in zv.iter_row_blocks(workers=workers):
                ix.add(blk)
            torch.cuda.synchronize()
            record(f"zarr v2, 100-row chunks, {'raw' if comp is None else 'zlib-1'}", nrows, nrows * a.dim * 4, time.perf_counter() - t0,
                   decode_threads=workers, chunk_files=(nrows + 99) // 100)
            ix.close()
    if a.out:
        (ROOT / a.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
