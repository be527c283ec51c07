"""Read (and write) the reference's on-disk vector format directly: a zarr v2 array on a file kvstore.

The reference's predict loop writes the section vectors into a tensorstore array created by
`TensorStoreFactory.instantiate` (/root/reference/src/vod_tools/ts_factory/ts_factory.py:57-92): driver `zarr`,
file kvstore, metadata `{"dtype": "<f2"|"<f4"|"<f8", "shape": [N, D], "chunks": [100, D], "fill_value": "NaN"}`
(tensorstore fills in its defaults: C order, `.` dimension separator, blosc/lz4 byte-shuffled compressor), next to a
`factory.json` that names the driver and the path.  `build_faiss_index` then reads that array row slice by row
slice, casts to float32, `index.add`s, writes a faiss file and the server re-reads it
(src/vod_search/faiss_search/build.py:51-81, factory.py:153-173, server.py:42) - three passes over N*D*4 bytes.

`ZarrVectors` is the `Sequence[np.ndarray]` the ingest side of this package accepts (`HipEngine`,
`build_hip_mips_index`, `store.save_vectors`): it decodes whole chunks and hands chunk-aligned row slices to
`HipFlatIndex.add`, which rounds to fp16/bf16 on the device - one pass, no intermediate file.

No zarr / tensorstore / numcodecs / blosc module exists in this image, so the (small) format layer is written out:
zarr v2 metadata, chunk files, and the blosc-1 container (header, block starts, split streams, byte shuffle) with the
lz4 / zstd / zlib inner codecs taken from pyarrow / the standard library.  blosclz and snappy inner codecs and
bit-shuffle are refused with a clear error.  PARITY NOTE: there is no tensorstore here to produce a reference file;
`tests/test_host_logic.py` round-trips this reader against the writer below and against a blosc container assembled
in the test from the published blosc-1 frame layout.
"""
from __future__ import annotations

import json
import pathlib
import struct
import typing as typ
import zlib

import numpy as np

_DTYPES = {"<f2": np.float16, "<f4": np.float32, "<f8": np.float64}
_BLOSC_MAX_SPLITS = 16
_BLOSC_MIN_BUFFERSIZE = 128


def _arrow_codec(name: str):
    try:
        import pyarrow as pa
    except Exception as exc:  # pragma: no cover
        raise RuntimeError(f"decoding {name}-compressed chunks needs pyarrow") from exc
    if not pa.Codec.is_available(name):
        raise RuntimeError(f"pyarrow was built without the {name} codec")
    return pa.Codec(name)


def _inner_decompress(fmt: int, data: bytes, size: int) -> bytes:
    """One blosc stream; `fmt` = compressor-format bits of the blosc header flags."""
    if fmt == 1:  # lz4 / lz4hc: raw LZ4 block
        return _arrow_codec("lz4_raw").decompress(data, decompressed_size=size).to_pybytes()
    if fmt == 3:  # zlib
        return zlib.decompress(data)
    if fmt == 4:  # zstd
        return _arrow_codec("zstd").decompress(data, decompressed_size=size).to_pybytes()
    raise NotImplementedError(
        f"blosc inner compressor format {fmt} ({ {0: 'blosclz', 2: 'snappy'}.get(fmt, 'unknown') }) is not supported; "
        "re-encode the array with cname lz4, zstd or zlib, or without a compressor"
    )


def blosc1_decompress(buf: bytes) -> bytes:
    """Decode one blosc-1 frame (the c-blosc 1.x container tensorstore and numcodecs write).

    Layout: 16-byte header `version, versionlz, flags, typesize, nbytes(u32), blocksize(u32), cbytes(u32)`;
    flags bit 0 = byte shuffle, bit 1 = stored uncompressed (memcpy), bit 2 = bit shuffle, bit 4 = blocks are not
    split, bits 5-7 = inner compressor.  Then one i32 start offset per block; a block is `typesize` streams (when
    split) or one, each `i32 compressed size` + payload (size == expected size means stored raw).
    """
    if len(buf) < 16:
        raise ValueError("truncated blosc frame")
    _version, _versionlz, flags, typesize = struct.unpack_from("<BBBB", buf, 0)
    nbytes, blocksize, cbytes = struct.unpack_from("<III", buf, 4)
    if cbytes > len(buf):
        raise ValueError(f"blosc frame says {cbytes} bytes, chunk holds {len(buf)}")
    if flags & 0x2:  # memcpyed
        return bytes(buf[16 : 16 + nbytes])
    if flags & 0x4:
        raise NotImplementedError("blosc bit-shuffle is not supported")
    if nbytes == 0:
        return b""
    shuffle = bool(flags & 0x1) and typesize > 1
    dont_split = bool(flags & 0x10)
    fmt = (flags >> 5) & 0x7
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack_from(f"<{nblocks}i", buf, 16)
    out = bytearray(nbytes)
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = (not dont_split) and typesize <= _BLOSC_MAX_SPLITS and (blocksize // typesize) >= _BLOSC_MIN_BUFFERSIZE and not leftover
        nstreams = typesize if split else 1
        ssize = bsize // nstreams
        pos = bstarts[b]
        parts = []
        for _ in range(nstreams):
            (csize,) = struct.unpack_from("<i", buf, pos)
            pos += 4
            payload = bytes(buf[pos : pos + csize])
            pos += csize
            parts.append(payload if csize == ssize else _inner_decompress(fmt, payload, ssize))
        block = b"".join(parts)
        if len(block) != bsize:
            raise ValueError(f"blosc block {b}: decoded {len(block)} bytes, expected {bsize}")
        if shuffle:
            nel = bsize // typesize
            body = np.frombuffer(block, dtype=np.uint8, count=nel * typesize).reshape(typesize, nel).T
            block = np.ascontiguousarray(body).tobytes() + block[nel * typesize :]
        out[b * blocksize : b * blocksize + bsize] = block
    return bytes(out)


def _decode_chunk(raw: bytes, compressor: dict | None) -> bytes:
    if compressor is None:
        return raw
    cid = compressor.get("id")
    if cid == "blosc":
        return blosc1_decompress(raw)
    if cid == "zlib":
        return zlib.decompress(raw)
    if cid == "gzip":
        return zlib.decompress(raw, 16 + zlib.MAX_WBITS)
    if cid == "zstd":
        return _zstd_unknown_size(raw)
    raise NotImplementedError(f"zarr compressor {cid!r} is not supported")


def _zstd_unknown_size(raw: bytes) -> bytes:
    # the zstd frame header carries the content size; pyarrow wants it passed in
    if len(raw) < 6 or raw[:4] != b"\x28\xb5\x2f\xfd":
        raise ValueError("not a zstd frame")
    fhd = raw[4]
    fcs_flag, single = fhd >> 6, (fhd >> 5) & 1
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[fhd & 3]
    width = (1 if single else 0, 2, 4, 8)[fcs_flag]
    if width == 0:
        raise NotImplementedError("zstd frame without a content size")
    size = int.from_bytes(raw[pos : pos + width], "little") + (256 if width == 2 else 0)
    return _arrow_codec("zstd").decompress(raw, decompressed_size=size).to_pybytes()


class ZarrVectors:
    """A 2-D float zarr v2 array on disk as a read-only `Sequence` of rows (`len`, `[i]`, `[a:b]`, `.shape`, `.dtype`)."""

    def __init__(self, path: str | pathlib.Path):
        self.path = pathlib.Path(path)
        meta_path = self.path / ".zarray"
        if not meta_path.exists():
            raise FileNotFoundError(f"`{self.path}` is not a zarr v2 array (no .zarray)")
        meta = json.loads(meta_path.read_text())
        if meta.get("zarr_format") != 2:
            raise ValueError(f"zarr_format {meta.get('zarr_format')} is not supported (expected 2)")
        if meta.get("order", "C") != "C":
            raise NotImplementedError("only C-order zarr arrays are supported")
        if meta.get("filters"):
            raise NotImplementedError("zarr filters are not supported")
        if meta["dtype"] not in _DTYPES:
            raise ValueError(f"expected a float16/32/64 array, got dtype {meta['dtype']!r}")
        if len(meta["shape"]) != 2 or len(meta["chunks"]) != 2:
            raise ValueError(f"expected a 2-D [N, D] array, got shape {meta['shape']}")
        self.dtype = np.dtype(_DTYPES[meta["dtype"]])
        self.shape = (int(meta["shape"][0]), int(meta["shape"][1]))
        self.chunks = (int(meta["chunks"][0]), int(meta["chunks"][1]))
        self.compressor = meta.get("compressor")
        self.sep = meta.get("dimension_separator", ".")
        fill = meta.get("fill_value")
        self.fill_value = {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf, None: 0.0}.get(fill, fill)

    def __len__(self) -> int:
        return self.shape[0]

    @property
    def ndim(self) -> int:
        return 2

    def _chunk(self, ci: int, cj: int) -> np.ndarray:
        """Decoded chunk (ci, cj) at its full chunk shape (edge chunks are stored padded, as zarr does)."""
        f = self.path / f"{ci}{self.sep}{cj}"
        if not f.exists():
            return np.full(self.chunks, self.fill_value, dtype=self.dtype)
        data = _decode_chunk(f.read_bytes(), self.compressor)
        want = self.chunks[0] * self.chunks[1] * self.dtype.itemsize
        if len(data) != want:
            raise ValueError(f"chunk {f.name}: {len(data)} bytes after decoding, expected {want}")
        return np.frombuffer(data, dtype=self.dtype).reshape(self.chunks)

    def read_rows(self, lo: int, hi: int) -> np.ndarray:
        """Rows [lo, hi) as one C-contiguous array of the stored dtype."""
        n, d = self.shape
        lo, hi = max(0, lo), min(n, hi)
        out = np.empty((max(0, hi - lo), d), dtype=self.dtype)
        if hi <= lo:
            return out
        cr = self.chunks[0]
        self._fill_rows(out, lo, lo // cr, (hi - 1) // cr + 1)
        return out

    def __getitem__(self, item):
        if isinstance(item, slice):
            lo, hi, step = item.indices(self.shape[0])
            if step != 1:
                raise NotImplementedError("only unit-stride row slices are supported")
            return self.read_rows(lo, hi)
        i = int(item)
        if i < 0:
            i += self.shape[0]
        if not 0 <= i < self.shape[0]:
            raise IndexError(i)
        return self.read_rows(i, i + 1)[0]

    def _fill_rows(self, out: np.ndarray, lo: int, ci_lo: int, ci_hi: int) -> None:
        """Decode the chunk rows [ci_lo, ci_hi) straight into `out` (whose first row is store row `lo`)."""
        n, d = self.shape
        cr, cc = self.chunks
        hi = lo + out.shape[0]
        raw_full_width = self.compressor is None and cc >= d and cc == d
        for ci in range(ci_lo, ci_hi):
            r0, r1 = max(lo, ci * cr), min(hi, (ci + 1) * cr)
            if raw_full_width and r0 == ci * cr:
                # an uncompressed full-width chunk IS a run of output rows: read the file into place (no bytes object, no copy)
                try:
                    with open(f"{self.path}/{ci}{self.sep}0", "rb", buffering=0) as f:
                        got = f.readinto(memoryview(out[r0 - lo : r1 - lo]).cast("B"))
                    if got != (r1 - r0) * d * self.dtype.itemsize:
                        raise ValueError(f"chunk {ci}{self.sep}0: short read ({got} bytes)")
                    continue
                except FileNotFoundError:
                    out[r0 - lo : r1 - lo] = self.fill_value
                    continue
            for cj in range((d + cc - 1) // cc):
                c0, c1 = cj * cc, min(d, (cj + 1) * cc)
                out[r0 - lo : r1 - lo, c0:c1] = self._chunk(ci, cj)[r0 - ci * cr : r1 - ci * cr, : c1 - c0]

    def iter_row_blocks(self, rows_per_block: int = 65536, workers: int = 8, prefetch: int = 2, chunks_per_task: int = 32) -> typ.Iterator[tuple[int, np.ndarray]]:
        """(first row, rows) blocks aligned to the chunk grid: every chunk file is decoded exactly once.

        The reference's stores are written with 100-row chunks (ts_factory.py:64-77): 100 k files for a 10 M-row index.  Reading +
        decompressing them (file reads, zlib / zstd / blosc release the GIL) runs on `workers` threads, `prefetch` blocks ahead of
        the block being consumed, each task decoding a run of `chunks_per_task` chunks straight into the block's rows - the
        consumer (the H2D ingest) never waits for a file and nothing is copied twice."""
        import collections
        import concurrent.futures

        n, d = self.shape
        cr = self.chunks[0]
        step = max(cr, rows_per_block // cr * cr)
        blocks = [(lo, min(n, lo + step)) for lo in range(0, n, step)]
        if workers <= 1:
            for lo, hi in blocks:
                out = np.empty((hi - lo, d), dtype=self.dtype)
                self._fill_rows(out, lo, lo // cr, (hi - 1) // cr + 1)
                yield lo, out
            return
        with concurrent.futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="vodhip-zarr") as pool:
            def submit(block):
                lo, hi = block
                out = np.empty((hi - lo, d), dtype=self.dtype)
                c_lo, c_hi = lo // cr, (hi - 1) // cr + 1
                return out, [pool.submit(self._fill_rows, out, lo, c, min(c_hi, c + chunks_per_task)) for c in range(c_lo, c_hi, chunks_per_task)]

            pending: collections.deque = collections.deque()
            nxt = 0
            for lo, _hi in blocks:
                while nxt < len(blocks) and len(pending) <= prefetch:
                    pending.append(submit(blocks[nxt]))
                    nxt += 1
                out, futs = pending.popleft()
                for fut in futs:
                    fut.result()
                yield lo, out


def write_zarr_vectors(path: str | pathlib.Path, vectors, dtype=np.float32, chunk_size: int = 100,
                       compressor: dict | None = None) -> pathlib.Path:
    """Write `vectors` ([N, D]) in the layout `TensorStoreFactory.instantiate` declares (ts_factory.py:57-92).

    `compressor` None stores raw chunks; `{"id": "zlib", "level": 1}` deflates them.  Also writes the `factory.json`
    the reference uses to re-open the store (`TensorStoreFactory.from_path`, ts_factory.py:94-104).
    """
    path = pathlib.Path(path)
    path.mkdir(parents=True, exist_ok=True)
    dt = np.dtype(dtype)
    code = {np.dtype(np.float16): "<f2", np.dtype(np.float32): "<f4", np.dtype(np.float64): "<f8"}[dt]
    n = len(vectors)
    d = int(np.asarray(vectors[0]).shape[-1]) if n else 0
    if compressor is not None and compressor.get("id") != "zlib":
        raise NotImplementedError("the writer stores raw or zlib chunks")
    meta = {"zarr_format": 2, "shape": [n, d], "chunks": [chunk_size, d], "dtype": code, "fill_value": "NaN",
            "order": "C", "filters": None, "compressor": compressor, "dimension_separator": "."}
    (path / ".zarray").write_text(json.dumps(meta, indent=1))
    (path / "factory.json").write_text(json.dumps({
        "driver": "zarr", "kvstore": {"driver": "file", "path": str(path)},
        "metadata": {"dtype": code, "shape": [n, d], "chunks": [chunk_size, d], "fill_value": "NaN"}}, indent=2))
    for ci, lo in enumerate(range(0, n, chunk_size)):
        rows = np.asarray(vectors[lo : lo + chunk_size]).astype(dt, copy=False)
        chunk = np.full((chunk_size, d), np.nan, dtype=dt)
        chunk[: len(rows)] = rows
        raw = chunk.tobytes()
        if compressor is not None:
            raw = zlib.compress(raw, int(compressor.get("level", 1)))
        (path / f"{ci}.0").write_bytes(raw)
    return path
