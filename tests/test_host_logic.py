"""CPU tests: host-side mirror of the reference interface (containers, codec, routing, server contract),
and the C-ABI library's exported symbols.  No compute call touches a GPU here."""
import ctypes
import json
import pickle
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

from vod_amd import io as vio
from vod_amd import types as vt
from vod_amd.search import base, client as vclient, sharded


def wall_clock_test(attempts: int = 3):
    """Tests of the batcher's timing rules assert wall-clock bounds (a scan is a 40-150 ms sleep, a grace wait 24 ms).  The bounds have slack,
    but a loaded box can stall a thread for longer than any slack: such a test passes if ONE of a few whole attempts meets its bounds
    (a rule that is broken misses them every time)."""
    import functools

    def wrap(fn):
        @functools.wraps(fn)
        def run(*a, **kw):
            for i in range(attempts):
                try:
                    return fn(*a, **kw)
                except AssertionError:
                    if i + 1 == attempts:
                        raise
        return run
    return wrap


def _load(name):
    return np.load(GOLDEN / f"{name}.npz")


# ---- C-ABI ------------------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    from vod_amd import _native
    from vod_amd.build import build_native

    build_native()
    lib = _native.load_library()
    header = (ROOT / "include" / "vodhip.h").read_text()
    declared = set(re.findall(r"\b(vodhip_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/vodhip.h but not exported"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    assert lib.vodhip_version() == 1


def test_native_errors_are_reported_not_swallowed():
    import ctypes

    from vod_amd import _native

    lib = _native.load_library()
    out = ctypes.c_int64()
    assert lib.vodhip_index_ntotal(None, ctypes.byref(out)) != 0
    with pytest.raises(_native.NativeLibraryError, match="NULL"):
        _native.check(lib.vodhip_index_ntotal(None, ctypes.byref(out)))


def test_product_has_no_cpu_fallback_and_never_imports_the_oracle():
    for path in (ROOT / "vod_amd").rglob("*.py"):
        text = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{path} imports the oracle"
    # the measurement / profiling helpers are not test infrastructure either: only tests/, smoke() and bench.py's CPU baseline use it
    for path in list((ROOT / "tools").rglob("*.py")) + list((ROOT / "tools").rglob("*.sh")):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", path.read_text(), flags=re.M), f"{path} imports the oracle"
    import torch

    if not torch.cuda.is_available():
        from vod_amd import _native
        from vod_amd.index import HipFlatIndex

        with pytest.raises(_native.NativeLibraryError):
            HipFlatIndex(8, 8)


# ---- containers -------------------------------------------------------------------------------------
def test_stack_samples_matches_reference_padding():
    g = _load("stack_samples_ragged")
    rows = [
        vt.RetrievalSample(scores=g["r0_s"], indices=g["r0_i"]),
        vt.RetrievalSample(scores=g["r1_s"], indices=g["r1_i"]),
        vt.RetrievalSample(scores=np.array([], dtype=np.float32), indices=np.array([], dtype=np.int64)),
    ]
    out = vt.RetrievalBatch.stack_samples(rows)
    np.testing.assert_array_equal(out.scores, g["out_scr"])
    np.testing.assert_array_equal(out.indices, g["out_idx"])
    assert out.scores.dtype == np.float32 and out.indices.dtype == np.int64


def test_retrieval_batch_contract():
    b = vt.RetrievalBatch.cast(scores=[[1.0, 3.0], [2.0, -np.inf]], indices=[[5, 6], [7, -1]])
    assert b.shape == (2, 2) and len(b) == 2
    assert isinstance(b[0], vt.RetrievalSample) and isinstance(b[0][1], vt.RetrievalTuple)
    s = b.sorted()
    assert s.indices.tolist() == [[6, 5], [7, -1]]
    w = b * 0.0
    assert np.isnan(w.scores[1, 1]) and w.scores[0, 0] == 0.0  # 0 * -inf -> NaN (reference quirk Q5)
    both = b + b
    assert both.shape == (4, 2)
    with pytest.raises(ValueError):
        vt.RetrievalBatch(scores=np.zeros((2, 2)), indices=np.zeros((2, 3), dtype=np.int64))
    with pytest.raises(ValueError):
        vt.RetrievalBatch(scores=np.zeros((2,)), indices=np.zeros((2,), dtype=np.int64))
    with pytest.raises(TypeError):
        b * "x"  # noqa: B018
    assert b.to_dict()["indices"] == [[5, 6], [7, -1]]


# ---- wire codec -------------------------------------------------------------------------------------
def test_codec_is_string_identical_to_the_reference():
    g = _load("io_codec")
    strings = json.loads((GOLDEN / "io_codec.json").read_text())
    for key in ("f32", "i64", "f16"):
        assert vio.serialize_np_array(g[key]) == strings[key]
        back = vio.deserialize_np_array(strings[key])
        np.testing.assert_array_equal(back, g[key])
        assert back.dtype == g[key].dtype


@pytest.mark.parametrize("native", [True, False])
def test_codec_fast_paths_keep_the_reference_bytes(native, monkeypatch):
    """The hand-written .npy header + base64 loops (libvodhip's host helper, or binascii without the library) must
    emit exactly `urlsafe_b64encode(np.save(...))` (src/vod_search/io.py:17-22) for every layout and length mod 3."""
    import base64
    import io

    if not native:
        monkeypatch.setattr(vio, "_lib_state", [None])
    else:
        monkeypatch.setattr(vio, "_lib_state", [])
        assert vio._codec_lib() is not None

    def ref(a):
        buf = io.BytesIO()
        np.save(buf, np.asarray(a), allow_pickle=True)
        return base64.urlsafe_b64encode(buf.getvalue()).decode("utf-8")

    rng = np.random.default_rng(0)
    cases = [rng.standard_normal((33, 70)).astype(np.float32), np.arange(12, dtype=np.int64).reshape(3, 4),
             np.zeros((0, 5), np.float32), np.float32(3.0), np.asfortranarray(rng.standard_normal((4, 5))),
             np.arange(24).reshape(2, 3, 4)[:, ::2], np.array([-np.inf, np.nan, 1.0], dtype=np.float32)]
    cases += [rng.integers(0, 255, size=(n,), dtype=np.uint8) for n in range(0, 13)]
    for a in cases:
        enc = vio.serialize_np_array(a)
        assert enc == ref(a)
        back = vio.deserialize_np_array(enc)
        assert back.dtype == np.asarray(a).dtype and back.shape == np.asarray(a).shape and back.flags.writeable
        np.testing.assert_array_equal(back, a)
        np.testing.assert_array_equal(vio.deserialize_np_array(enc.replace("-", "+").replace("_", "/")), a)  # std alphabet
    big = vio.serialize_np_array(cases[0])
    np.testing.assert_array_equal(vio.deserialize_np_array(big[:100] + "\n" + big[100:]), cases[0])  # lenient, like b64decode
    # the JSON helpers produce / accept ordinary JSON
    body = vio.json_body({"vectors": big}, {"top_k": 7, "subset_ids": [["a"], []]})
    assert json.loads(body) == {"vectors": big, "top_k": 7, "subset_ids": [["a"], []]}
    assert vio.parse_json_body(body, ("vectors",)) == json.loads(body)
    assert vio.parse_json_body(json.dumps({"top_k": 1, "vectors": big}).encode(), ("vectors",)) == {"top_k": 1, "vectors": big}
    assert vio.parse_json_body(b'{"top_k": 2}', ("vectors",)) == {"top_k": 2}
    with pytest.raises(ValueError):
        vio.parse_json_body(b'[1, 2]', ("vectors",))


def test_codec_refuses_pickled_payloads():
    import base64
    import io

    buf = io.BytesIO()
    np.save(buf, np.array([{"a": 1}], dtype=object), allow_pickle=True)
    with pytest.raises(ValueError):
        vio.deserialize_np_array(base64.urlsafe_b64encode(buf.getvalue()).decode())


# ---- logical shard routing --------------------------------------------------------------------------
class _TableClient(base.SearchClient):
    def __init__(self, scr, idx):
        self.scr, self.idx = scr, idx

    def ping(self):
        return True

    def search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k=3):  # noqa: ARG002
        n = len(text)
        return vt.RetrievalBatch(scores=self.scr[:n, :top_k].copy(), indices=self.idx[:n, :top_k].copy())


def test_sharded_scatter_gather_matches_reference_except_pad_offset_quirk():
    g = _load("shard_scatter_gather")
    p = json.loads((GOLDEN / "manifest.json").read_text())["shard_scatter_gather"]["params"]
    c = sharded.ShardedSearchClient(
        shards={"a": _TableClient(g["a_scr"], g["a_idx"]), "b": _TableClient(g["b_scr"], g["b_idx"])}, offsets=p["offsets"]
    )
    out = c.search(text=[""] * 5, vector=np.zeros((5, 4), dtype=np.float32), shard=p["shard"], top_k=p["top_k"])
    np.testing.assert_array_equal(out.scores, g["out_scr"])
    ref = g["out_idx"]
    pad = np.isneginf(g["out_scr"])
    np.testing.assert_array_equal(out.indices[~pad], ref[~pad])
    assert np.all(out.indices[pad] == -1)        # ours: pads stay -1
    assert set(ref[pad].tolist()) <= {-1, 99}    # reference: pad of shard "b" became offset-1 (quirk Q1)
    import asyncio

    out2 = asyncio.run(c.async_search(text=[""] * 5, vector=np.zeros((5, 4), dtype=np.float32), shard=p["shard"], top_k=p["top_k"]))
    np.testing.assert_array_equal(out2.indices, out.indices)
    with pytest.raises(ValueError):
        c.search(text=[""], shard=None)
    with pytest.raises(ValueError):
        c.search(text=[""], shard=["nope"])


# ---- master / client --------------------------------------------------------------------------------
def test_master_refuses_pickling_and_client_is_picklable(tmp_path):
    m = vclient.HipMipsMaster(tmp_path / "v.npy", port=12345, skip_setup=True)
    with pytest.raises(base.DoNotPickleError):
        pickle.dumps(m)
    c = pickle.loads(pickle.dumps(m.get_client()))
    assert c.url == "http://localhost:12345" and c.requires_vectors
    assert m.service_name == "hip_mips_master-12345"
    with m as mm:  # skip_setup: no server is spawned
        assert mm._server_proc is None
    assert vclient.HipMipsMaster(tmp_path / "v.npy", port=-1, skip_setup=True).port > 0
    cmd = m._make_cmd()
    assert "-m" in cmd and "vod_amd.search.server" in cmd and "--vectors-path" in cmd


def test_client_ping_false_when_no_server():
    from vod_amd.search.socket import find_available_port

    assert vclient.HipMipsClient(port=find_available_port()).ping() is False


# ---- server contract, with a test double engine (the oracle) -----------------------------------------
class _OracleEngine:
    def __init__(self, x):
        self.x = x

    @property
    def ntotal(self):
        return len(self.x)

    def search(self, q, k):
        from oracle.flat_ip import flat_ip_topk

        if q.shape[1] != self.x.shape[1]:
            raise ValueError("dimension mismatch")
        return flat_ip_topk(q, self.x, k)


def test_server_routes_and_wire_format():
    from fastapi.testclient import TestClient

    from vod_amd.search.server import create_app

    rng = np.random.default_rng(0)
    x = rng.integers(-4, 5, size=(50, 8)).astype(np.float32)
    q = rng.integers(-4, 5, size=(3, 8)).astype(np.float32)
    http = TestClient(create_app(_OracleEngine(x)))
    assert "OK" in http.get("/").text
    assert "ERROR" in TestClient(create_app(_OracleEngine(x[:0]))).get("/").text  # empty index is not healthy (Q12)
    r = http.post("/fast-search", json={"vectors": vio.serialize_np_array(q), "top_k": 60})
    assert r.status_code == 200
    scores = vio.deserialize_np_array(r.json()["scores"])
    ids = vio.deserialize_np_array(r.json()["indices"])
    assert scores.dtype == np.float32 and ids.dtype == np.int64 and scores.shape == (3, 60)
    assert np.all(ids[:, 50:] == -1) and np.all(np.isneginf(scores[:, 50:]))
    r2 = http.post("/search", json={"vectors": q.tolist(), "top_k": 5})
    assert r2.status_code == 200 and np.array(r2.json()["indices"]).tolist() == ids[:, :5].tolist()
    # binary transport (extension): raw npy in, raw f32|i64 out
    import io as _io

    buf = _io.BytesIO()
    np.save(buf, q)
    r3 = http.post("/raw-search", params={"top_k": 7}, content=buf.getvalue(), headers={"content-type": "application/octet-stream"})
    assert r3.status_code == 200 and r3.headers["x-nq"] == "3" and r3.headers["x-k"] == "7"
    bs = np.frombuffer(r3.content, dtype=np.float32, count=21).reshape(3, 7)
    bi = np.frombuffer(r3.content, dtype=np.int64, count=21, offset=84).reshape(3, 7)
    np.testing.assert_array_equal(bs, scores[:, :7])
    np.testing.assert_array_equal(bi, ids[:, :7])
    # contract errors
    assert http.post("/fast-search", json={"vectors": vio.serialize_np_array(q), "top_k": 3, "extra": 1}).status_code == 422
    bad = http.post("/fast-search", json={"vectors": vio.serialize_np_array(q[0]), "top_k": 3})
    assert bad.status_code == 500 and "Expected 2D array" in bad.json()["detail"]
    assert http.post("/fast-search", json={"vectors": vio.serialize_np_array(q[:, :4]), "top_k": 3}).status_code == 500


def test_store_roundtrip_and_factory_protocol(tmp_path):
    from vod_amd import factory, store

    x = np.random.default_rng(1).normal(size=(1000, 16)).astype(np.float32)
    p = store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16, chunk=300)
    back = store.open_vectors(p)
    np.testing.assert_array_equal(np.asarray(back), x.astype(np.float16))
    assert store.fingerprint_vectors(x) == store.fingerprint_vectors(x.copy())
    assert store.fingerprint_vectors(x) != store.fingerprint_vectors(x + 1)
    calls = []
    m0 = factory.build_hip_mips_index(x, config={"port": 23456}, cache_dir=tmp_path, barrier_fn=calls.append, skip_setup=False)
    m1 = factory.build_hip_mips_index(x, config={"port": 23456}, cache_dir=tmp_path, barrier_fn=calls.append, skip_setup=True)
    assert m0.vectors_path == m1.vectors_path and m0.vectors_path.exists() and len(calls) == 2
    assert m1.skip_setup and m0.get_client().url == m1.get_client().url
    with pytest.raises(FileNotFoundError):
        factory.build_hip_mips_index(x + 2, config={"port": 23456}, cache_dir=tmp_path / "other", skip_setup=True)
    with pytest.raises(ValueError):
        factory.build_hip_mips_index(x, config={"factory": "IVF100,Flat"}, cache_dir=tmp_path)


def _native_error():
    from vod_amd._native import NativeLibraryError

    return NativeLibraryError


def test_native_batcher_fuses_concurrent_requests_and_splits_results():
    """`vodhip_batcher` with a callback engine (no GPU needed): four callers, held together by a fixed window, go out as fewer scans with
    k = max(k_i); every caller gets its own rows and columns - identical to separate searches."""
    import threading

    from vod_amd.search.native import NativeBatcher

    rng = np.random.default_rng(2)
    x = rng.integers(-4, 5, size=(500, 8)).astype(np.float32)

    class Counting(_OracleEngine):
        calls = 0
        sizes: list = []

        def search(self, q, k):
            Counting.calls += 1
            Counting.sizes.append((len(q), k))
            return super().search(q, k)

    eng = Counting(x)
    mb = NativeBatcher(engine=eng, dim=8, window_us=250_000)
    qs = [rng.integers(-4, 5, size=(n, 8)).astype(np.float32) for n in (3, 1, 5, 2)]
    ks = [4, 9, 2, 9]
    out = [None] * 4

    def work(i):
        out[i] = mb.search(qs[i], ks[i], client=i + 1)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=20)
    assert Counting.calls < 4 and sum(n for n, _ in Counting.sizes) == 11      # fused into fewer scans
    assert all(k == 9 for n, k in Counting.sizes if n > 5) or Counting.calls >= 1
    st = mb.stats()
    assert st["requests"] == 4 and st["queries"] == 11 and st["batches"] == Counting.calls and st["fused_requests_max"] >= 2
    from oracle.flat_ip import flat_ip_topk

    for i in range(4):
        rs, ri = flat_ip_topk(qs[i], x, ks[i])
        np.testing.assert_array_equal(out[i][1], ri)                               # identical to separate searches
        np.testing.assert_array_equal(out[i][0], rs)
    # argument errors come back as errors of THAT call; the engine's own exception type survives the callback
    with pytest.raises(_native_error()):
        mb.search(qs[0], 0)
    with pytest.raises(ValueError):
        mb.search(np.zeros((2, 5), np.float32), 3)
    mb.close()


# ---- zarr v2 vector store (the reference's tensorstore hand-off format, ts_factory.py:57-92) ----
def _blosc1_frame(data: bytes, typesize: int, blocksize: int, cname: str, shuffle: bool, dont_split: bool) -> bytes:
    """Assemble a blosc-1 frame from its published layout (header, block starts, split streams), test side only."""
    import struct
    import zlib

    import pyarrow as pa

    fmt = {"lz4": 1, "zlib": 3, "zstd": 4}[cname]

    def comp(b: bytes) -> bytes:
        if cname == "lz4":
            return pa.Codec("lz4_raw").compress(b).to_pybytes()
        if cname == "zstd":
            return pa.Codec("zstd").compress(b).to_pybytes()
        return zlib.compress(b)

    nbytes = len(data)
    nblocks = (nbytes + blocksize - 1) // blocksize
    flags = (1 if shuffle else 0) | (0x10 if dont_split else 0) | (fmt << 5)
    body, starts = b"", []
    base = 16 + 4 * nblocks
    for b in range(nblocks):
        blk = data[b * blocksize : (b + 1) * blocksize]
        if shuffle and typesize > 1:
            nel = len(blk) // typesize
            arr = np.frombuffer(blk, dtype=np.uint8, count=nel * typesize).reshape(nel, typesize).T
            blk = np.ascontiguousarray(arr).tobytes() + blk[nel * typesize :]
        leftover = len(blk) != blocksize
        split = (not dont_split) and typesize <= 16 and blocksize // typesize >= 128 and not leftover
        nstreams = typesize if split else 1
        ssize = len(blk) // nstreams
        starts.append(base + len(body))
        for s in range(nstreams):
            piece = blk[s * ssize : (s + 1) * ssize]
            c = comp(piece)
            if len(c) >= len(piece):  # incompressible: stored raw, size == expected size
                c = piece
            body += struct.pack("<i", len(c)) + c
    head = struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, base + len(body))
    return head + struct.pack(f"<{nblocks}i", *starts) + body


@pytest.mark.parametrize("cname,shuffle,dont_split", [("lz4", True, False), ("lz4", False, False), ("zstd", True, True), ("zlib", True, False)])
def test_blosc1_container_decoder(cname, shuffle, dont_split):
    from vod_amd.zarr_store import blosc1_decompress

    rng = np.random.default_rng(3)
    x = np.round(rng.standard_normal(2500), 1).astype(np.float32)  # compressible, 10,000 bytes: 2 full blocks + leftover
    raw = x.tobytes()
    frame = _blosc1_frame(raw, 4, 4096, cname, shuffle, dont_split)
    assert blosc1_decompress(frame) == raw
    # stored (memcpy) frames and the refusal of unsupported inner codecs
    import struct

    assert blosc1_decompress(struct.pack("<BBBBIII", 2, 1, 0x2, 4, len(raw), len(raw), 16 + len(raw)) + raw) == raw
    bad = bytearray(frame)
    bad[2] = (bad[2] & 0x1F) | (0 << 5)  # blosclz
    with pytest.raises(NotImplementedError, match="blosclz"):
        blosc1_decompress(bytes(bad))


@pytest.mark.parametrize("dtype,compressor", [(np.float32, None), (np.float16, {"id": "zlib", "level": 1}), (np.float32, "blosc")])
def test_zarr_vector_store_reads_what_the_reference_layout_declares(tmp_path, dtype, compressor):
    from vod_amd import store
    from vod_amd.zarr_store import ZarrVectors, write_zarr_vectors

    rng = np.random.default_rng(7)
    x = np.round(rng.standard_normal((1234, 48)), 2).astype(dtype)
    path = tmp_path / "vectors"
    if compressor == "blosc":  # tensorstore's default compressor: blosc / lz4 / byte shuffle
        write_zarr_vectors(path, x, dtype=dtype, chunk_size=100)
        meta = json.loads((path / ".zarray").read_text())
        meta["compressor"] = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}
        (path / ".zarray").write_text(json.dumps(meta))
        for f in path.iterdir():
            if f.name[0].isdigit():
                f.write_bytes(_blosc1_frame(f.read_bytes(), x.dtype.itemsize, 8192, "lz4", True, False))
    else:
        write_zarr_vectors(path, x, dtype=dtype, chunk_size=100, compressor=compressor)
    assert json.loads((path / "factory.json").read_text())["driver"] == "zarr"
    z = store.open_vectors(path)
    assert isinstance(z, ZarrVectors) and z.shape == x.shape and len(z) == 1234 and z.dtype == x.dtype
    np.testing.assert_array_equal(z[0:1234], x)
    np.testing.assert_array_equal(z[95:305], x[95:305])  # crosses chunk boundaries
    np.testing.assert_array_equal(z[1233], x[1233])
    np.testing.assert_array_equal(z[-1], x[-1])
    got = np.concatenate([rows for _lo, rows in z.iter_row_blocks(300)])
    np.testing.assert_array_equal(got, x)
    # a chunk that was never written reads as the fill value
    (path / "3.0").unlink()
    assert np.all(np.isnan(z[300:400])) and np.array_equal(z[200:300], x[200:300])
    # the factory serves the zarr array in place (no npy copy), the npy hand-off accepts it as a row source
    from vod_amd import factory

    m = factory.build_hip_mips_index(ZarrVectors(path), config={"port": 23457}, cache_dir=tmp_path / "cache", skip_setup=True)
    assert m.vectors_path == path and not (tmp_path / "cache" / "indices").exists()
    np.testing.assert_array_equal(np.load(store.save_vectors(tmp_path / "v.npy", ZarrVectors(path)[0:300].astype(np.float16))), x[:300].astype(np.float16))


def test_samples_to_dict_has_the_collate_field_contract():
    """Keys / dtypes of `_samples_to_dict` (realm_collate.py:247-278), the hand-off to RetrievalGradients."""
    torch = pytest.importorskip("torch")
    from vod_amd.core.sample import PrioritySampledSections, samples_to_dict

    b, n = 3, 4
    ps = PrioritySampledSections(
        batch=vt.RetrievalBatch(indices=np.arange(b * n).reshape(b, n), scores=np.ones((b, n), np.float64), labels=np.eye(b, n, dtype=np.int64)),
        log_weights=np.zeros((b, n), np.float32), max_sampling_id=np.zeros(b), lse_pos=np.zeros(b, np.float32),
        lse_neg=np.ones(b, np.float32), raw_scores={"dense": np.full((b, n), np.nan, np.float32)},
    )
    rel = [[1.0, 0.0, 0.0, 0.0]] * b
    d = samples_to_dict(ps, rel, prefix="section__")
    assert set(d) == {f"section__{k}" for k in ("idx", "score", "label", "relevance", "log_weight", "lse_pos", "lse_neg", "dense")}
    assert d["section__label"].dtype == np.bool_ and d["section__relevance"] is rel
    t = samples_to_dict(ps, rel, prefix="section__", as_torch=True)
    assert t["section__score"].dtype == torch.float32 and t["section__label"].dtype == torch.bool
    assert t["section__relevance"].dtype == torch.float32 and t["section__idx"].dtype == torch.int64
    assert t["section__lse_neg"].shape == (b,) and torch.isnan(t["section__dense"]).all()
    ps_nolabel = PrioritySampledSections(batch=vt.RetrievalBatch(indices=np.zeros((1, 1), np.int64), scores=np.zeros((1, 1))),
                                         log_weights=np.zeros((1, 1)), max_sampling_id=np.zeros(1), lse_pos=np.zeros(1), lse_neg=np.zeros(1), raw_scores={})
    with pytest.raises(ValueError):
        samples_to_dict(ps_nolabel, [[0.0]])


def test_factory_port_is_resolved_once_for_all_ranks(tmp_path):
    """`port < 0` (pick a free port) must yield the SAME port on every rank: rank 0 picks, `broadcast_fn` hands it on
    (the reference: `_resolve_ports`, src/vod_search/factory.py:380-394); a connect-only rank may not invent its own."""
    import numpy as np

    from vod_amd import factory

    vecs = np.zeros((4, 8), dtype=np.float32)
    cfg = factory.HipMipsFactoryConfig(port=-1)
    sent = {}

    def bcast(p):  # rank 0's side of fabric.broadcast(p, 0)
        sent["port"] = p
        return p

    m0 = factory.build_hip_mips_index(vecs, config=cfg, cache_dir=tmp_path, broadcast_fn=bcast)
    assert m0.port == sent["port"] and m0.port > 0
    m1 = factory.build_hip_mips_index(vecs, config=cfg, cache_dir=tmp_path, skip_setup=True, broadcast_fn=lambda _p: sent["port"])
    assert m1.port == m0.port and m1.get_client().port == m0.port
    with pytest.raises(ValueError, match="resolve the port once"):
        factory.build_hip_mips_index(vecs, config=cfg, cache_dir=tmp_path, skip_setup=True)
    # the CONFIG default is the reference config's -1 = "pick a free port" (src/vod_configs/search.py:134): two default-config
    # indexes on one host do not collide; 6637 is only the master's constructor default (faiss_search/client.py:124)
    assert factory.HipMipsFactoryConfig().port == -1
    a = factory.build_hip_mips_index(vecs, cache_dir=tmp_path)
    b = factory.build_hip_mips_index(vecs, cache_dir=tmp_path)
    assert a.port > 0 and b.port > 0
    assert vclient.HipMipsMaster(tmp_path / "v.npy", skip_setup=True).port == 6637


def test_master_command_line_for_a_multi_gpu_group(tmp_path):
    m = vclient.HipMipsMaster(tmp_path / "v.npy", devices=[0, 1, 2, 3], port=7001, skip_setup=True)
    cmd = m._make_cmd()
    assert cmd[cmd.index("--devices") + 1] == "0,1,2,3" and "--device" not in cmd
    m1 = vclient.HipMipsMaster(tmp_path / "v.npy", device=2, port=7001, skip_setup=True)
    cmd1 = m1._make_cmd()
    assert cmd1[cmd1.index("--device") + 1] == "2" and "--devices" not in cmd1


# ---- the HTTP shell - libvodhip's native front (the one hot-route server) - over real sockets -----------------------------------
SHELLS = ["native"]


def _serve_in_thread(engine, micro_batch_wait_ms=0.0, uds=None, shell="native"):
    assert shell == "native"
    from vod_amd.search.native import NativeHttpFront
    from vod_amd.search.server import Endpoints

    endpoints = Endpoints(engine, micro_batch_wait_ms)
    batcher = endpoints._batcher_for(8)  # the test engines have no fixed dimension of their own: every store here is 8 wide
    endpoints.batcher = batcher
    front = NativeHttpFront(batcher, endpoints)
    port = front.listen("127.0.0.1", 0, uds)
    front.start()

    def stop_native():
        front.close()
        endpoints.close()

    stop_native.front = front
    return port, stop_native


@pytest.mark.parametrize("shell", SHELLS)
def test_http_shell_routes_wire_format_and_errors(shell):
    import http.client

    import requests

    rng = np.random.default_rng(0)
    x = rng.integers(-4, 5, size=(50, 8)).astype(np.float32)
    q = rng.integers(-4, 5, size=(3, 8)).astype(np.float32)
    port, stop = _serve_in_thread(_OracleEngine(x), shell=shell)
    try:
        url = f"http://127.0.0.1:{port}"
        c = vclient.HipMipsClient("http://127.0.0.1", port)
        assert c.ping() and requests.get(url + "/").json() == "OK"
        # the reference's own client code path: requests.post(json=...) with the reference codec strings
        r = requests.post(url + "/fast-search", json={"vectors": vio.serialize_np_array(q), "top_k": 60})
        assert r.status_code == 200 and set(r.json()) == {"scores", "indices"}
        scores, ids = vio.deserialize_np_array(r.json()["scores"]), vio.deserialize_np_array(r.json()["indices"])
        assert scores.dtype == np.float32 and ids.dtype == np.int64 and scores.shape == (3, 60)
        assert np.all(ids[:, 50:] == -1) and np.all(np.isneginf(scores[:, 50:]))
        # our client, both routes, float32 and float16 on the wire, over ONE kept-alive connection each
        for binary in (False, True):
            for wire in (None, "float16"):
                cl = vclient.HipMipsClient("http://127.0.0.1", port, binary=binary, wire_dtype=wire)
                for _ in range(3):
                    res = cl.search(vector=q, top_k=7)
                np.testing.assert_array_equal(res.indices, ids[:, :7])
                np.testing.assert_array_equal(res.scores, scores[:, :7])
                clone = pickle.loads(pickle.dumps(cl))  # a DataLoader worker's copy opens its own connection
                np.testing.assert_array_equal(clone.search(vector=q, top_k=7).indices, ids[:, :7])
        r2 = requests.post(url + "/search", json={"vectors": q.tolist(), "top_k": 5})
        assert r2.status_code == 200 and np.array(r2.json()["indices"]).tolist() == ids[:, :5].tolist()
        # contract errors: same status codes and bodies as the FastAPI shell
        assert requests.post(url + "/fast-search", json={"vectors": vio.serialize_np_array(q), "top_k": 3, "extra": 1}).status_code == 422
        assert requests.post(url + "/fast-search", json={"vectors": 3, "top_k": 3}).status_code == 422
        assert requests.post(url + "/fast-search", data=b"[1, 2]").status_code == 422
        bad = requests.post(url + "/fast-search", json={"vectors": vio.serialize_np_array(q[0]), "top_k": 3})
        assert bad.status_code == 500 and "Expected 2D array" in bad.json()["detail"]
        assert requests.post(url + "/fast-search", json={"vectors": vio.serialize_np_array(q[:, :4]), "top_k": 3}).status_code == 500
        with pytest.raises(requests.HTTPError):
            c.search(vector=q[:, :4], top_k=3)
        assert c.search(vector=q, top_k=2).indices.tolist() == ids[:, :2].tolist()  # the connection survived the 500
        assert requests.get(url + "/nope").status_code == 404 and requests.get(url + "/fast-search").status_code == 405
        assert requests.post(url + "/raw-search?top_k=x", data=b"").status_code == 422
        # protocol corners: Expect: 100-continue, chunked upload refused, two pipelined requests on one connection
        hc = http.client.HTTPConnection("127.0.0.1", port)
        body = bytes(vio.json_body_with_arrays({"vectors": q}, {"top_k": 4}))
        hc.request("POST", "/fast-search", body=body, headers={"Expect": "100-continue", "content-type": "application/json"})
        resp = hc.getresponse()
        assert resp.status == 200 and b"scores" in resp.read()
        hc.putrequest("POST", "/fast-search")
        hc.putheader("Transfer-Encoding", "chunked")
        hc.endheaders()
        assert hc.getresponse().status == 501
        import socket as _socket

        s = _socket.create_connection(("127.0.0.1", port))
        one = b"POST /fast-search HTTP/1.1\r\nHost: x\r\nContent-Length: %d\r\n\r\n" % len(body) + body
        s.sendall(one + one)
        got = b""
        while got.count(b'"scores"') < 2:
            chunk = s.recv(65536)
            assert chunk, "connection closed before both pipelined replies arrived"
            got += chunk
        s.close()
        # reply bytes: identical whichever shell (and, for the native front, whichever side - native codec or host fallback) wrote them
        want = bytes(vio.json_body_with_arrays({"scores": scores[:, :4], "indices": ids[:, :4]}))
        assert requests.post(url + "/fast-search", data=body).content == want
        assert requests.post(url + "/fast-search", data=body[:-1] + b', "subset_ids": null}').content == want
        if shell == "native":
            st = requests.get(url + "/stats").json()
            assert st["requests_native"] >= 20 and st["requests_fallback"] >= 8 and st["queries"] >= 60
            # requests the native parser leaves to the host end up in the same batcher: counted once, answered identically
            assert stop.front.get_stat("open_connections") >= 0
    finally:
        stop()


@pytest.mark.parametrize("shell", SHELLS)
def test_http_shell_on_a_unix_domain_socket(tmp_path, shell):
    """the same routes on a Unix-domain socket (`--uds`); the client's searches go through it, `ping` keeps using TCP,
    a pickled client (DataLoader worker) re-opens its own connection, and the socket file is removed on shutdown."""
    import os

    from oracle.flat_ip import flat_ip_topk

    rng = np.random.default_rng(6)
    x = rng.integers(-4, 5, size=(200, 8)).astype(np.float32)
    q = rng.integers(-4, 5, size=(5, 8)).astype(np.float32)
    path = str(tmp_path / "vodhip.sock")
    port, stop = _serve_in_thread(_OracleEngine(x), uds=path, shell=shell)
    try:
        assert os.path.exists(path)
        rs, ri = flat_ip_topk(q, x, 9)
        import socket as _socket

        for binary in (False, True):
            for native in (True, False):  # libvodhip's client / the Python exchange
                c = vclient.HipMipsClient("http://127.0.0.1", port, binary=binary, uds=path, native=native)
                assert c.ping()
                for _ in range(3):
                    res = c.search(vector=q, top_k=9)
                np.testing.assert_array_equal(res.indices, ri)
                np.testing.assert_array_equal(res.scores, rs)
                if native:
                    assert c._local.native_h is not None and getattr(c._local, "lean", None) is None   # the searches never left the library
                else:
                    assert c._local.lean.sock.family == _socket.AF_UNIX        # ... went through the Unix-domain socket
                clone = pickle.loads(pickle.dumps(c))
                np.testing.assert_array_equal(clone.search(vector=q, top_k=9).indices, ri)
        # a socket path that does not exist on this host (a client on another machine, a rank that derived another path - round-3 advisor):
        # the searches fall back to the TCP address instead of failing
        for native in (True, False):
            far = vclient.HipMipsClient("http://127.0.0.1", port, uds=str(tmp_path / "nobody-listens.sock"), native=native)
            np.testing.assert_array_equal(far.search(vector=q, top_k=9).indices, ri)
            if not native:
                assert far._local.lean.sock.family != _socket.AF_UNIX
        # a second server must not take over (unlink) a socket a live listener still answers on (round-4 advisor)
        with pytest.raises(Exception, match="live server"):
            _serve_in_thread(_OracleEngine(x), uds=path, shell=shell)
        assert os.path.exists(path)
        np.testing.assert_array_equal(vclient.HipMipsClient("http://127.0.0.1", port, uds=path).search(vector=q, top_k=9).indices, ri)
        # conflicting repeated Content-Length headers are a 400 (RFC 9110), agreeing ones are fine
        import http.client as _hc

        for lens, want in (((5, 7), 400), ((2, 2), 200)):
            raw = _socket.create_connection(("127.0.0.1", port))
            raw.sendall(b"POST /fast-search HTTP/1.1\r\nHost: x\r\n" + b"".join(b"Content-Length: %d\r\n" % n for n in lens)
                        + b"Connection: close\r\n\r\n" + b"{}" + b"x" * 8)
            status = int(raw.recv(65536).split(b" ", 2)[1])
            raw.close()
            assert (status == 400) == (want == 400), (lens, status)
    finally:
        stop()
    assert not os.path.exists(path)


def test_default_socket_directory_is_private_to_the_user():
    import os
    import stat

    from vod_amd.search.socket import private_socket_dir

    d = private_socket_dir()
    st = os.stat(d)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not st.st_mode & 0o022
    m = vclient.HipMipsMaster("synthetic:10x8", port=7011, skip_setup=True, uds=True)
    assert os.path.dirname(m.uds) == d and m.uds.endswith("vodhip-7011.sock")


@pytest.mark.parametrize("shell", SHELLS)
def test_http_shell_fuses_concurrent_clients(shell):
    """32 concurrent clients: every caller gets its own exact rows, and the engine saw fewer, larger batches."""
    import concurrent.futures

    from oracle.flat_ip import flat_ip_topk

    rng = np.random.default_rng(5)
    x = rng.integers(-4, 5, size=(400, 8)).astype(np.float32)

    class Counting(_OracleEngine):
        calls = 0

        def search(self, q, k):
            Counting.calls += 1
            return super().search(q, k)

    Counting.calls = 0
    port, stop = _serve_in_thread(Counting(x), micro_batch_wait_ms=20.0, shell=shell)
    try:
        qs = [rng.integers(-4, 5, size=(1 + i % 5, 8)).astype(np.float32) for i in range(32)]

        # ONE client object per route, shared by all threads - as the reference shares its client between executor threads
        # (sharded_search.py:159-167): each thread gets its own kept-alive connection
        shared = {b: vclient.HipMipsClient("http://127.0.0.1", port, binary=b) for b in (False, True)}

        def one(i):
            return shared[bool(i % 2)].search(vector=qs[i], top_k=3 + i % 4)

        with concurrent.futures.ThreadPoolExecutor(32) as pool:
            results = list(pool.map(one, range(32)))
        for i, res in enumerate(results):
            rs, ri = flat_ip_topk(qs[i], x, 3 + i % 4)
            np.testing.assert_array_equal(res.indices, ri)
            np.testing.assert_array_equal(res.scores, rs)
        assert Counting.calls < 32
    finally:
        stop()


def test_usable_cpus_honours_affinity_and_quota(monkeypatch):
    """`vod_amd.hostcpu`: the thread-pool cap of the server processes = scheduler affinity capped by the cgroup quota."""
    import builtins
    import io
    import os

    from vod_amd import hostcpu

    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(256)), raising=False)
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if str(path) == "/sys/fs/cgroup/cpu.max":
            return io.StringIO("1600000 100000\n")
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert hostcpu.usable_cpus() == 16
    monkeypatch.setattr(builtins, "open", lambda path, *a, **k: io.StringIO("max 100000\n") if str(path) == "/sys/fs/cgroup/cpu.max" else real_open(path, *a, **k))
    assert hostcpu.usable_cpus() == 256
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: {0, 1, 2}, raising=False)
    assert hostcpu.usable_cpus() == 3
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    import torch

    before = torch.get_num_threads()
    try:
        assert hostcpu.limit_cpu_threads(2) == 2 and os.environ["OMP_NUM_THREADS"] == "2" and torch.get_num_threads() <= 2
    finally:
        torch.set_num_threads(before)


@wall_clock_test()
def test_native_batcher_gathers_the_next_batch_while_the_engine_is_busy():
    """Batch-while-busy, no window: the first request runs at once; requests that arrive while it is on the (slow) engine end up in ONE
    following batch; the engine is never entered twice at a time; a failing batch fails its own callers only."""
    import threading
    import time

    from oracle.flat_ip import flat_ip_topk
    from vod_amd.search.native import NativeBatcher

    rng = np.random.default_rng(3)
    x = rng.integers(-4, 5, size=(300, 8)).astype(np.float32)

    class Slow(_OracleEngine):
        def __init__(self, rows):
            super().__init__(rows)
            self.sizes, self.inside, self.overlap = [], 0, False

        def search(self, q, k):
            self.inside += 1
            self.overlap |= self.inside > 1
            self.sizes.append(len(q))
            time.sleep(0.15)
            try:
                if k == 7:
                    raise KeyError("injected engine failure")
                return super().search(q, k)
            finally:
                self.inside -= 1

    eng = Slow(x)
    mb = NativeBatcher(engine=eng, dim=8)
    qs = [rng.integers(-4, 5, size=(2, 8)).astype(np.float32) for _ in range(9)]
    out, errs = [None] * 9, []

    def work(i):
        out[i] = mb.search(qs[i], 5, client=100 + i)

    first = threading.Thread(target=work, args=(0,))
    t0 = time.monotonic()
    first.start()
    time.sleep(0.05)  # request 0 is on the engine now; the other eight arrive during its 150 ms
    rest = [threading.Thread(target=work, args=(i,)) for i in range(1, 9)]
    for t in rest:
        t.start()
    for t in [first] + rest:
        t.join(timeout=30)
    assert time.monotonic() - t0 < 0.15 * 2 + 0.14    # (three scans would be 0.45 s) request 0 did not wait for anybody, the eight others shared ONE scan
    assert not eng.overlap
    assert eng.sizes == [2, 16], eng.sizes               # the eight waiting requests went out as one batch of 16 queries
    for i in range(9):
        rs, ri = flat_ip_topk(qs[i], x, 5)
        np.testing.assert_array_equal(out[i][1], ri)
        np.testing.assert_array_equal(out[i][0], rs)

    def bad():
        try:
            mb.search(qs[0], 7)
        except KeyError as e:
            errs.append(e)

    bt = threading.Thread(target=bad)
    bt.start()
    bt.join(timeout=30)
    assert len(errs) == 1
    s1, i1 = mb.search(qs[1], 5)                        # the batcher keeps serving after a failed batch
    np.testing.assert_array_equal(i1, flat_ip_topk(qs[1], x, 5)[1])
    mb.close()


@wall_clock_test()
def test_native_batcher_waits_for_expected_company_only():
    """The grace rule: with several recently active clients an idle engine waits (a bounded moment) for the ones still missing, so closed-loop
    clients stay in ONE batch instead of falling into two alternating groups; a lone client never waits."""
    import threading
    import time

    from vod_amd.search.native import NativeBatcher

    rng = np.random.default_rng(4)
    x = rng.integers(-4, 5, size=(200, 8)).astype(np.float32)

    class Timed(_OracleEngine):
        def __init__(self, rows):
            super().__init__(rows)
            self.sizes = []

        def search(self, q, k):
            self.sizes.append(len(q))
            time.sleep(0.04)  # a "scan" of 40 ms
            return super().search(q, k)

    eng = Timed(x)
    mb = NativeBatcher(engine=eng, dim=8, grace_us=30_000, grace_pct=60)   # grace = min(30 ms, 60 % of the measured scan) = 24 ms
    q = rng.integers(-4, 5, size=(2, 8)).astype(np.float32)
    lone = []
    for _ in range(3):
        t0 = time.monotonic()
        mb.search(q, 3, client=1)                       # a lone client: three scans, no waiting (also measures the scan)
        lone.append(time.monotonic() - t0)
    assert min(lone) < 0.04 + 0.018 and sorted(lone)[1] < 0.04 + 0.05, lone   # (a grace wait would add 24 ms to EVERY call; min / median: a loaded box hiccups)
    assert mb.get_stat("flat_scan_ns") > 30e6

    # four closed-loop clients whose come-back time (5-15 ms) is shorter than the grace: once the batcher has learnt each client's rhythm
    # (two returns), every round must be ONE batch of all four
    eng.sizes.clear()
    rounds = 8

    def loop(c, n=rounds, pause=None):
        for _ in range(n):
            mb.search(q, 3, client=10 + c)
            time.sleep(0.005 + 0.003 * c if pause is None else pause)

    threads = [threading.Thread(target=loop, args=(c,)) for c in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    full = sum(1 for n in eng.sizes if n == 8)
    assert full >= rounds - 4, eng.sizes
    assert mb.get_stat("grace_waits") >= rounds - 4
    # ... while a client that pauses LONGER than the grace between its requests (a worker tokenising its next batch: 70 ms here, grace
    # 24 ms) is not waited for: the closed-loop client next to it keeps the latency of a plain scan
    lat = []

    def fast():
        for _ in range(6):
            t0 = time.monotonic()
            mb.search(q, 3, client=31)
            lat.append(time.monotonic() - t0)
            time.sleep(0.004)

    slow = threading.Thread(target=loop, args=(20, 5, 0.07))
    fastt = threading.Thread(target=fast)
    slow.start()
    time.sleep(0.2)  # the slow client's rhythm is known by now
    fastt.start()
    fastt.join(timeout=60)
    slow.join(timeout=60)
    # most of its requests cost one plain scan; the others queued behind the slow client's scan or were fused with it when it WAS due
    # within the grace (70 ms after its last answer) - never "every request + the whole grace", which is what waiting for any recently
    # seen client gave
    plain = sum(1 for v in lat if v < 0.04 + 0.018)
    assert plain >= 2, lat
    # a client that went away is not waited for once it is forgotten
    for c in (11, 12, 13, 30, 31):   # (the HTTP front forgets a client when its connection closes)
        mb.forget_client(c)
    after = []
    for _ in range(3):
        t0 = time.monotonic()
        mb.search(q, 3, client=10)
        after.append(time.monotonic() - t0)
    assert min(after) < 0.04 + 0.018 and sorted(after)[1] < 0.04 + 0.05, after
    mb.close()


def _np_header(arr):
    import io as _io

    head = _io.BytesIO()
    np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(arr))
    return head.getvalue()


def test_native_wire_pieces_match_numpy_and_the_python_codec():
    """The native front's codec against NumPy and `vod_amd.io`, byte for byte, without a socket: the `.npy` header it writes, the
    headers it accepts, the /fast-search document it takes (and the ones it leaves to the host), the reply body it builds."""
    import ctypes
    import json

    from vod_amd import _native
    from vod_amd import io as vio

    lib = _native.load_library()
    buf = (ctypes.c_uint8 * 256)()
    rng = np.random.default_rng(9)
    for rows, cols in [(1, 1), (32, 100), (64, 768), (1024, 768), (7, 2048), (99999, 3), (123456789, 12), (5, 10 ** 6), (10 ** 11, 1)]:
        for code, dt in ((2, np.float32), (0, np.float16), (3, np.int64)):
            n = lib.vodhip_wire_npy_header(code, rows, cols, buf, 256)
            want = _np_header(np.lib.stride_tricks.as_strided(np.zeros(1, dt), shape=(rows, cols), strides=(0, 0)))
            assert bytes(buf[:n]) == want, (rows, cols, dt)
    # parse: what NumPy writes for 2-D float32 / float16 is accepted with the right geometry; other layouts are left to the host
    dt_c, r_c, c_c, off_c = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()

    def parse(raw: bytes):
        rc = lib.vodhip_wire_parse_npy(raw, len(raw), ctypes.byref(dt_c), ctypes.byref(r_c), ctypes.byref(c_c), ctypes.byref(off_c))
        return rc, dt_c.value, r_c.value, c_c.value, off_c.value

    import io as _io

    for arr in (rng.normal(size=(3, 8)).astype(np.float32), rng.normal(size=(17, 5)).astype(np.float16), np.zeros((0, 4), np.float32)):
        f = _io.BytesIO()
        np.save(f, arr)
        raw = f.getvalue()
        rc, dt, r, c, off = parse(raw)
        assert rc == 0 and (r, c) == arr.shape and dt == (2 if arr.dtype == np.float32 else 0)
        assert np.array_equal(np.frombuffer(raw, arr.dtype, offset=off).reshape(arr.shape), arr)
        assert parse(raw[:-1])[0] == -1 if arr.size else True                   # truncated data
    for arr in (np.zeros((3, 8), np.float64), np.zeros(5, np.float32), np.zeros((2, 3, 4), np.float32), np.asfortranarray(np.ones((3, 4), np.float32)),
                np.zeros((3, 4), ">f4"), np.zeros((3, 4), np.int32)):
        f = _io.BytesIO()
        np.save(f, arr)
        assert parse(f.getvalue())[0] == -1, arr.dtype
    assert parse(b"not an npy file at all")[0] == -1
    # the /fast-search document
    vb, ve, tk = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()

    def doc(body: bytes):
        rc = lib.vodhip_wire_parse_fast_search(body, len(body), ctypes.byref(vb), ctypes.byref(ve), ctypes.byref(tk))
        return rc, body[vb.value : ve.value] if rc == 0 else None, tk.value

    arr = rng.normal(size=(4, 8)).astype(np.float32)
    body = bytes(vio.json_body_with_arrays({"vectors": arr}, {"top_k": 7}))             # what `HipMipsClient` sends
    rc, payload, k = doc(body)
    assert rc == 0 and k == 7 and np.array_equal(vio.deserialize_np_array(payload), arr)
    ref_body = json.dumps({"vectors": vio.serialize_np_array(arr), "top_k": 12}).encode()  # what the reference's client sends (requests json=)
    rc, payload, k = doc(ref_body)
    assert rc == 0 and k == 12 and payload == vio.serialize_np_array(arr).encode()
    assert doc(b' {"top_k":5 ,\n "vectors" : "QUJD" , "subset_ids": null } ')[::2] == (0, 5)
    assert doc(b'{"vectors": "QUJD"}')[::2] == (0, 3)                                    # the model's default (models.py)
    for other in (b'{"vectors": "QUJD", "top_k": 5, "extra": 1}', b'{"vectors": "QU\\u0041", "top_k": 5}', b'{"vectors": "QUJD", "top_k": 5.0}',
                  b'{"vectors": "QUJD", "top_k": "5"}', b'{"vectors": "QUJD", "subset_ids": [["a"]]}', b'{"top_k": 5}', b'[1, 2]',
                  b'{"vectors": "QUJD", "vectors": "QUJD"}', b'{"vectors": "QUJD", "top_k": 5} trailing', b'{"vectors": ["QUJD"]}', b'{"vectors": "QUJD"', b''):
        assert doc(other)[0] == 1, other
    # the reply body = the Python shell's, byte for byte (and so the golden wire strings of tests/golden/io_codec.json)
    for nq, k in [(1, 1), (3, 10), (32, 100), (64, 7), (5, 2048)]:
        scores = rng.normal(size=(nq, k)).astype(np.float32)
        scores[0, -1] = -np.inf
        ids = rng.integers(-1, 10 ** 9, size=(nq, k)).astype(np.int64)
        want = bytes(vio.json_body_with_arrays({"scores": scores, "indices": ids}))
        need = lib.vodhip_wire_fast_search_reply(None, None, nq, k, None, 0)
        assert need == len(want)
        out = ctypes.create_string_buffer(need)
        assert lib.vodhip_wire_fast_search_reply(scores.ctypes.data, ids.ctypes.data, nq, k, out, need) == need
        assert out.raw == want


def test_native_http_front_survives_malformed_and_hostile_requests():
    """The native front parses untrusted bytes in C++: random mutations of valid requests, truncated heads and bodies, lying
    Content-Lengths, binary junk, oversized heads - every connection must end in a well-formed HTTP reply or a clean close, never a
    crash or a hang, and a valid request must still be answered bit-exactly afterwards (and between the attacks)."""
    import socket as _socket

    from oracle.flat_ip import flat_ip_topk

    rng = np.random.default_rng(99)
    x = rng.integers(-4, 5, size=(300, 8)).astype(np.float32)
    q = rng.integers(-4, 5, size=(4, 8)).astype(np.float32)
    port, stop = _serve_in_thread(_OracleEngine(x), shell="native")
    try:
        good_fast = bytes(vio.json_body_with_arrays({"vectors": q}, {"top_k": 5}))
        buf = __import__("io").BytesIO()
        np.save(buf, q)
        good_raw = buf.getvalue()

        def head(path, n, extra=b""):
            return b"POST " + path + b" HTTP/1.1\r\nHost: x\r\n" + extra + b"Content-Length: " + str(n).encode() + b"\r\n\r\n"

        def exchange(payload: bytes, wait=0.5) -> bytes:
            s = _socket.create_connection(("127.0.0.1", port), timeout=5)
            s.settimeout(wait)
            got = b""
            try:
                s.sendall(payload)
                try:
                    s.shutdown(_socket.SHUT_WR)
                except OSError:
                    pass
                while True:
                    chunk = s.recv(65536)
                    if not chunk:
                        break
                    got += chunk
            except (OSError, _socket.timeout):
                pass
            finally:
                s.close()
            return got

        def check_alive():
            c = vclient.HipMipsClient("http://127.0.0.1", port)
            res = c.search(vector=q, top_k=5)
            rs, ri = flat_ip_topk(q, x, 5)
            np.testing.assert_array_equal(res.indices, ri)
            np.testing.assert_array_equal(res.scores, rs)

        attacks = [
            b"", b"\r\n\r\n", b"GET", b"POST /fast-search HTTP/1.1\r\n\r\n", b"\x00" * 5000, bytes(rng.integers(0, 256, size=3000, dtype=np.uint8)),
            head(b"/fast-search", len(good_fast) + 50) + good_fast,                       # body shorter than announced, then EOF
            head(b"/fast-search", 10) + good_fast,                                         # body longer than announced (rest parsed as a request)
            head(b"/fast-search", 2 ** 62),                                                # absurd length -> 413
            b"POST /fast-search HTTP/1.1\r\nContent-Length: -5\r\n\r\n", b"POST /fast-search HTTP/1.1\r\nContent-Length: 1e3\r\n\r\n",
            b"POST /fast-search HTTP/1.1\r\n" + b"X-Pad: " + b"a" * 70000 + b"\r\n\r\n",    # 64 KB of headers -> 431
            head(b"/fast-search", 9) + b'{"vectors',                                       # truncated JSON
            head(b"/fast-search", 2) + b"{}", head(b"/fast-search", 4) + b"null", head(b"/fast-search", 27) + b'{"vectors": "", "top_k": 3}',
            head(b"/fast-search", 34) + b'{"vectors": "!!!!####", "top_k": 3}',
            head(b"/raw-search?top_k=5", 20) + b"\x93NUMPY\x01\x00\xff\xff" + b"x" * 10,        # header length beyond the body
            head(b"/raw-search?top_k=5", len(good_raw) - 7) + good_raw[:-7],                    # truncated data
            head(b"/raw-search?top_k=99999999999999999999", len(good_raw)) + good_raw,
            head(b"/raw-search?top_k=5&x=" + b"y" * 3000, len(good_raw)) + good_raw,
        ]
        for a in attacks:
            reply = exchange(a)
            assert reply == b"" or reply.startswith(b"HTTP/1.1 "), reply[:80]
        check_alive()
        # random mutations of the two valid requests: flip / delete / insert bytes anywhere (head and body)
        for trial in range(300):
            base = bytearray((head(b"/fast-search", len(good_fast)) + good_fast) if trial % 2 else (head(b"/raw-search?top_k=5", len(good_raw)) + good_raw))
            for _ in range(int(rng.integers(1, 6))):
                pos = int(rng.integers(0, len(base)))
                op = int(rng.integers(0, 3))
                if op == 0:
                    base[pos] = int(rng.integers(0, 256))
                elif op == 1:
                    del base[pos : pos + int(rng.integers(1, 9))]
                else:
                    base[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8))
            reply = exchange(bytes(base), wait=0.3)
            assert reply == b"" or reply.startswith(b"HTTP/1.1 "), (trial, reply[:80])
            if trial % 60 == 59:
                check_alive()
        check_alive()
        assert stop.front.get_stat("open_connections") <= 2  # nothing leaked: the attack connections are gone
    finally:
        stop()


def test_wire_codec_vector_paths_equal_pythons_base64():
    """`vodhip_b64url_encode / _decode` (wire_codec.cpp: AVX-512 VBMI / AVX2 / table-driven, chosen by cpuid) against Python's `base64` on
    random lengths around every vector width, with a head || data seam at every offset class, both alphabets on the way in, and corrupted
    text refused (the caller then falls back to the lenient library decoder)."""
    import base64
    import ctypes

    from vod_amd import _native

    lib = _native.load_library()
    rng = np.random.default_rng(12)

    def enc(head: bytes, data: bytes) -> bytes:
        out = ctypes.create_string_buffer(4 * ((len(head) + len(data) + 2) // 3) + 8)
        n = lib.vodhip_b64url_encode(head, len(head), data, len(data), out)
        return out.raw[:n]

    def dec(text: bytes):
        out = (ctypes.c_uint8 * (3 * len(text) // 4 + 8))()
        n = lib.vodhip_b64url_decode(text, len(text), out)
        return None if n < 0 else bytes(out[:n])

    sizes = list(range(0, 140)) + [191, 192, 193, 255, 256, 257, 1000, 4093, 4096, 65537]
    for nd in sizes:
        for nh in (0, 1, 2, 3, 64, 128):
            head = bytes(rng.integers(0, 256, size=nh, dtype=np.uint8))
            data = bytes(rng.integers(0, 256, size=nd, dtype=np.uint8))
            ref = base64.urlsafe_b64encode(head + data)
            assert enc(head, data) == ref, (nh, nd)
            assert dec(ref) == head + data, (nh, nd)
            assert dec(base64.b64encode(head + data)) == head + data, (nh, nd)   # the standard alphabet is accepted too
            body = ref.rstrip(b"=")
            if len(body) > 8:
                broken = bytearray(ref)
                broken[int(rng.integers(0, len(body)))] = int(rng.choice(list(b"!@# \n*~\x80\xff")))
                assert dec(bytes(broken)) is None, (nh, nd)


def test_native_batcher_can_be_closed_while_callers_are_inside():
    """Shutdown with searches in every state.  (1) `vodhip_batcher_destroy` itself: one request running in the engine, five queued
    behind it - each caller returns (its rows, or the "shut down" error), none hangs, the handle is freed after the last one left.
    (2) `NativeBatcher.close()` racing threads that search in a loop: the calls already counted finish normally, later ones are refused
    by the wrapper and never reach the freed handle."""
    import threading
    import time

    from vod_amd.search.native import NativeBatcher

    rng = np.random.default_rng(5)
    x = rng.integers(-4, 5, size=(200, 8)).astype(np.float32)

    class Slow(_OracleEngine):
        def search(self, q, k):
            time.sleep(0.08)
            return super().search(q, k)

    # (1) the library
    mb = NativeBatcher(engine=Slow(x), dim=8, grace_us=0)
    outcomes: list = []

    def once(i):
        q = rng.integers(-4, 5, size=(2, 8)).astype(np.float32)
        time.sleep(0.005 * i)  # the first request starts a batch of its own, the others queue behind it
        try:
            s, ids = mb.search(q, 3, client=i + 1)
            outcomes.append(ids.shape == (2, 3))
        except _native_error() as exc:
            outcomes.append("shut" in str(exc))

    threads = [threading.Thread(target=once, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    deadline = time.monotonic() + 10.0   # until all six are inside the library (assembled into a batch, or queued): on a loaded box too
    while mb.get_stat("requests") + mb.get_stat("pending") < 6 and time.monotonic() < deadline:
        time.sleep(0.002)
    assert mb.get_stat("requests") + mb.get_stat("pending") == 6
    h, mb._h = mb._h, None
    assert mb._lib.vodhip_batcher_destroy(h) == 0
    for t in threads:
        t.join(timeout=20)
    assert not any(t.is_alive() for t in threads) and len(outcomes) == 6 and all(outcomes)
    assert any(o is True for o in outcomes)

    # (2) the wrapper
    mb = NativeBatcher(engine=Slow(x), dim=8)
    outcomes = []

    def loop(i):
        q = rng.integers(-4, 5, size=(2, 8)).astype(np.float32)
        try:
            while True:
                s, ids = mb.search(q, 3, client=i + 1)
                outcomes.append(ids.shape == (2, 3))
        except RuntimeError as exc:
            outcomes.append("closed" in str(exc))

    threads = [threading.Thread(target=loop, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    time.sleep(0.2)
    mb.close()
    for t in threads:
        t.join(timeout=20)
    assert not any(t.is_alive() for t in threads) and len(outcomes) >= 12 and all(outcomes)


def test_native_wire_parser_refuses_shapes_that_overflow():
    from vod_amd import _native

    lib = _native.load_library()
    for shape in ("(99999999999, 99999999999)", "(4611686018427387904, 8)", "(3, 4611686018427387904)"):
        head = ("{'descr': '<f4', 'fortran_order': False, 'shape': %s, }" % shape).ljust(117) + "\n"
        blob = b"\x93NUMPY\x01\x00" + len(head).to_bytes(2, "little") + head.encode() + b"\0" * 64
        dt, rows, cols, off = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        assert lib.vodhip_wire_parse_npy(blob, len(blob), ctypes.byref(dt), ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(off)) == -1


def test_native_http_front_sends_any_content_type_the_fallback_chooses():
    import http.client

    from vod_amd.search.native import NativeBatcher, NativeHttpFront

    class Ends:
        def handle(self, method, path, query, body, client=0):
            return 200, "text/" + "x" * 900, b"ok", {"x-long": "y" * 2000}

    mb = NativeBatcher(engine=_OracleEngine(np.zeros((4, 8), np.float32)), dim=8)
    front = NativeHttpFront(mb, Ends())
    port = front.listen("127.0.0.1", 0)
    front.start()
    try:
        c = http.client.HTTPConnection("127.0.0.1", port, timeout=5)
        c.request("GET", "/anything")
        r = c.getresponse()
        assert r.status == 200 and r.read() == b"ok"
        assert r.getheader("content-type") == "text/" + "x" * 900 and r.getheader("x-long") == "y" * 2000
        c.close()
    finally:
        front.close()
        mb.close()


def test_client_lean_connection_and_its_fallbacks():
    """`HipMipsClient` talks to `http://` servers through its own minimal HTTP/1.1 exchange (`_LeanConnection`).  It must survive what a
    kept-alive connection meets: the server closing it between two requests (re-opened once), error replies with bodies, a reply without a
    Content-Length (handed to `http.client` from then on), `Connection: close`."""
    import socket as _socket
    import threading

    from oracle.flat_ip import flat_ip_topk

    rng = np.random.default_rng(8)
    x = rng.integers(-4, 5, size=(100, 8)).astype(np.float32)
    q = rng.integers(-4, 5, size=(3, 8)).astype(np.float32)
    rs, ri = flat_ip_topk(q, x, 4)
    port, stop = _serve_in_thread(_OracleEngine(x), shell="native")
    try:
        c = vclient.HipMipsClient("http://127.0.0.1", port, native=False)
        np.testing.assert_array_equal(c.search(vector=q, top_k=4).indices, ri)
        first = c._local.lean
        first.sock.shutdown(_socket.SHUT_RDWR)                         # what an idle timeout on the server's side looks like from here
        np.testing.assert_array_equal(c.search(vector=q, top_k=4).indices, ri)
        assert c._local.lean is not first and c._local.lean.used
        with pytest.raises(vclient.requests.exceptions.HTTPError, match="500"):
            c.search(vector=q, top_k=0)                                # an error reply (with its JSON body) on the same connection
        np.testing.assert_array_equal(c.search(vector=q, top_k=4).indices, ri)
    finally:
        stop()

    # a server that answers without Content-Length (close-delimited body): the client falls back to http.client for good
    body = bytes(vio.json_body_with_arrays({"scores": rs, "indices": ri}))
    srv = _socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(4)

    def serve():
        for _ in range(2):
            conn, _a = srv.accept()
            data = b""
            while b"\r\n\r\n" not in data:
                data += conn.recv(65536)
            head, _, rest = data.partition(b"\r\n\r\n")
            need = int([ln for ln in head.split(b"\r\n") if ln.lower().startswith(b"content-length")][0].split(b":")[1])
            while len(rest) < need:
                rest += conn.recv(65536)
            conn.sendall(b"HTTP/1.1 200 OK\r\ncontent-type: application/json\r\nconnection: close\r\n\r\n" + body)
            conn.close()

    t = threading.Thread(target=serve, daemon=True)
    t.start()
    try:
        c2 = vclient.HipMipsClient("http://127.0.0.1", srv.getsockname()[1], native=False)
        np.testing.assert_array_equal(c2.search(vector=q, top_k=4).indices, ri)
        assert c2._local.no_lean is True
    finally:
        t.join(timeout=10)
        srv.close()


def test_npy_readers_take_every_layout_the_reference_codec_can_send():
    """The readers parse the `.npy` header natively for the service's own layouts (2-D float32 / float16 / int64) and through NumPy for
    everything else: same arrays either way, views where the alignment allows, errors for truncated payloads."""
    rng = np.random.default_rng(12)
    cases = [rng.normal(size=(5, 7)).astype(np.float32), rng.normal(size=(3, 4)).astype(np.float16), rng.integers(-9, 9, size=(6, 2)).astype(np.int64),
             rng.normal(size=(4, 3)), rng.integers(0, 5, size=(7,)).astype(np.int32), rng.normal(size=(2, 3, 4)).astype(np.float32),
             np.zeros((0, 8), np.float32), np.asfortranarray(rng.normal(size=(3, 5)).astype(np.float32)), rng.uniform(size=(4, 4)) > 0.5]
    for arr in cases:
        text = vio.serialize_np_array(arr)
        body = b'{"k": "' + text.encode() + b'"}'
        got = vio.deserialize_np_array_span(body, 7, 7 + len(text))
        np.testing.assert_array_equal(got, arr)
        assert got.dtype == arr.dtype and got.shape == arr.shape
        buf = __import__("io").BytesIO()
        np.save(buf, arr)
        view = vio.load_npy_view(buf.getvalue())
        np.testing.assert_array_equal(view, arr)
        assert view.dtype == arr.dtype
    # a header that promises more data than the payload holds
    buf = __import__("io").BytesIO()
    np.save(buf, cases[0])
    cut = buf.getvalue()[:-8]
    with pytest.raises(Exception):
        vio.load_npy_view(cut)


def test_native_client_of_the_library():
    """`vodhip_client_*` (what `HipMipsClient` uses by default, and what a non-Python consumer links): both routes and both query dtypes
    equal the oracle; the request bytes are the Python codec's; a stale keep-alive is re-opened; an error reply comes back as its HTTP
    status + body (-> HTTPError with the server's trace); an unframed reply or a dead address falls through to the Python path, which
    reports it; a timeout is a ReadTimeout."""
    import socket as _socket
    import threading

    from oracle.flat_ip import flat_ip_topk
    from vod_amd import _native

    rng = np.random.default_rng(21)
    x = rng.integers(-4, 5, size=(150, 8)).astype(np.float32)
    q = rng.integers(-4, 5, size=(6, 8)).astype(np.float32)
    rs, ri = flat_ip_topk(q, x, 7)
    port, stop = _serve_in_thread(_OracleEngine(x), shell="native")
    lib = _native.load_library()
    try:
        for binary in (False, True):
            for dt in (np.float32, np.float16):
                c = vclient.HipMipsClient("http://127.0.0.1", port, binary=binary, wire_dtype=np.dtype(dt).name)
                res = c.search(vector=q, top_k=7)
                np.testing.assert_array_equal(res.indices, ri)
                np.testing.assert_array_equal(res.scores, rs)
                assert c._local.native_h is not None and res.meta["time"] > 0
        # plain C-ABI use, as a cgo / JNI consumer would: create, search, error reply, destroy
        h = ctypes.c_void_p()
        assert lib.vodhip_client_create(b"127.0.0.1", port, None, ctypes.byref(h)) == 0
        s, i = np.empty((6, 7), np.float32), np.empty((6, 7), np.int64)
        for route in (0, 1):
            assert lib.vodhip_client_search(h, q.ctypes.data, 2, 6, 8, 7, route, 5.0, s.ctypes.data, i.ctypes.data) == 0
            np.testing.assert_array_equal(i, ri)
            np.testing.assert_array_equal(s, rs)
        # top_k outside [1, VODHIP_MAX_K] never leaves the native client (a 200 reply would not fit the caller's [nq, k] buffers)
        for bad_k in (0, -3, 5000):
            assert lib.vodhip_client_search(h, q.ctypes.data, 2, 6, 8, bad_k, 0, 5.0, s.ctypes.data, i.ctypes.data) == -1
            assert b"out of range" in lib.vodhip_last_error()
        assert lib.vodhip_client_search(h, q.ctypes.data, 2, 6, 8, 7, 0, 5.0, s.ctypes.data, i.ctypes.data) == 0   # the connection survived
        bad = np.zeros((6, 5), np.float32)                                                                            # wrong dimension: 500 as well
        assert lib.vodhip_client_search(h, bad.ctypes.data, 2, 6, 5, 7, 1, 5.0, s.ctypes.data, i.ctypes.data) == 500
        assert b"detail" in lib.vodhip_client_last_body(h)
        assert lib.vodhip_client_search(h, q.ctypes.data, 2, 6, 8, 7, 1, 5.0, s.ctypes.data, i.ctypes.data) == 0   # the connection survived
        assert lib.vodhip_client_destroy(h) == 0
        # through the Python client: HTTPError with the status, then business as usual
        c = vclient.HipMipsClient("http://127.0.0.1", port)
        with pytest.raises(vclient.requests.exceptions.HTTPError, match="500"):
            c.search(vector=q, top_k=5000)
        np.testing.assert_array_equal(c.search(vector=q, top_k=7).indices, ri)
    finally:
        stop()
    # the server is gone: the native client fails, the Python path reports it the way callers of the reference client expect
    with pytest.raises(vclient.requests.exceptions.ConnectionError):
        c.search(vector=q, top_k=7)

    # a server that never answers: ReadTimeout from the native path
    srv = _socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(2)
    held = []
    t = threading.Thread(target=lambda: held.append(srv.accept()), daemon=True)
    t.start()
    try:
        slow = vclient.HipMipsClient("http://127.0.0.1", srv.getsockname()[1])
        with pytest.raises(vclient.requests.exceptions.ReadTimeout):
            slow.search(vector=q, top_k=7, timeout=0.3)
    finally:
        t.join(timeout=5)
        for conn, _a in held:
            conn.close()
        srv.close()


def test_native_client_reopens_a_connection_the_server_closed_while_idle():
    import socket as _socket
    import threading

    rng = np.random.default_rng(22)
    q = rng.integers(-4, 5, size=(3, 8)).astype(np.float32)
    scores = rng.normal(size=(3, 4)).astype(np.float32)
    ids = rng.integers(0, 99, size=(3, 4)).astype(np.int64)
    body = bytes(vio.json_body_with_arrays({"scores": scores, "indices": ids}))
    srv = _socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(4)
    seen = []

    def serve():
        for _ in range(2):  # answers ONE request per connection, keeps quiet about closing it
            conn, _a = srv.accept()
            data = b""
            while b"\r\n\r\n" not in data:
                data += conn.recv(65536)
            head, _, rest = data.partition(b"\r\n\r\n")
            need = int([ln for ln in head.split(b"\r\n") if ln.lower().startswith(b"content-length")][0].split(b":")[1])
            while len(rest) < need:
                rest += conn.recv(65536)
            seen.append(rest)
            conn.sendall(b"HTTP/1.1 200 OK\r\ncontent-type: application/json\r\ncontent-length: %d\r\n\r\n" % len(body) + body)
            conn.close()

    t = threading.Thread(target=serve, daemon=True)
    t.start()
    try:
        c = vclient.HipMipsClient("http://127.0.0.1", srv.getsockname()[1])
        for _ in range(2):
            res = c.search(vector=q, top_k=4)
            np.testing.assert_array_equal(res.indices, ids)
            np.testing.assert_array_equal(res.scores, scores)
        assert c._local.native_h is not None and getattr(c._local, "lean", None) is None   # both went through the library
        # the request document is the Python codec's, byte for byte (what the reference server parses)
        assert seen[0] == bytes(vio.json_body_with_arrays({"vectors": q}, {"top_k": 4})) == seen[1]
    finally:
        t.join(timeout=10)
        srv.close()


def test_factory_serves_float32_vectors_exactly_by_default(tmp_path):
    """`exact_f32=None` (the default): float32 / float64 vectors - what the reference's index holds (build.py:65-73) - get a float32 hand-off
    file and a server started with --exact-f32, so the drop-in returns the reference's result unless the caller opts out; float16 vectors
    are served as they are; an explicit True / False wins."""
    from vod_amd import factory

    rng = np.random.default_rng(0)
    x32 = rng.normal(size=(50, 8)).astype(np.float32)
    cases = [(x32, None, True, np.float32), (x32.astype(np.float64), None, True, np.float32), (x32.astype(np.float16), None, False, np.float16),
             (x32, False, False, np.float16), (x32.astype(np.float16), True, True, np.float32)]
    for j, (x, flag, want_exact, want_file) in enumerate(cases):
        cfg = {"port": 23470 + j}
        if flag is not None:
            cfg["exact_f32"] = flag
        m = factory.build_hip_mips_index(x, config=cfg, cache_dir=tmp_path / str(j))
        assert m.exact_f32 is want_exact and ("--exact-f32" in m._make_cmd()) is want_exact
        assert np.load(m.vectors_path, mmap_mode="r").dtype == want_file
    # the two modes of the same vectors never share a cached store
    a = factory.build_hip_mips_index(x32, config={"port": 23480}, cache_dir=tmp_path / "c")
    b = factory.build_hip_mips_index(x32, config={"port": 23481, "exact_f32": False}, cache_dir=tmp_path / "c")
    assert a.vectors_path != b.vectors_path
