"""End to end on the GPU box: HipMipsMaster spawns the real server process (owner of the GPU index), the
picklable HTTP client queries it -- the counterpart of /root/reference/examples/search/faiss.py
(config 1: 100k x 384 fp32 vectors, batch 32, top-10) -- and the result must equal the CPU oracle."""
import os
import pickle

import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_master_client_roundtrip_config1(tmp_path, monkeypatch):
    from oracle.flat_ip import flat_ip_topk
    from vod_amd import factory
    from vod_amd.search import ShardedSearchMaster

    monkeypatch.chdir(tmp_path)  # the master writes <service>-<port>.std{out,err}.log into the cwd
    rng = np.random.default_rng(0)
    n, d, nq, k = 100_000, 384, 32, 10
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    q = rng.integers(-8, 9, size=(nq, d)).astype(np.float32)
    master = factory.build_hip_mips_index(x, config={"port": -1, "logging_level": "warning"}, cache_dir=tmp_path)
    sharded = ShardedSearchMaster(shards={"corpus": master}, offsets={"corpus": 1000})
    with sharded:
        client = pickle.loads(pickle.dumps(sharded.get_client()))  # what a DataLoader worker receives
        assert client.ping()
        res = client.search(text=[""] * nq, vector=q, shard=["corpus"] * nq, top_k=k)
        rs, ri = flat_ip_topk(q, x, k, id_base=1000)
        np.testing.assert_array_equal(res.indices, ri)
        np.testing.assert_array_equal(res.scores, rs)
        assert res.scores.dtype == np.float32 and res.indices.dtype == np.int64
        direct = master.get_client()
        r2 = direct.search(vector=q[:3], top_k=5)
        assert r2.meta["time"] > 0 and r2.indices.shape == (3, 5)
        r3 = direct.search_py(q[:3], top_k=5)
        np.testing.assert_array_equal(r3.indices, r2.indices)
        binary = type(direct)(host=direct.host, port=direct.port, binary=True)  # raw-bytes transport, same results
        r4 = binary.search(vector=q, top_k=k)
        np.testing.assert_array_equal(r4.indices + 1000, ri)
        np.testing.assert_array_equal(r4.scores, rs)
        import requests

        with pytest.raises(requests.exceptions.HTTPError):
            direct.search(vector=q[0], top_k=5)  # 1-D query -> HTTP 500 with the trace in `detail`
    assert not master.get_client().ping()  # server terminated on exit
    assert os.path.exists(f"{master.service_name}.stderr.log")


def test_index_save_load_roundtrip(tmp_path):
    from vod_amd.index import HipFlatIndex

    rng = np.random.default_rng(1)
    x = rng.normal(size=(5000, 96)).astype(np.float32)
    q = torch.from_numpy(rng.normal(size=(9, 96)).astype(np.float32)).cuda()
    for dt in (torch.float16, torch.bfloat16):
        with HipFlatIndex(96, 5000, dtype=dt) as a:
            a.add(x)
            a.save(tmp_path / "store.npy", chunk=1234)
            ref = a.search(q, 20)
            stored = a.stored_rows()
        with HipFlatIndex.load(tmp_path / "store.npy", dtype=dt, chunk=999) as b:
            assert b.ntotal == 5000
            assert torch.equal(b.stored_rows(), stored)  # re-rounding stored values is the identity
            got = b.search(q, 20)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def test_server_subset_ids_roundtrip(tmp_path, monkeypatch):
    """`subset_ids` forwarded over the wire and honoured by the GPU index (the reference's faiss client drops them)."""
    from oracle.flat_ip import topk_desc_tiebreak
    from vod_amd import store
    from vod_amd.search.client import HipMipsClient, HipMipsMaster

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(3)
    n, d, nq, k = 20000, 64, 12, 10
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    q = rng.integers(-8, 9, size=(nq, d)).astype(np.float32)
    names = np.array([f"doc{v}" for v in rng.integers(0, 7, size=n)])
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    np.save(tmp_path / "subsets.npy", names)

    class Master(HipMipsMaster):
        def _make_cmd(self):
            return super()._make_cmd() + ["--subset-ids-path", str(tmp_path / "subsets.npy")]

    subset_ids = [[f"doc{r % 7}"] if r % 3 else [] for r in range(nq)]
    subset_ids[1] = ["doc1", "doc5"]
    subset_ids[2] = ["nope"]  # unknown id: restricted to nothing
    with Master(tmp_path / "v.npy", port=-1, logging_level="warning") as m:
        c = HipMipsClient(host=m.host, port=m.port, forward_subset_ids=True)
        res = c.search(vector=q, subset_ids=subset_ids, top_k=k)
        plain = m.get_client().search(vector=q, subset_ids=subset_ids, top_k=k)  # reference behaviour: ignored
    full = q.astype(np.float64) @ x.astype(np.float64).T
    rs, ri = topk_desc_tiebreak(full, k)
    np.testing.assert_array_equal(plain.indices, ri)
    masked = full.copy()
    for r, names_r in enumerate(subset_ids):
        if names_r:
            masked[r, ~np.isin(names, names_r)] = np.nan
    ms, mi = topk_desc_tiebreak(masked, k)
    np.testing.assert_array_equal(res.indices, mi)
    np.testing.assert_array_equal(res.scores, ms)
    assert np.all(res.indices[2] == -1)


def test_engine_ingests_the_reference_zarr_store_directly(tmp_path):
    """the tensorstore/zarr array the predict loop writes (float32, 100-row chunks) goes to HBM
    in one pass -- no float32 faiss copy, no index file."""
    from oracle.flat_ip import flat_ip_topk
    from vod_amd.search.server import HipEngine
    from vod_amd.zarr_store import write_zarr_vectors

    rng = np.random.default_rng(5)
    n, d, nq, k = 20_050, 96, 17, 20
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    q = rng.integers(-8, 9, size=(nq, d)).astype(np.float32)
    path = write_zarr_vectors(tmp_path / "vectors", x, dtype=np.float32, chunk_size=100, compressor={"id": "zlib", "level": 1})
    engine = HipEngine(str(path))
    assert engine.ntotal == n
    s, i = engine.search(q, k)
    rs, ri = flat_ip_topk(q, x, k)
    np.testing.assert_array_equal(i, ri)
    np.testing.assert_array_equal(s, rs)


def test_multi_gpu_group_server_with_one_device(tmp_path, monkeypatch):
    """`HipMipsMaster(devices=[0])`: owner process -> one worker per listed GPU on an RCCL group -> rank 0 answers HTTP.
    With one device the whole N-rank path runs (broadcast-free dispatch, local search, packed RCCL all-gather from
    itself, merge); results must equal the oracle, subset filtering included, and the group must die with the master.
    Counterpart of `FaissMaster(serve_on_gpu=True)` (/root/reference/src/vod_search/faiss_search/client.py:118-137)."""
    from oracle.flat_ip import flat_ip_topk, topk_desc_tiebreak
    from vod_amd import factory, store
    from vod_amd.search.client import HipMipsClient, HipMipsMaster

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(8)
    n, d, nq, k = 60_000, 128, 300, 100
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    q = rng.integers(-8, 9, size=(nq, d)).astype(np.float32)
    master = factory.build_hip_mips_index(x, config={"port": -1, "logging_level": "warning"}, cache_dir=tmp_path, devices=[0])
    assert master.devices == [0] and "--devices" in master._make_cmd()
    with master:
        client = master.get_client()
        assert client.ping()
        for lo, hi, kk in [(0, nq, k), (0, 1, 5), (10, 43, 100)]:
            res = client.search(vector=q[lo:hi], top_k=kk)
            rs, ri = flat_ip_topk(q[lo:hi], x, kk)
            np.testing.assert_array_equal(res.indices, ri)
            np.testing.assert_array_equal(res.scores, rs)
    assert not master.get_client().ping()

    names = np.array([f"doc{v}" for v in rng.integers(0, 5, size=n)])
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    np.save(tmp_path / "subsets.npy", names)

    class Master(HipMipsMaster):
        def _make_cmd(self):
            return super()._make_cmd() + ["--subset-ids-path", str(tmp_path / "subsets.npy")]

    subset_ids = [[f"doc{r % 5}"] if r % 2 else [] for r in range(40)]
    with Master(tmp_path / "v.npy", port=-1, logging_level="warning", devices=[0]) as m:
        c = HipMipsClient(host=m.host, port=m.port, forward_subset_ids=True)
        res = c.search(vector=q[:40], subset_ids=subset_ids, top_k=20)
    masked = q[:40].astype(np.float64) @ x.astype(np.float64).T
    for r, names_r in enumerate(subset_ids):
        if names_r:
            masked[r, ~np.isin(names, names_r)] = np.nan
    ms, mi = topk_desc_tiebreak(masked, 20)
    np.testing.assert_array_equal(res.indices, mi)
    np.testing.assert_array_equal(res.scores, ms)


def test_multi_gpu_group_server_three_workers_sharing_the_gpu(tmp_path):
    """The N > 1 group on real kernels: `devices=[0, 0, 0]` with `group_backend="gloo"` starts three workers that share the
    box's one GPU (RCCL refuses two ranks on a device; gloo carries the request broadcast and the packed per-shard
    top-k through the host).  Every worker holds its own row shard in its own HIP context, rank 0 answers HTTP:
    broadcast -> three fused local top-k with their row offsets -> all-gather -> HIP merge.  Results must equal the
    oracle over the WHOLE store (ties across shard boundaries included: integer-valued rows), subset filter included."""
    from oracle.flat_ip import flat_ip_topk, topk_desc_tiebreak
    from vod_amd import store
    from vod_amd.search.client import HipMipsClient, HipMipsMaster

    rng = np.random.default_rng(21)
    n, d, nq, k = 50_000, 128, 130, 100
    x = rng.integers(-6, 7, size=(n, d)).astype(np.float32)
    q = rng.integers(-6, 7, size=(nq, d)).astype(np.float32)
    names = np.array([f"doc{v}" for v in rng.integers(0, 5, size=n)])
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    np.save(tmp_path / "subsets.npy", names)

    class Master(HipMipsMaster):
        def _make_cmd(self):
            return super()._make_cmd() + ["--subset-ids-path", str(tmp_path / "subsets.npy")]

    subset_ids = [[f"doc{r % 5}", "doc4"] if r % 2 else [] for r in range(40)]
    with Master(tmp_path / "v.npy", port=-1, logging_level="warning", devices=[0, 0, 0], group_backend="gloo") as m:
        assert "--group-backend" in m._make_cmd()
        c = HipMipsClient(host=m.host, port=m.port, forward_subset_ids=True)
        assert c.ping()
        for lo, hi, kk in [(0, nq, k), (5, 6, 3), (0, 70, 250)]:
            res = c.search(vector=q[lo:hi], top_k=kk)
            rs, ri = flat_ip_topk(q[lo:hi], x, kk)
            np.testing.assert_array_equal(res.indices, ri)
            np.testing.assert_array_equal(res.scores, rs)
        # malformed requests (round-2 advisor: they killed every worker of the group) come back as HTTP 500 and the group keeps serving
        import requests

        for bad_k in (0, 5000):
            with pytest.raises(requests.HTTPError):
                c.search(vector=q[:4], top_k=bad_k)
        with pytest.raises(requests.HTTPError):
            c.search(vector=q[:4, :7], top_k=5)
        with pytest.raises(requests.HTTPError):
            c.search(vector=q[:2], subset_ids=[[f"doc{j}" for j in range(70)], []], top_k=5)
        assert c.ping()
        res = c.search(vector=q[:40], subset_ids=subset_ids, top_k=20)
        t_exit = time.monotonic()
    assert time.monotonic() - t_exit < 20, "orderly shutdown: rank 0 stops the workers, nobody waits for the kill timeout"
    assert not m.get_client().ping()
    masked = q[:40].astype(np.float64) @ x.astype(np.float64).T
    for r, names_r in enumerate(subset_ids):
        if names_r:
            masked[r, ~np.isin(names, names_r)] = np.nan
    ms, mi = topk_desc_tiebreak(masked, 20)
    np.testing.assert_array_equal(res.indices, mi)
    np.testing.assert_array_equal(res.scores, ms)


def test_single_process_node_server_three_shards(tmp_path):
    """`devices=[0, 0, 0], group_backend="node"`: no worker processes - the ONE server process drives every device through
    `vodhip_node_index_*` (the shape of the reference's server with `faiss.index_cpu_to_all_gpus(shard=True)`).  Same contract as
    the worker groups above: the oracle's answer over the whole store, subset filter included, errors as HTTP 500."""
    import requests

    from oracle.flat_ip import flat_ip_topk, topk_desc_tiebreak
    from vod_amd import store
    from vod_amd.search.client import HipMipsClient, HipMipsMaster

    rng = np.random.default_rng(22)
    n, d, nq, k = 50_000, 128, 130, 100
    x = rng.integers(-6, 7, size=(n, d)).astype(np.float32)
    q = rng.integers(-6, 7, size=(nq, d)).astype(np.float32)
    names = np.array([f"doc{v}" for v in rng.integers(0, 5, size=n)])
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    np.save(tmp_path / "subsets.npy", names)

    class Master(HipMipsMaster):
        def _make_cmd(self):
            return super()._make_cmd() + ["--subset-ids-path", str(tmp_path / "subsets.npy")]

    subset_ids = [[f"doc{r % 5}", "doc4"] if r % 2 else [] for r in range(40)]
    with Master(tmp_path / "v.npy", port=-1, logging_level="warning", devices=[0, 0, 0], group_backend="node") as m:
        c = HipMipsClient(host=m.host, port=m.port, forward_subset_ids=True)
        assert c.ping()
        for lo, hi, kk in [(0, nq, k), (5, 6, 3), (0, 70, 250)]:
            res = c.search(vector=q[lo:hi], top_k=kk)
            rs, ri = flat_ip_topk(q[lo:hi], x, kk)
            np.testing.assert_array_equal(res.indices, ri)
            np.testing.assert_array_equal(res.scores, rs)
        for bad_k in (0, 5000):
            with pytest.raises(requests.HTTPError):
                c.search(vector=q[:4], top_k=bad_k)
        with pytest.raises(requests.HTTPError):
            c.search(vector=q[:4, :7], top_k=5)
        assert c.ping()
        res = c.search(vector=q[:40], subset_ids=subset_ids, top_k=20)
    assert not m.get_client().ping()
    masked = q[:40].astype(np.float64) @ x.astype(np.float64).T
    for r, names_r in enumerate(subset_ids):
        if names_r:
            masked[r, ~np.isin(names, names_r)] = np.nan
    ms, mi = topk_desc_tiebreak(masked, 20)
    np.testing.assert_array_equal(res.indices, mi)
    np.testing.assert_array_equal(res.scores, ms)


def test_micro_batcher_on_the_gpu_engine_mixes_plain_and_subset_requests(tmp_path):
    """SURVEY 8(f) row 4 on the real engine: concurrent requests are fused into shared GPU batches, each caller gets its
    own rows / k, and subset requests (which bypass the batcher) are serialised with the fused batches by ONE lock
    (round-1 advisor finding: they used to enter the index concurrently)."""
    import concurrent.futures

    from fastapi.testclient import TestClient
    from oracle.flat_ip import flat_ip_topk, topk_desc_tiebreak
    from vod_amd import io as vio
    from vod_amd import store
    from vod_amd.search.server import HipEngine, create_app

    rng = np.random.default_rng(13)
    n, d = 30_000, 64
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    names = np.array([f"doc{v}" for v in rng.integers(0, 4, size=n)])
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    np.save(tmp_path / "subsets.npy", names)
    engine = HipEngine(str(tmp_path / "v.npy"), subset_ids_path=str(tmp_path / "subsets.npy"))
    app = create_app(engine, micro_batch_wait_ms=20.0)
    jobs = []
    for j in range(24):
        q = rng.integers(-8, 9, size=(int(rng.integers(1, 9)), d)).astype(np.float32)
        k = int(rng.choice([3, 10, 50]))
        sub = [["doc1"] for _ in range(len(q))] if j % 3 == 0 else None
        jobs.append((q, k, sub))

    def call(job):
        q, k, sub = job
        with TestClient(app) as tc:
            body = {"vectors": vio.serialize_np_array(q), "top_k": k}
            if sub is not None:
                body["subset_ids"] = sub
            r = tc.post("/fast-search", json=body)
            assert r.status_code == 200, r.text
            data = r.json()
            return vio.deserialize_np_array(data["scores"]), vio.deserialize_np_array(data["indices"])

    with concurrent.futures.ThreadPoolExecutor(8) as pool:
        results = list(pool.map(call, jobs))
    for (q, k, sub), (s, i) in zip(jobs, results):
        full = q.astype(np.float64) @ x.astype(np.float64).T
        if sub is not None:
            full[:, names != "doc1"] = np.nan
        rs, ri = topk_desc_tiebreak(full, k)
        np.testing.assert_array_equal(i, ri)
        np.testing.assert_array_equal(s, rs)


def test_multi_gpu_group_server_eight_workers_sharing_the_gpu(tmp_path):
    """The node's full width behind one address: `devices=[0] * 8` on gloo = 8 worker processes (one row shard of a 2 M-row store
    each, 8 HIP contexts on the box's one GPU), rank 0 answering HTTP.  Request broadcast -> 8 fused local top-k with their row
    offsets -> ONE packed all-gather -> 8-way HIP merge; ids and scores must equal the oracle over the WHOLE store (integer-valued
    rows: ties across all seven shard boundaries)."""
    from oracle.flat_ip import flat_ip_topk
    from vod_amd import store
    from vod_amd.search.client import HipMipsClient, HipMipsMaster

    rng = np.random.default_rng(88)
    n, d, nq, k = 2_000_000, 64, 96, 100
    x = np.empty((n, d), dtype=np.float16)
    for lo in range(0, n, 250_000):
        x[lo : lo + 250_000] = rng.integers(-6, 7, size=(250_000, d), dtype=np.int8)
    q = rng.integers(-6, 7, size=(nq, d)).astype(np.float32)
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    rs, ri = flat_ip_topk(q, x, k)
    with HipMipsMaster(tmp_path / "v.npy", port=-1, logging_level="warning", devices=[0] * 8, group_backend="gloo") as m:
        c = HipMipsClient(host=m.host, port=m.port)
        assert c.ping()
        res = c.search(vector=q, top_k=k)
        np.testing.assert_array_equal(res.indices, ri)
        np.testing.assert_array_equal(res.scores, rs)
        assert len({int(i) * 8 // n for i in ri.ravel()}) == 8, "the answer must draw on every shard"
        raw = HipMipsClient(host=m.host, port=m.port, binary=True, wire_dtype="float16").search(vector=q[:7], top_k=13)
        np.testing.assert_array_equal(raw.indices, ri[:7, :13])
    assert not m.get_client().ping()


def test_engine_orders_concurrent_callers_itself(tmp_path):
    """`HipEngine.search` from many threads at once: the library's batcher fuses the callers into shared scans (up to two batches in
    flight on its stream; result rows written straight into device-visible host memory) - each must get ITS answer, subset requests
    (never fused) and a malformed one included."""
    import concurrent.futures

    from oracle.flat_ip import flat_ip_topk, topk_desc_tiebreak
    from vod_amd import store
    from vod_amd.search.server import HipEngine

    rng = np.random.default_rng(31)
    n, d = 40_000, 64
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    names = np.array([f"doc{v}" for v in rng.integers(0, 4, size=n)])
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    np.save(tmp_path / "subsets.npy", names)
    engine = HipEngine(str(tmp_path / "v.npy"), subset_ids_path=str(tmp_path / "subsets.npy"))
    jobs = []
    for j in range(60):
        q = rng.integers(-8, 9, size=(int(rng.integers(1, 200)), d)).astype(np.float32)
        k = int(rng.choice([1, 7, 64, 100, 300]))
        sub = [["doc2"] for _ in range(len(q))] if j % 5 == 0 else None
        jobs.append((q, k, sub))

    def call(job):
        q, k, sub = job
        return engine.search(q, k, subset_ids=sub)

    with concurrent.futures.ThreadPoolExecutor(12) as pool:
        futs = [pool.submit(call, job) for job in jobs]
        bad = pool.submit(engine.search, np.zeros((3, d), np.float32), 5000)  # refused at enqueue: holds no ticket, blocks nobody
        results = [f.result(timeout=120) for f in futs]
        with pytest.raises(Exception, match="out of range"):
            bad.result(timeout=120)
    assert engine.batcher.get_stat("in_flight") == 0 and engine.batcher.get_stat("pending") == 0
    for (q, k, sub), (s, i) in zip(jobs, results):
        if sub is None:
            rs, ri = flat_ip_topk(q, x, k)
        else:
            full = q.astype(np.float64) @ x.astype(np.float64).T
            full[:, names != "doc2"] = np.nan
            rs, ri = topk_desc_tiebreak(full, k)
        np.testing.assert_array_equal(i, ri)
        np.testing.assert_array_equal(s, rs)


def test_concurrent_callers_while_every_search_overflows(tmp_path):
    """Round-3 advisor (high): `finish` of search i ran its recovery passes on the shared workspace while another thread enqueued search
    i + 1 on the same handle.  Now the handle carries a mutex and ONE completion thread finishes: many threads, candidate lists of 256
    entries and a store whose top scores are tied thousands of times (every batch overflows and recovers) - results must equal the
    serial answers, which equal the oracle."""
    import concurrent.futures

    from oracle.flat_ip import flat_ip_topk
    from vod_amd import store
    from vod_amd.search.server import HipEngine

    rng = np.random.default_rng(77)
    n, d = 60_000, 64
    x = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    x[n // 2 :] = x[n // 2]  # 30 k copies of one row: every query ties far more rows than a candidate list holds
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    engine = HipEngine(str(tmp_path / "v.npy"))
    engine.index.set_param("cand_cap", 256)
    jobs = [(np.sign(x[n // 2])[None, :] * rng.integers(1, 4, size=(int(rng.integers(1, 40)), d))).astype(np.float32) for _ in range(48)]
    ks = [int(rng.choice([5, 50, 200])) for _ in jobs]
    with concurrent.futures.ThreadPoolExecutor(16) as pool:
        results = list(pool.map(lambda a: engine.search(a[0], a[1]), zip(jobs, ks)))
    assert engine.index.get_stat("last_safe_reruns") >= 0
    recovered = 0
    for q, k, (s, i) in zip(jobs, ks, results):
        rs, ri = flat_ip_topk(q, x, k)
        np.testing.assert_array_equal(i, ri)
        np.testing.assert_array_equal(s, rs)
    # the same jobs one by one: the recovery path really ran
    for q, k in list(zip(jobs, ks))[:6]:
        engine.search(q, k)
        recovered += engine.index.get_stat("last_safe_reruns") > 0
    assert recovered >= 1
    engine.close()


def test_engine_can_be_closed_while_searches_are_in_flight(tmp_path):
    """`HipEngine.close()` with caller threads inside the batcher and batches on the device: searches already counted finish with
    their exact rows, later ones are refused (`RuntimeError: ... closed`), nothing hangs and nothing touches the freed handles."""
    import threading

    from oracle.flat_ip import flat_ip_topk
    from vod_amd import store
    from vod_amd.search.server import HipEngine

    rng = np.random.default_rng(13)
    n, d = 200_000, 64
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    for trial in range(3):
        engine = HipEngine(str(tmp_path / "v.npy"))
        answered, refused, wrong = [], [], []

        def loop(i):
            r = np.random.default_rng(100 * trial + i)
            try:
                while True:
                    q = r.integers(-8, 9, size=(int(r.integers(1, 300)), d)).astype(np.float32)
                    s, ids = engine.batcher.search(q, 20, client=i + 1)
                    answered.append((q, s, ids))
            except RuntimeError as exc:
                (refused if "closed" in str(exc) else wrong).append(str(exc))
            except Exception as exc:  # noqa: BLE001
                wrong.append(repr(exc))

        threads = [threading.Thread(target=loop, args=(i,)) for i in range(10)]
        for t in threads:
            t.start()
        time.sleep(0.15 + 0.1 * trial)
        batcher = engine.batcher
        batcher.close()          # drains the counted calls, refuses the rest, then frees the library handle
        for t in threads:
            t.join(timeout=60)
        assert not any(t.is_alive() for t in threads)
        engine.close()
        assert not wrong and len(refused) == 10 and len(answered) > 10
        for q, s, ids in answered[-12:]:
            rs, ri = flat_ip_topk(q, x, 20)
            np.testing.assert_array_equal(ids, ri)
            np.testing.assert_array_equal(s, rs)


def test_master_with_a_unix_domain_socket(tmp_path):
    """`HipMipsMaster(uds=True)`: the spawned server also listens on a Unix-domain socket, `get_client()` searches through it
    (SURVEY 8f-4's transport item), results equal the oracle; the socket file goes away with the server."""
    import os

    from oracle.flat_ip import flat_ip_topk
    from vod_amd import store
    from vod_amd.search.client import HipMipsMaster

    rng = np.random.default_rng(41)
    x = rng.integers(-6, 7, size=(20_000, 64)).astype(np.float32)
    q = rng.integers(-6, 7, size=(70, 64)).astype(np.float32)
    store.save_vectors(tmp_path / "v.npy", x, dtype=np.float16)
    with HipMipsMaster(tmp_path / "v.npy", port=-1, logging_level="warning", uds=True) as m:
        c = m.get_client()
        assert c.uds and os.path.exists(c.uds) and c.ping()
        import socket as _socket

        for binary in (False, True):
            for native in (True, False):  # libvodhip's client / the Python exchange, both over the socket file
                cl = type(c)(host=c.host, port=c.port, uds=c.uds, binary=binary, native=native)
                res = cl.search(vector=q, top_k=50)
                rs, ri = flat_ip_topk(q, x, 50)
                np.testing.assert_array_equal(res.indices, ri)
                np.testing.assert_array_equal(res.scores, rs)
                if native:
                    assert cl._local.native_h is not None and getattr(cl._local, "lean", None) is None
                else:
                    assert cl._local.lean.sock.family == _socket.AF_UNIX
        path = c.uds
    assert not os.path.exists(path)
