"""`build_hip_mips_index`: counterpart of `vod_search.factory.build_faiss_index`
(/root/reference/src/vod_search/factory.py:131-190).

Same protocol: the store is content-addressed under `<cache_dir>/indices/<fingerprint>.npy`; only the
process with `skip_setup=False` (rank 0) writes it, every rank then passes `barrier_fn` and gets a
master bound to the same host/port (non-zero ranks construct it with `skip_setup=True` and only call
`get_client()`, src/vod_ops/workflows/train.py:67).
"""
from __future__ import annotations

import dataclasses
import logging
import pathlib
import typing as typ

import numpy as np

from vod_amd import store
from vod_amd.search.client import HipMipsMaster


@dataclasses.dataclass(frozen=True)
class HipMipsFactoryConfig:
    """The few fields of the reference's `FaissFactoryConfig` (src/vod_configs/search.py:124-154) this path needs."""

    factory: str = "Flat"           # only the exact index exists here
    metric: str = "inner_product"   # src/vod_configs/search.py:130
    dtype: str = "float16"          # HBM storage type of the SCAN copy (float16 | bfloat16)
    exact_f32: bool | None = None   # also keep the float32 rows and return the float32 brute-force result on the unrounded vectors and
                                    # queries - what the reference's faiss IndexFlat computes (build.py:65-73); the vector file handed
                                    # to the server is float32 then.  False: scores are dot products of the values rounded to `dtype`.
                                    # None (default) = decided by the vectors handed to `build_hip_mips_index`: True for float32 /
                                    # float64 vectors (the reference's: the drop-in returns the reference's result by default),
                                    # False for float16 vectors (nothing to keep: the scan copy IS the data)
    host: str = "http://localhost"
    port: int = -1                  # the reference config's default (src/vod_configs/search.py:134): < 0 = pick a free port, so two
                                    # default-config indexes on one host (hybrid set-ups, shards, two jobs) never collide; ONE
                                    # process resolves it and tells the others - see `resolve_port` / `broadcast_fn`
    logging_level: str = "CRITICAL"
    device: int = 0
    devices: tuple[int, ...] | None = None  # row-shard the store over these GPUs behind one address
    uds: bool = False               # also serve on a Unix-domain socket and hand its path to the clients (single-host jobs)
    group_backend: str = "nccl"     # with `devices`: "nccl" / "gloo" = one worker process per GPU on a process group; "node" = ONE
                                    # server process drives every GPU (vodhip_node_index: the reference server's own shape)
    http: str = "native"            # the server's HTTP shell: libvodhip's native front | "uvicorn" (FastAPI fallback)
    # Request fusion is ON by default (the DataLoader workers of every trainer rank each send their own small batch:
    # src/vod_dataloaders/realm_dataloader.py:92-118): concurrent requests share corpus scans, a lone request never waits.
    micro_batch_wait_ms: float = 0.0  # > 0: every batch additionally waits this long for company (a fixed window; not needed)
    batcher_params: tuple[tuple[str, int], ...] = ()  # vodhip_batcher_set_param pairs, e.g. (("grace_us", 0),) = never wait for
                                    # expected company, (("max_queries", 1024),) = smaller fused batches

    def fingerprint(self) -> dict:
        fp = {"factory": self.factory, "metric": self.metric, "dtype": self.dtype}
        if self.exact_f32:  # (absent when off / undecided: the cache names of existing fp16 stores do not change)
            fp["exact_f32"] = True
        return fp


def resolve_port(config: HipMipsFactoryConfig, broadcast_fn: None | typ.Callable[[int], int] = None) -> HipMipsFactoryConfig:
    """Turn `port < 0` into a concrete free port that EVERY rank agrees on.

    The reference does this in `_resolve_ports` (/root/reference/src/vod_search/factory.py:380-394): rank 0 picks a free
    port and `fabric.broadcast(port, 0)` hands it to the others, before any master is built.  `broadcast_fn(port) -> port`
    plays the fabric's role (e.g. `lambda p: fabric.broadcast(p, 0)`); without one the port is only valid in this process."""
    if config.port >= 0:
        return config
    from vod_amd.search.socket import find_available_port

    port = find_available_port()
    if broadcast_fn is not None:
        port = int(broadcast_fn(port))
    return dataclasses.replace(config, port=port)


logger = logging.getLogger(__name__)


def store_bytes(n_rows: int, dim: int, exact_f32: bool, n_devices: int = 1) -> int:
    """HBM bytes ONE device holds for its contiguous share of an `n_rows` x `dim` store (rows padded to 64 columns): 2 bytes per element
    of the scan copy; with `exact_f32` + 4 for the float32 plane + 8 per row for the row statistics of the error bound."""
    per_dev = -(-int(n_rows) // max(1, int(n_devices)))
    dim_pad = -(-int(dim) // 64) * 64
    return per_dev * dim_pad * (6 if exact_f32 else 2) + (per_dev * 8 if exact_f32 else 0)


def _free_device_bytes(devices: typ.Sequence[int]) -> "int | None":
    """The smallest free HBM among `devices`, or None when this process cannot tell (no torch / no visible GPU: the server decides)."""
    try:
        import torch

        if not torch.cuda.is_available():
            return None
        return min(int(torch.cuda.mem_get_info(int(d))[0]) for d in devices)
    except Exception:  # noqa: BLE001
        return None


def build_hip_mips_index(
    vectors: typ.Sequence[np.ndarray],
    *,
    config: HipMipsFactoryConfig | dict | None = None,
    cache_dir: str | pathlib.Path,
    skip_setup: bool = False,
    barrier_fn: None | typ.Callable[[str], None] = None,
    free_resources: bool = False,
    serve_on_gpu: bool = True,  # noqa: ARG001 - the reference's keyword (factory.py:139); this engine only exists on the GPU
    devices: None | typ.Sequence[int] = None,
    broadcast_fn: None | typ.Callable[[int], int] = None,
) -> HipMipsMaster:
    if config is None:
        config = HipMipsFactoryConfig()
    elif isinstance(config, dict):
        config = HipMipsFactoryConfig(**config)
    if config.port < 0:
        if skip_setup and broadcast_fn is None:
            # a rank that only connects cannot invent the port the serving rank picked (round-1 bug: every rank drew its own)
            raise ValueError("port < 0 with skip_setup=True: resolve the port once (resolve_port / broadcast_fn) and pass it to every rank")
        config = resolve_port(config, broadcast_fn)
    if devices is None and config.devices is not None:
        devices = list(config.devices)
    if config.exact_f32 is None:  # auto: float32 / float64 vectors are served exactly, float16 vectors as they are
        try:
            vec_dtype = np.dtype(getattr(vectors, "dtype", None) or np.asarray(vectors[0]).dtype) if len(vectors) else np.dtype(np.float16)
        except Exception:  # noqa: BLE001 - an exotic sequence: fall back to looking at one row
            vec_dtype = np.asarray(vectors[0]).dtype
        want_exact = bool(vec_dtype in (np.dtype(np.float32), np.dtype(np.float64)))
        if want_exact:
            # the float32 plane triples the store (2 -> 6 bytes per element + 8 per row): say so, and do not let the AUTO choice turn a store
            # that fits into one that fails at create (40 M x 1024: 82 GB -> 246 GB) - an explicit exact_f32=True is taken at its word
            n_rows, dim = len(vectors), int(np.asarray(vectors[0]).shape[-1]) if len(vectors) else 0
            n_dev = max(1, len(devices) if devices else 1)
            need, plain = store_bytes(n_rows, dim, True, n_dev), store_bytes(n_rows, dim, False, n_dev)
            free = _free_device_bytes(devices if devices else [config.device])
            if free is not None and need > 0.92 * free:
                logger.warning("exact_f32 (auto): the float32 plane needs %.1f GB per device, %.1f GB are free - serving the %s copy only "
                               "(%.1f GB; results carry its rounding).  Pass exact_f32=True to insist.", need / 1e9, free / 1e9, config.dtype, plain / 1e9)
                want_exact = False
            else:
                logger.info("exact_f32 (auto): float32 vectors are served exactly - %.1f GB per device instead of %.1f GB (exact_f32=False)",
                            need / 1e9, plain / 1e9)
        config = dataclasses.replace(config, exact_f32=want_exact)
    if config.factory != "Flat" or config.metric != "inner_product":
        raise ValueError("the HIP MIPS engine is an exact inner-product index (factory='Flat', metric='inner_product')")
    from vod_amd.zarr_store import ZarrVectors

    if isinstance(vectors, ZarrVectors):
        # the predict loop's tensorstore/zarr array is served as it lies: no float32 copy, no index file
        # (the reference re-reads it, casts, adds, writes and re-reads a faiss file: build.py:51-81, factory.py:153-173)
        path = vectors.path
    else:
        fp = store.fingerprint_vectors(vectors, config.fingerprint())
        path = pathlib.Path(cache_dir, "indices", f"{fp}.npy")
        path.parent.mkdir(parents=True, exist_ok=True)
    if not isinstance(vectors, ZarrVectors) and not skip_setup and not path.exists():
        tmp = path.with_suffix(".tmp.npy")
        store.save_vectors(tmp, vectors, dtype=np.float16 if (config.dtype == "float16" and not config.exact_f32) else np.float32)
        tmp.rename(path)
    if barrier_fn is not None:
        barrier_fn(f"hip mips store: `{path.name}`")
    if not path.exists():
        raise FileNotFoundError(f"Could not find the vector store at `{path}`.")
    return HipMipsMaster(
        vectors_path=path,
        logging_level=config.logging_level,
        host=config.host,
        port=config.port,
        skip_setup=skip_setup,
        free_resources=free_resources,
        dtype=config.dtype,
        exact_f32=bool(config.exact_f32),
        device=config.device,
        devices=None if devices is None else list(devices),
        group_backend=config.group_backend,
        uds=config.uds,
        http=config.http,
        micro_batch_wait_ms=config.micro_batch_wait_ms,
        batcher_params=dict(config.batcher_params),
    )
