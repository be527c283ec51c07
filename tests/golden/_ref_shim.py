"""Import shim used ONLY by `make_golden.py`, ONLY in the build container.

It lets the reference's *leaf* modules (merge / normalize / numpy_ops / sample /
in_batch_negatives / io / sharded_search / vod_gradients.retrieval / retrieval types)
be imported from the read-only checkout at /root/reference without executing the
package `__init__` files (which pull faiss, omegaconf, tensorstore, lightning ...).

Nothing in here is reference code: it only fabricates empty package shells and
identity stand-ins for the third-party decorators the leaf modules use
(`numba.njit`, `tenacity.retry`, `loguru.logger`).  The product, the GPU tests,
`bench.py` and `__graft_entry__.smoke()` never import this file.
"""
from __future__ import annotations

import importlib
import pathlib
import sys
import types

REF_SRC = pathlib.Path("/root/reference/src")


def _identity_decorator(*dargs, **dkwargs):
    # usable both as @njit and @njit(parallel=True, ...)
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn

    return wrap


class _Anything:
    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def _shell(name: str, path: pathlib.Path) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__path__ = [str(path)]  # type: ignore[attr-defined]
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


def install() -> dict[str, types.ModuleType]:
    """Install the stubs + package shells and import the reference leaf modules."""
    if not REF_SRC.exists():
        raise RuntimeError("/root/reference is not available: golden fixtures can only be regenerated in the build container")
    import numpy as np

    if not hasattr(np, "float_"):
        np.float_ = np.float64  # removed in NumPy 2; the reference's numpy_ops uses it in a TypeVar bound

    # --- third-party stand-ins -------------------------------------------------
    numba = types.ModuleType("numba")
    numba.njit = _identity_decorator
    numba.jit = _identity_decorator
    numba.prange = range
    numba.set_num_threads = lambda n: None
    typed = types.ModuleType("numba.typed")
    typed.List = list
    numba.typed = typed
    sys.modules["numba"] = numba
    sys.modules["numba.typed"] = typed

    tenacity = types.ModuleType("tenacity")
    tenacity.retry = _identity_decorator
    tenacity.stop_after_attempt = lambda *a, **k: None
    tenacity.wait_random_exponential = lambda *a, **k: None
    sys.modules["tenacity"] = tenacity

    loguru = types.ModuleType("loguru")
    loguru.logger = _Anything()
    sys.modules["loguru"] = loguru

    # --- bare package shells (their __init__ is never executed) -----------------
    vod_types = _shell("vod_types", REF_SRC / "vod_types")
    _shell("vod_dataloaders", REF_SRC / "vod_dataloaders")
    _shell("vod_dataloaders.core", REF_SRC / "vod_dataloaders" / "core")
    vod_search = _shell("vod_search", REF_SRC / "vod_search")
    _shell("vod_models", REF_SRC / "vod_models")
    _shell("vod_models.vod_gradients", REF_SRC / "vod_models" / "vod_gradients")
    _shell("vod_models.monitoring", REF_SRC / "vod_models" / "monitoring")

    out: dict[str, types.ModuleType] = {}
    retrieval = importlib.import_module("vod_types.retrieval")
    for n in ("RetrievalBatch", "RetrievalData", "RetrievalSample", "RetrievalTuple"):
        setattr(vod_types, n, getattr(retrieval, n))
    out["retrieval"] = retrieval
    try:
        batch = importlib.import_module("vod_types.batch")
        for n in ("RealmBatch", "RealmOutput", "Batch"):
            setattr(vod_types, n, getattr(batch, n))
        out["batch"] = batch
    except Exception as exc:  # pragma: no cover - depends on the container's pydantic
        out["batch_error"] = exc  # type: ignore[assignment]

    base = importlib.import_module("vod_search.base")
    vod_search.SearchClient = base.SearchClient
    vod_search.SearchMaster = base.SearchMaster
    out["base"] = base

    for short, full in {
        "numpy_ops": "vod_dataloaders.core.numpy_ops",
        "merge": "vod_dataloaders.core.merge",
        "normalize": "vod_dataloaders.core.normalize",
        "sample": "vod_dataloaders.core.sample",
        "in_batch_negatives": "vod_dataloaders.core.in_batch_negatives",
        "search": "vod_dataloaders.core.search",
        "io": "vod_search.io",
        "sharded_search": "vod_search.sharded_search",
        "functional": "vod_models.monitoring.functional",
    }.items():
        out[short] = importlib.import_module(full)
    if "batch" in out:
        importlib.import_module("vod_models.vod_gradients.base")
        out["gradients"] = importlib.import_module("vod_models.vod_gradients.retrieval")
    return out
