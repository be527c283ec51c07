"""ctypes binding of libvodhip.so (the C-ABI declared in include/vodhip.h).

The library is built in-tree (`vod_amd/csrc/libvodhip.so`, see `vod_amd/build.py`) so that it travels
with the repository snapshot.  Loading fails loudly: there is no Python/CPU fallback.
"""
from __future__ import annotations

import ctypes
import pathlib
import threading

_LIB_NAME = "libvodhip.so"
_lock = threading.Lock()
_lib: ctypes.CDLL | None = None

F16, BF16, F32 = 0, 1, 2
EXACT_F32 = 0x100  # VODHIP_EXACT_F32: OR-ed into the store dtype at create
HOST, DEVICE = 0, 1
MAX_K = 2048
MAX_ENGINES = 4


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing or a native call failed."""


def lib_path() -> pathlib.Path:
    import os

    override = os.environ.get("VODHIP_LIB")  # A/B runs of two builds in one process environment (still a libvodhip build)
    if override:
        return pathlib.Path(override)
    return pathlib.Path(__file__).resolve().parent / "csrc" / _LIB_NAME


_c = ctypes
_vp, _i64, _i32, _f32p = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p

# name -> (restype, argtypes): every symbol include/vodhip.h declares
SIGNATURES: dict[str, tuple] = {
    "vodhip_last_error": (_c.c_char_p, []),
    "vodhip_version": (_i32, []),
    "vodhip_index_create": (_i32, [_i32, _i64, _i32, _i64, _c.POINTER(_vp)]),
    "vodhip_index_destroy": (_i32, [_vp]),
    "vodhip_index_add": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "vodhip_index_reset": (_i32, [_vp]),
    "vodhip_index_ntotal": (_i32, [_vp, _c.POINTER(_i64)]),
    "vodhip_index_dim": (_i32, [_vp, _c.POINTER(_i64)]),
    "vodhip_index_capacity": (_i32, [_vp, _c.POINTER(_i64)]),
    "vodhip_index_data": (_i32, [_vp, _c.POINTER(_vp), _c.POINTER(_i64), _c.POINTER(_i32)]),
    "vodhip_index_get_rows": (_i32, [_vp, _i64, _i64, _vp, _i32, _vp]),
    "vodhip_index_data_f32": (_i32, [_vp, _c.POINTER(_vp), _c.POINTER(_i64)]),
    "vodhip_index_get_rows_f32": (_i32, [_vp, _i64, _i64, _vp, _i32, _vp]),
    "vodhip_index_search": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp]),
    "vodhip_index_search_async": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp]),
    "vodhip_index_search_finish": (_i32, [_vp, _vp]),
    "vodhip_index_set_row_labels": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "vodhip_index_set_query_labels": (_i32, [_vp, _vp, _i32]),
    "vodhip_index_set_param": (_i32, [_vp, _c.c_char_p, _i64]),
    "vodhip_debug_schedule": (_i32, [_i64, _i32, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _c.POINTER(_i64), _i32]),
    "vodhip_index_get_stat": (_i32, [_vp, _c.c_char_p, _c.POINTER(_i64)]),
    "vodhip_debug_tile_order": (_i32, [_i64, _c.POINTER(_i64), _c.POINTER(_i64)]),
    "vodhip_debug_read_probe": (_i32, [_i32, _c.POINTER(_i64), _i32]),
    "vodhip_node_index_create": (_i32, [_i32, _c.POINTER(_i32), _i64, _i32, _i64, _c.POINTER(_vp)]),
    "vodhip_node_index_destroy": (_i32, [_vp]),
    "vodhip_node_index_add": (_i32, [_vp, _vp, _i64, _i32]),
    "vodhip_node_index_reset": (_i32, [_vp]),
    "vodhip_node_index_ntotal": (_i32, [_vp, _c.POINTER(_i64)]),
    "vodhip_node_index_n_shards": (_i32, [_vp]),
    "vodhip_node_index_shard": (_i32, [_vp, _i32, _c.POINTER(_vp), _c.POINTER(_i64), _c.POINTER(_i32)]),
    "vodhip_node_index_set_param": (_i32, [_vp, _c.c_char_p, _i64]),
    "vodhip_node_index_get_stat": (_i32, [_vp, _c.c_char_p, _c.POINTER(_i64)]),
    "vodhip_node_index_peer_access": (_i32, [_vp, _c.POINTER(_i32), _i32]),
    "vodhip_node_index_set_row_labels": (_i32, [_vp, _vp, _i64]),
    "vodhip_node_index_set_query_labels": (_i32, [_vp, _vp, _i32, _i32]),
    "vodhip_node_index_search": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "vodhip_node_index_search_async": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "vodhip_node_index_search_finish": (_i32, [_vp]),
    "vodhip_merge_topk": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "vodhip_merge_topk_strided": (_i32, [_vp, _i64, _vp, _i64, _i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "vodhip_merge_hybrid": (
        _i32,
        [_vp, _vp, _i32, _i32, _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_i32), _c.POINTER(_c.c_float), _i64,
         _vp, _vp, _vp, _c.POINTER(_vp), _i32, _vp, _vp, _vp],
    ),
    "vodhip_retrieval_forward": (
        _i32, [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    ),
    "vodhip_retrieval_forward_aux": (
        _i32, [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _c.c_float, _c.c_float, _c.c_float,
               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    ),
    "vodhip_priority_sample": (
        _i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _c.c_float, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    ),
    "vodhip_priority_sample_merged": (
        _i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _c.c_float, _i32, _i32,
               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    ),
    "vodhip_collate": (_i32, [_vp, _vp]),
    "vodhip_flatten_inbatch": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vodhip_retrieval_backward": (_i32, [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "vodhip_gather_by_id": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vodhip_b64url_encode": (_i64, [_vp, _i64, _vp, _i64, _vp]),
    "vodhip_b64url_decode": (_i64, [_vp, _i64, _vp]),
    # H6 serving (vodhip_serve.hip, vodhip_http.hip)
    "vodhip_batcher_create": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _c.POINTER(_vp)]),
    "vodhip_batcher_destroy": (_i32, [_vp]),
    "vodhip_batcher_set_param": (_i32, [_vp, _c.c_char_p, _i64]),
    "vodhip_batcher_get_stat": (_i32, [_vp, _c.c_char_p, _c.POINTER(_i64)]),
    "vodhip_batcher_search": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _i32, _c.c_uint64, _vp, _vp]),
    "vodhip_batcher_forget_client": (_i32, [_vp, _c.c_uint64]),
    "vodhip_client_create": (_i32, [_c.c_char_p, _i32, _c.c_char_p, _c.POINTER(_vp)]),
    "vodhip_client_destroy": (_i32, [_vp]),
    "vodhip_client_search": (_i32, [_vp, _vp, _i32, _i64, _i64, _i32, _i32, _c.c_double, _vp, _vp]),
    "vodhip_client_last_body": (_c.c_char_p, [_vp]),
    "vodhip_http_reply_set": (_i32, [_vp, _i32, _c.c_char_p, _vp, _i64, _c.c_char_p]),
    "vodhip_http_create": (_i32, [_vp, _i64, _vp, _vp, _i64, _c.POINTER(_vp)]),
    "vodhip_http_listen_tcp": (_i32, [_vp, _c.c_char_p, _i32]),
    "vodhip_http_listen_unix": (_i32, [_vp, _c.c_char_p]),
    "vodhip_http_start": (_i32, [_vp]),
    "vodhip_http_stop": (_i32, [_vp]),
    "vodhip_http_destroy": (_i32, [_vp]),
    "vodhip_http_get_stat": (_i32, [_vp, _c.c_char_p, _c.POINTER(_i64)]),
    "vodhip_wire_npy_header": (_i64, [_i32, _i64, _i64, _vp, _i64]),
    "vodhip_wire_parse_npy": (_i32, [_vp, _i64, _c.POINTER(_i32), _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64)]),
    "vodhip_wire_parse_fast_search": (_i32, [_vp, _i64, _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64)]),
    "vodhip_wire_fast_search_reply": (_i64, [_vp, _vp, _i64, _i32, _vp, _i64]),
}

# callback types of the serving layer (include/vodhip.h: vodhip_search_fn, vodhip_http_fallback_fn)
SEARCH_FN = _c.CFUNCTYPE(_i32, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp)
HTTP_FALLBACK_FN = _c.CFUNCTYPE(None, _vp, _c.c_char_p, _c.c_char_p, _vp, _i64, _c.c_uint64, _vp)


class CollateArgs(ctypes.Structure):
    """`vodhip_collate_args_t` (include/vodhip.h), field for field."""

    _fields_ = [
        ("lookup_idx", _vp), ("lookup_lbl", _vp),
        ("engine_idx", _vp * MAX_ENGINES), ("engine_scr", _vp * MAX_ENGINES),
        ("engine_weight", _c.c_float * MAX_ENGINES), ("engine_k", _c.c_int32 * MAX_ENGINES),
        ("k_lookup", _c.c_int32), ("n_engines", _c.c_int32),
        ("nq", _i64),
        ("noise", _vp), ("noise_stride", _i64),
        ("k_positive", _c.c_int32), ("k_total", _c.c_int32), ("max_support_size", _c.c_int32), ("in_batch_negatives", _c.c_int32),
        ("temperature", _c.c_float), ("flags", _c.c_int32),
        ("merged_idx", _vp), ("merged_lbl", _vp), ("merged_scr", _vp), ("merged_raw", _vp * MAX_ENGINES), ("row_cursor", _vp),
        ("out_local", _vp), ("out_ids", _vp), ("out_scores", _vp), ("out_log_weights", _vp), ("out_labels", _vp),
        ("out_raw", _vp * MAX_ENGINES), ("out_lse_pos", _vp), ("out_lse_neg", _vp), ("out_max_sampling_id", _vp),
        ("flat_ids", _vp), ("flat_scores", _vp), ("flat_log_weights", _vp), ("flat_labels", _vp), ("flat_raw", _vp * MAX_ENGINES),
        ("flat_n_unique", _vp),
    ]


def load_library() -> ctypes.CDLL:
    """Load libvodhip.so once; raise NativeLibraryError if it is absent or lacks a declared symbol."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not path.exists():
            raise NativeLibraryError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C vod_amd/csrc`.  vod_amd has no CPU fallback."
            )
        # PyTorch-ROCm ships its own libamdhip64.so.7 + libhsa-runtime64; libvodhip.so needs the same SONAME.  Whichever is
        # loaded first serves both, and the system runtime loaded FIRST next to torch's HSA runtime finds no device
        # ("no ROCm-capable device is detected"): make sure torch's copy is the one in the process.
        import torch  # noqa: F401

        try:
            lib = ctypes.CDLL(str(path))
        except OSError as exc:  # pragma: no cover - depends on the machine
            raise NativeLibraryError(f"cannot load {path}: {exc}") from exc
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise NativeLibraryError(f"{path} does not export `{name}` (stale build?)") from exc
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(status: int) -> None:
    """Raise with the library's thread-local message when a call returned an error."""
    if status != 0:
        msg = load_library().vodhip_last_error()
        raise NativeLibraryError(msg.decode("utf-8", "replace") if msg else f"native call failed with status {status}")


def torch_dtype_code(dtype) -> int:
    import torch

    try:
        return {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}[dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {dtype}; expected float16, bfloat16 or float32") from None


def numpy_dtype_code(dtype) -> int:
    import numpy as np

    dt = np.dtype(dtype)
    if dt == np.float16:
        return F16
    if dt == np.float32:
        return F32
    raise TypeError(f"unsupported numpy dtype {dt}; expected float16 or float32")


def current_stream_ptr(device=None) -> int:
    import torch

    if device is None:
        return int(torch.cuda.current_stream().cuda_stream)
    idx = device if isinstance(device, int) else device.index
    try:  # the raw handle without building a Stream object (this sits on the launch path of latency-bound kernels)
        return int(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx))
    except AttributeError:  # pragma: no cover - private API absent
        return int(torch.cuda.current_stream(device).cuda_stream)
