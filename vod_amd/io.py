"""Wire codec of the search server: NumPy array <-> urlsafe-base64 of its `.npy` bytes.

String-compatible with the reference (/root/reference/src/vod_search/io.py:17-32): the payload is exactly
`base64.urlsafe_b64encode(np.save(...))`.  Decoding refuses pickled object arrays (the reference passes
`allow_pickle=True`; nothing on this path needs it and it would execute untrusted bytes).

Same bytes, fewer passes: for a 1024 x 768 float32 batch (3 MB) `np.save` into a BytesIO, `getvalue()`,
`urlsafe_b64encode` (encode + translate) and `np.load` of a second BytesIO cost 8-9 ms per direction; the
`.npy` header is written/parsed here and the array bytes go through `binascii` once (~3 ms per direction).
`json_body` / `parse_json_body` let client and server skip `json.dumps` / `json.loads` over the 4 MB string
(the payload alphabet needs no escaping) while staying valid JSON for any other peer.
"""
from __future__ import annotations

import base64
import binascii
import ctypes
import io
import json

import numpy as np

_TO_URLSAFE = bytes.maketrans(b"+/", b"-_")
_FROM_URLSAFE = bytes.maketrans(b"-_", b"+/")
_lib_state: list = []  # [lib or None] once probed


def _codec_lib():
    """libvodhip's host-side base64 loops (3 MB in ~1.5 ms instead of ~7); `binascii` when the library cannot be
    loaded in this process (a client box without ROCm).  This is the codec only - no search runs without the library."""
    if not _lib_state:
        try:
            from vod_amd import _native

            _lib_state.append(_native.load_library())
        except Exception:
            _lib_state.append(None)
    return _lib_state[0]


def _b64url_encode(head: bytes, arr: np.ndarray | None) -> str:
    lib = _codec_lib()
    n_data = 0 if arr is None else arr.nbytes
    if lib is None:
        raw = head if arr is None else head + arr.tobytes()
        return binascii.b2a_base64(raw, newline=False).translate(_TO_URLSAFE).decode("ascii")
    out = bytearray(4 * ((len(head) + n_data + 2) // 3))
    dst = (ctypes.c_char * len(out)).from_buffer(out)
    n = lib.vodhip_b64url_encode(head, len(head), arr.ctypes.data if n_data else None, n_data, ctypes.addressof(dst))
    if n != len(out):
        raise RuntimeError("base64 encoder returned an unexpected length")
    del dst
    return out.decode("ascii")


def _b64url_decode(data: bytes) -> np.ndarray:
    """Decoded bytes as a (writable, owned) uint8 array."""
    lib = _codec_lib()
    if lib is not None:
        out = np.empty(3 * len(data) // 4 + 3, dtype=np.uint8)
        n = lib.vodhip_b64url_decode(data, len(data), out.ctypes.data)
        if n >= 0:
            return out[:n]
    return np.frombuffer(bytearray(binascii.a2b_base64(data.translate(_FROM_URLSAFE))), dtype=np.uint8)


def serialize_np_array(array: np.ndarray) -> str:
    arr = np.asarray(array)
    if arr.dtype.hasobject or not arr.flags.c_contiguous or arr.ndim == 0:
        buf = io.BytesIO()  # uncommon layouts: let NumPy decide (Fortran order flag, 0-d arrays, refusal of objects)
        np.save(buf, arr, allow_pickle=False)
        return _b64url_encode(buf.getvalue(), None)
    head = io.BytesIO()
    np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(arr))
    return _b64url_encode(head.getvalue(), arr)


def deserialize_np_array(encoded: str | bytes, *, dtype=None, writable: bool = True) -> np.ndarray:  # noqa: ARG001
    """Inverse of `serialize_np_array` (`writable` is kept for callers; the result always owns writable memory)."""
    data = encoded.encode("ascii") if isinstance(encoded, str) else bytes(encoded)
    raw = _b64url_decode(data)
    arr = None
    if raw[:8].tobytes() == b"\x93NUMPY\x01\x00":
        head = io.BytesIO(raw[:65546].tobytes())
        head.seek(8)
        shape, fortran, dt = np.lib.format.read_array_header_1_0(head)
        if not dt.hasobject and not fortran:
            count = int(np.prod(shape, dtype=np.int64))
            off = head.tell()
            if off + count * dt.itemsize > raw.size:
                raise ValueError("truncated .npy payload")
            arr = raw[off : off + count * dt.itemsize].view(dt).reshape(shape) if off % dt.itemsize == 0 or dt.itemsize == 1 \
                else np.frombuffer(raw[off : off + count * dt.itemsize].tobytes(), dtype=dt).reshape(shape).copy()
    if arr is None:
        arr = np.load(io.BytesIO(raw.tobytes()), allow_pickle=False)
    if dtype is not None:
        arr = arr.astype(dtype)
    return arr


def json_body(fields: dict[str, str], extra: dict | None = None) -> bytes:
    """`{"k": "<payload>", ..., **extra}` as JSON bytes without running the payload strings through an encoder."""
    parts = [json.dumps(k) + ': "' + v + '"' for k, v in fields.items()]
    parts += [json.dumps(k) + ": " + json.dumps(v) for k, v in (extra or {}).items()]
    return ("{" + ", ".join(parts) + "}").encode("ascii")


def parse_json_body(body: bytes, payload_keys: tuple[str, ...]) -> dict:
    """`json.loads` for bodies whose big fields are escape-free strings: those are sliced out, the rest is parsed."""
    out: dict = {}
    rest = body
    for key in payload_keys:
        tag = b'"' + key.encode("ascii") + b'"'
        i = rest.find(tag)
        if i < 0:
            continue
        j = rest.find(b'"', rest.find(b":", i + len(tag)) + 1)
        k = rest.find(b'"', j + 1)
        if j < 0 or k < 0 or b"\\" in rest[j + 1 : k]:
            return json.loads(body)
        out[key] = rest[j + 1 : k].decode("ascii")
        rest = rest[:j] + b'""' + rest[k + 1 :]
    small = json.loads(rest)
    if not isinstance(small, dict):
        raise ValueError("expected a JSON object")
    small.update(out)
    return small


# ---- buffer-level codec (round 3): the same bytes on the wire without intermediate str / bytes copies ----------------------
# A 1024 x 768 float32 query batch is 4.2 MB of base64.  `serialize_np_array` -> str -> `json_body` -> bytes copies it three
# times on the way out, `parse_json_body` -> str -> bytes -> decode three times on the way in; at ~1 GB/s per Python-level copy
# that is most of the reference format's per-request floor once the search itself takes a millisecond.  These helpers write the
# base64 text straight into the request / response buffer and decode it straight out of the received body.
def _addr(buf, offset: int = 0) -> int:
    """Address of `buf[offset]` for a bytes / bytearray / contiguous memoryview (the object must outlive the call)."""
    if isinstance(buf, bytes):
        return ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p).value + offset
    mv = memoryview(buf)
    if mv.readonly:
        return ctypes.cast(ctypes.c_char_p(mv.obj if isinstance(mv.obj, bytes) else bytes(mv)), ctypes.c_void_p).value + offset
    return ctypes.addressof((ctypes.c_char * mv.nbytes).from_buffer(mv)) + offset


def npy_header(arr: np.ndarray) -> bytes:
    head = io.BytesIO()
    np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(arr))
    return head.getvalue()


def b64_len(n_bytes: int) -> int:
    return 4 * ((n_bytes + 2) // 3)


def json_body_with_arrays(arrays: dict[str, np.ndarray], extra: dict | None = None) -> bytearray:
    """`{"name": "<urlsafe-b64(np.save(array))>", ..., **extra}` - byte for byte what `json_body` of `serialize_np_array` strings
    gives - assembled in ONE buffer: the encoder writes each array's base64 text in place."""
    lib = _codec_lib()
    items = []
    for name, a in arrays.items():
        arr = np.asarray(a)
        if lib is None or arr.dtype.hasobject or not arr.flags.c_contiguous or arr.ndim == 0:
            return bytearray(json_body({k: serialize_np_array(v) for k, v in arrays.items()}, extra))
        items.append((json.dumps(name).encode("ascii") + b': "', npy_header(arr), arr))
    tail = "".join(", " + json.dumps(k) + ": " + json.dumps(v) for k, v in (extra or {}).items()).encode("ascii") + b"}"
    total = 1 + sum(len(p) + b64_len(len(h) + arr.nbytes) + 1 for p, h, arr in items) + 2 * (len(items) - 1) + len(tail)
    out = bytearray(total)
    base = ctypes.addressof((ctypes.c_char * total).from_buffer(out))
    pos = 0
    out[pos : pos + 1] = b"{"
    pos += 1
    for i, (prefix, head, arr) in enumerate(items):
        if i:
            out[pos : pos + 2] = b", "
            pos += 2
        out[pos : pos + len(prefix)] = prefix
        pos += len(prefix)
        n = lib.vodhip_b64url_encode(head, len(head), arr.ctypes.data if arr.nbytes else None, arr.nbytes, base + pos)
        if n != b64_len(len(head) + arr.nbytes):
            raise RuntimeError("base64 encoder returned an unexpected length")
        pos += n
        out[pos : pos + 1] = b'"'
        pos += 1
    out[pos : pos + len(tail)] = tail
    assert pos + len(tail) == total
    return out


def find_payload_spans(body, payload_keys: tuple[str, ...]) -> tuple[dict, dict[str, tuple[int, int]]]:
    """Split a JSON object body into (the small fields, parsed) and the [start, end) byte spans of the big escape-free string
    fields named in `payload_keys` - nothing is copied.  Falls back to a full `json.loads` when a payload is not a plain string."""
    mv = body if isinstance(body, (bytes, bytearray)) else bytes(body)
    spans: dict[str, tuple[int, int]] = {}
    pieces = []
    cursor = 0
    order = []
    for key in payload_keys:
        tag = b'"' + key.encode("ascii") + b'"'
        i = mv.find(tag)
        if i < 0:
            continue
        colon = mv.find(b":", i + len(tag))
        j = mv.find(b'"', colon + 1)
        k = mv.find(b'"', j + 1)
        # the shortcut only holds when the match IS the top-level key followed by a plain string: anything else - the text occurring
        # twice (e.g. inside `subset_ids`), something between key and colon, a non-string value, an escape - goes through `json.loads`
        twice = k >= 0 and mv.find(tag, k + 1) >= 0  # (the payload itself is base64 text: no quote, so only the rest is searched)
        if twice or colon < 0 or j < 0 or k < 0 or mv[i + len(tag) : colon].strip() or mv[colon + 1 : j].strip() or mv.find(b"\\", j + 1, k) >= 0:
            full = json.loads(bytes(mv))
            if not isinstance(full, dict):
                raise ValueError("expected a JSON object")
            return full, {}
        order.append((j + 1, k, key))
    order.sort()
    for start, end, key in order:  # the small remainder: everything but the payload characters
        pieces.append(bytes(mv[cursor:start]))
        cursor = end
        spans[key] = (start, end)
    pieces.append(bytes(mv[cursor:]))
    small = json.loads(b"".join(pieces))
    if not isinstance(small, dict):
        raise ValueError("expected a JSON object")
    return small, spans


_NPY_DTYPES = {2: np.dtype("<f4"), 0: np.dtype("<f2"), 3: np.dtype("<i8")}  # vodhip_wire_parse_npy's dtype codes


def _npy_layout(raw: np.ndarray):
    """(dtype, shape, data offset) of `.npy` bytes held in a uint8 array - by libvodhip's header parser for the layouts this service
    exchanges (version 1.0, C order, 2-D, float32 / float16 / int64: ~1 us instead of NumPy's ~25 us `literal_eval` of the header
    dict), by NumPy's own reader for everything else.  None: not something a view can be taken of (objects, Fortran order)."""
    lib = _codec_lib()
    if lib is not None:
        dt, rows, cols, off = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        if lib.vodhip_wire_parse_npy(raw.ctypes.data, raw.size, ctypes.byref(dt), ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(off)) == 0:
            return _NPY_DTYPES[dt.value], (rows.value, cols.value), off.value
    if raw[:8].tobytes() != b"\x93NUMPY\x01\x00":
        return None
    head = io.BytesIO(raw[:65546].tobytes())
    head.seek(8)
    shape, fortran, dt_ = np.lib.format.read_array_header_1_0(head)
    if dt_.hasobject or fortran:
        return None
    return dt_, shape, head.tell()


def deserialize_np_array_span(body, start: int, end: int) -> np.ndarray:
    """`deserialize_np_array(body[start:end])` without materialising the slice: base64 text -> owned, writable array."""
    lib = _codec_lib()
    if lib is None:
        return deserialize_np_array(bytes(body[start:end]))
    n_in = end - start
    raw = np.empty(3 * n_in // 4 + 3, dtype=np.uint8)
    keep = body if isinstance(body, (bytes, bytearray)) else bytes(body)
    n = lib.vodhip_b64url_decode(_addr(keep, start), n_in, raw.ctypes.data)
    if n < 0:
        return deserialize_np_array(bytes(body[start:end]))  # lenient decoder (embedded newlines, ...)
    raw = raw[:n]
    layout = _npy_layout(raw)
    if layout is not None:
        dt, shape, off = layout
        count = int(np.prod(shape, dtype=np.int64))
        if off + count * dt.itemsize > raw.size:
            raise ValueError("truncated .npy payload")
        if off % dt.itemsize == 0 or dt.itemsize == 1:
            return raw[off : off + count * dt.itemsize].view(dt).reshape(shape)
        return np.frombuffer(raw[off : off + count * dt.itemsize].tobytes(), dtype=dt).reshape(shape).copy()
    return np.load(io.BytesIO(raw.tobytes()), allow_pickle=False)


def load_npy_view(body) -> np.ndarray:
    """`np.load` of raw `.npy` bytes as a VIEW into `body` where the layout allows it (C order, no objects, aligned data):
    a 3 MB query batch is not copied once more on its way to the device.  Anything else goes through `np.load`."""
    mv = memoryview(body)
    if mv.nbytes >= 10:
        layout = _npy_layout(np.frombuffer(mv, dtype=np.uint8))
        if layout is not None:
            dt, shape, off = layout
            count = int(np.prod(shape, dtype=np.int64))
            if off + count * dt.itemsize <= mv.nbytes and off % dt.itemsize == 0:
                return np.frombuffer(mv, dtype=dt, count=count, offset=off).reshape(shape)
    return np.load(io.BytesIO(bytes(mv)), allow_pickle=False)
