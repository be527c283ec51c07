"""vod_amd -- MI355X-native dense-retrieval scoring path for VOD-style retrieval-augmented training.

Only the hot path is here: the HBM-resident corpus vector store, the fused
inner-product top-k search, the hybrid score merge and the in-batch retrieval loss, all as
hand-written gfx950 HIP kernels behind the C-ABI in `include/vodhip.h`, plus the host-side mirror
of the reference's `vod_search` client/master interface for that path.

There is NO CPU fallback: importing the compute entry points without the built
`vod_amd/csrc/libvodhip.so` raises `NativeLibraryError`.
"""
from vod_amd._native import NativeLibraryError, lib_path, load_library  # noqa: F401

__version__ = "0.1.0"
