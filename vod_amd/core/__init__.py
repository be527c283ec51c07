"""Collate-side stages glued to the dense search: hybrid merge (HIP), search fan-out (host)."""
from vod_amd.core.merge import merge_hybrid, merge_hybrid_tensors  # noqa: F401
from vod_amd.core.search import LOOKUP_CLIENT_NAME, async_hybrid_search, merge_search_results  # noqa: F401
