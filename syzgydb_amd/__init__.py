"""syzgydb_amd -- MI355X-native brute-force vector scan behind SyzgyDB's
Collection.Search.  The compute lives in libsyzgy_scan.so
(hand-written HIP for gfx950, C ABI in include/syzgy_scan.h); this package is
the host-side mirror of the reference's API for that path.
"""
from ._lib import (SZG_COSINE, SZG_EUCLIDEAN, SzgError, LIB_PATH)  # noqa: F401
from .index import ScanIndex, pack_allow_bits, f64_probe  # noqa: F401
from .collection import (Collection, CollectionOptions, Document, SearchArgs, SearchResult,  # noqa: F401
                         SearchResults, Euclidean, Cosine)
from . import codec  # noqa: F401
from . import lsh  # noqa: F401
from .pager import SpanfilePager  # noqa: F401

__all__ = ["ScanIndex", "Collection", "CollectionOptions", "Document", "SearchArgs",
           "SearchResult", "SearchResults", "Euclidean", "Cosine", "codec", "SzgError",
           "pack_allow_bits", "f64_probe", "SpanfilePager", "lsh", "SZG_COSINE", "SZG_EUCLIDEAN", "LIB_PATH"]
