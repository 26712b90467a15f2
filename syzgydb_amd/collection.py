"""Host-side mirror of the reference's Collection API for the search path.

Same names, argument meaning and error behaviour as collection.go:
CollectionOptions (:31-48), Document (:100-110), SearchArgs (:140-158),
SearchResult/SearchResults (:115-135), Collection.AddDocument (:427-457),
GetDocument (:463-484), UpdateDocument (:490-509), removeDocument (:511-521),
Search (:569-711).  What differs is only WHERE the exact scan runs: the HOT
LOOP (:672-684) is one call through the C ABI into the HIP library.

Storage, the LSH index, REST, and the filter language stay in the reference
and are out of scope here: documents live in host memory, and any Precision
is answered by the exact scan (the reference's LSH path is unchanged Go code).
"""
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import codec
from .index import ScanIndex, pack_allow_bits

Euclidean = 0  # collection.go:186-189
Cosine = 1


@dataclass
class CollectionOptions:
    """collection.go:31-48."""
    Name: str = ""
    DistanceMethod: int = Euclidean
    DimensionCount: int = 0
    Quantization: int = 0  # 0 -> 64, as NewCollection does (collection.go:254-256)


@dataclass
class Document:
    ID: int
    Vector: np.ndarray
    Metadata: bytes


@dataclass
class SearchResult:
    ID: int
    Metadata: bytes
    Distance: float = 0.0


@dataclass
class SearchResults:
    Results: List[SearchResult] = field(default_factory=list)
    PercentSearched: float = 0.0


@dataclass
class SearchArgs:
    """collection.go:140-158."""
    Vector: Optional[np.ndarray] = None
    Filter: Optional[Callable[[int, bytes], bool]] = None
    K: int = 0
    Radius: float = 0.0
    Offset: int = 0
    Limit: int = 0
    Precision: str = ""
    FilterKey: Optional[str] = None  # not in the reference: names the filter for the bitmask cache


class Collection:
    def __init__(self, options: CollectionOptions, devices=None, sketch=False, strict_order=True):
        if options.Quantization == 0:
            options.Quantization = 64  # collection.go:254-256
        if options.DistanceMethod not in (Euclidean, Cosine):
            raise ValueError("unsupported distance method")  # collection.go:282
        codec.vector_size(options.Quantization, options.DimensionCount)  # panics like :809
        self.options = options
        self.DimensionCount = options.DimensionCount
        self.Quantization = options.Quantization
        self.DistanceMethod = options.DistanceMethod
        self._index = ScanIndex(self.DimensionCount, self.Quantization, self.DistanceMethod,
                                devices=devices)
        if sketch and self.Quantization == 32:
            # 8-bit sketch pre-pass for lone Searches on float32 collections (same answers, +25 % device memory)
            self._index.set_option("sketch", 1)
        self._row_of = {}    # id -> row
        self._id_of = []     # row -> id (None once tombstoned)
        self._meta = []      # row -> metadata bytes
        self._closed = False
        # filter -> allow-bitmask cache: a filter's verdicts only change when the
        # collection does, so they are kept per (filter key, collection version)
        self._version = 0
        self._mask_cache = {}
        # Row order only matters where the reference's own answer depends on its visit order: ties at the k
        # boundary, equal distances in a radius result (collection.go:608).  Production iterates a Go map (random
        # order per run, spanfile.go:525): every order is a reference order, rows are simply appended
        # (strict_order=False).  The reference's deterministic mode (seeded tests, spanfile.go:522) pins the order
        # to sort.Strings of the decimal ids; that is what the parity tests compare with, hence the default.  A new
        # id is ALWAYS appended in place; one that sorts before the last ("10" < "9": ordinary ascending integers
        # do that) only marks the order stale, and the mirror is re-paged in order the first time an answer
        # actually contains such a tie -- not on every add (the same rule as go/syzgy_gpu.go).
        self.strict_order = bool(strict_order)
        self._last_idstr = ""
        self._order_stale = False
        self.resorts = 0

    @classmethod
    def from_spanfile(cls, path, devices=None):
        """Open an existing SyzgyDB collection file: the header record's options
        override the caller's (collection.go:241-252), every record's packed vector
        is paged into HBM in IterateSortedRecords order (collection.go:297-311)."""
        from .pager import SpanfilePager
        with SpanfilePager(path) as pg:
            c = cls(CollectionOptions(Name=str(path), DistanceMethod=pg.metric,
                                      DimensionCount=pg.dim, Quantization=pg.quant_bits),
                    devices=devices)
            pg.load_into(c._index)
            for row, id in enumerate(pg.ids()):
                c._row_of[int(id)] = row
                c._id_of.append(int(id))
                c._meta.append(pg.metadata(row))
                c._last_idstr = str(int(id))
        return c

    def _note_appended(self, id):
        """Row order bookkeeping for a new id appended at the end of the mirror."""
        s = str(id)
        if s < self._last_idstr:
            self._order_stale = self.strict_order
        else:
            self._last_idstr = s

    def _resort(self):
        """Re-page the live rows in sorted decimal-string id order (reload of the mirror)."""
        live = sorted((str(id), id, row) for id, row in self._row_of.items())
        data = self._index.read_rows(0, self._index.rows) if self._index.rows else None
        rows = [row for _, _, row in live]
        new_meta = [self._meta[r] for r in rows]
        self._index.load(data[rows] if rows else np.zeros((0, self._index.row_bytes), np.uint8))
        self._id_of = [id for _, id, _ in live]
        self._meta = new_meta
        self._row_of = {id: i for i, id in enumerate(self._id_of)}
        self._last_idstr = live[-1][0] if live else ""
        self._order_stale = False
        self.resorts += 1
        self._version += 1  # rows are renumbered: cached filter masks no longer apply

    # -- CRUD (host bookkeeping + mirror maintenance) ---------------------------
    def AddDocument(self, id: int, vector, metadata: bytes = b""):
        v = np.asarray(vector, dtype=np.float64).reshape(-1)
        if v.size != self.DimensionCount:
            # log.Panicf in the reference (collection.go:432-434)
            raise ValueError("vector size does not match the expected number of dimensions: "
                             "expected %d, got %d" % (self.DimensionCount, v.size))
        row_bytes = codec.encode_rows(v.reshape(1, -1), self.Quantization)
        id = int(id)
        self._version += 1
        if id in self._row_of:  # WriteRecord of an existing id replaces the record
            row = self._row_of[id]
            self._index.overwrite(row, row_bytes)
            self._meta[row] = bytes(metadata)
        else:
            self._index.append(row_bytes)
            self._row_of[id] = len(self._id_of)
            self._id_of.append(id)
            self._meta.append(bytes(metadata))
            self._note_appended(id)

    def AddDocuments(self, ids, vectors, metadatas=None):
        """Bulk ingest (not in the reference; same effect as AddDocument in a loop for new ids)."""
        V = np.atleast_2d(np.asarray(vectors, dtype=np.float64))
        if V.shape[1] != self.DimensionCount:
            raise ValueError("vector size does not match the expected number of dimensions")
        ids = [int(i) for i in ids]
        self._version += 1
        if any(i in self._row_of for i in ids) or len(set(ids)) != len(ids):
            for j, i in enumerate(ids):
                self.AddDocument(i, V[j], metadatas[j] if metadatas else b"")
            return
        self._index.append_vectors(V)  # quantize + pack on the device (szg_index_append_f64)
        for j, i in enumerate(ids):
            self._row_of[i] = len(self._id_of)
            self._id_of.append(i)
            self._meta.append(bytes(metadatas[j]) if metadatas else b"")
            self._note_appended(i)

    def GetDocument(self, id: int) -> Document:
        row = self._row_of.get(int(id))
        if row is None:
            raise KeyError("record not found")  # spanfile.go:516
        data = self._index.read_rows(row, 1)
        vec = codec.decode_rows(data, self.DimensionCount, self.Quantization)[0]
        return Document(ID=int(id), Vector=vec, Metadata=self._meta[row])

    def UpdateDocument(self, id: int, new_metadata: bytes):
        row = self._row_of.get(int(id))
        if row is None:
            raise KeyError("record not found")
        self._version += 1
        self._meta[row] = bytes(new_metadata)

    def removeDocument(self, id: int):
        row = self._row_of.pop(int(id), None)
        if row is None:
            raise KeyError("record not found")
        self._version += 1
        self._index.tombstone(row)
        self._id_of[row] = None
        self._meta[row] = b""

    def GetDocumentCount(self) -> int:
        return len(self._row_of)

    def GetAllIDs(self):
        return sorted(self._row_of)

    def computeAverageDistance(self, samples: int, intn=None) -> float:
        """collection.go:348-400: mean distance over up to `samples` random pairs of
        distinct documents.  `intn(n)` plays rand.Intn (Go's generator is the caller's);
        the pairs' distances come from one szg_pair_distances call, the in-order float64
        sum and the division stay here as in the reference (:389-398)."""
        if samples <= 0:
            return 0.0
        ids = self.GetAllIDs()
        if len(ids) < 2:
            return 0.0
        if intn is None:
            import random
            intn = random.randrange
        a, b = [], []
        for _ in range(samples):
            id1 = ids[intn(len(ids))]
            id2 = ids[intn(len(ids))]
            if id1 == id2:
                continue  # :375-377
            a.append(self._row_of[id1])
            b.append(self._row_of[id2])
        if not a:
            return 0.0
        total = 0.0
        for d in self._index.pair_distances(a, b):
            total += float(d)
        return total / float(len(a))

    def Close(self):
        if not self._closed:
            self._index.close()
            self._closed = True

    # -- Search ---------------------------------------------------------------
    def _allow_mask(self, flt, key=None):
        """One bit per row from the caller's filter (collection.go:592-594).  `key` (e.g. the
        REST layer's filter text) makes the verdicts reusable until the collection changes;
        without it the filter object itself is the key."""
        if flt is None:
            return None
        ck = (key if key is not None else id(flt), self._version)
        hit = self._mask_cache.get(ck)
        if hit is not None and (key is not None or hit[1] is flt):
            return hit[0]
        mask = np.zeros(len(self._id_of), dtype=bool)
        for row, id_ in enumerate(self._id_of):
            if id_ is not None:
                mask[row] = bool(flt(id_, self._meta[row]))
        bits = pack_allow_bits(mask)
        if len(self._mask_cache) > 16:
            self._mask_cache.clear()
        self._mask_cache[ck] = (bits, flt)
        return bits

    def Search(self, args: SearchArgs) -> SearchResults:
        """collection.go:569-711."""
        n_records = len(self._row_of)
        results: List[SearchResult] = []
        points_searched = 0

        if args.Radius == 0 and args.K == 0:
            # listing mode (collection.go:633-669): sorted *string* id order
            for id in sorted(self._row_of, key=lambda i: str(i)):
                row = self._row_of[id]
                if args.Filter is not None and not args.Filter(id, self._meta[row]):
                    continue
                points_searched += 1
                if args.Offset > 0 and points_searched <= args.Offset:
                    continue
                results.append(SearchResult(ID=id, Metadata=self._meta[row]))
                if args.Limit > 0 and len(results) >= args.Limit:
                    break
        else:
            q = np.asarray(args.Vector, dtype=np.float64).reshape(-1)
            if q.size != self.DimensionCount:
                # undefined in the reference (collection.go:814, :823); rejected here
                raise ValueError("query length %d != dimension %d" % (q.size, self.DimensionCount))

            def scan():
                allow = self._allow_mask(args.Filter, getattr(args, "FilterKey", None)) if n_records else None
                if n_records == 0:
                    return [], []
                if args.Radius > 0:  # K is ignored (collection.go:598-605)
                    return self._index.search_radius(q, args.Radius, allow=allow)
                r, d, c = self._index.search_topk(q, args.K, allow=allow)
                return r[0, : c[0]], d[0, : c[0]]
            if self._order_stale and self._index.options.get("tie_mode", 0) != 0:
                # tie_mode 1 keeps the fast answer for ties and counts no replay: the test below would never see one,
                # so the deterministic mode re-pages up front
                self._resort()
            if self._order_stale:
                # rows out of sort.Strings order: did the answer depend on the visit order?  (the library re-answers
                # ties among the best k+1 by an exact replay in ROW order and counts it; a radius result depends on
                # the order when two of its distances are equal)
                before = self._index.stats()["full_replays"]
                rows, dist = scan()
                tie = self._index.stats()["full_replays"] != before or \
                    (args.Radius > 0 and len(set(float(x) for x in dist)) != len(dist))
                if tie:
                    self._resort()
                    rows, dist = scan()
            else:
                rows, dist = scan()
            for row, dd in zip(rows, dist):
                row = int(row)
                results.append(SearchResult(ID=self._id_of[row], Metadata=self._meta[row],
                                            Distance=float(dd)))
            points_searched = n_records  # counted before the filter (collection.go:589)

        percent = float(points_searched) / float(n_records) * 100 if n_records else 0.0
        return SearchResults(Results=results, PercentSearched=percent)

    def SearchBatch(self, args_list) -> List[SearchResults]:
        """Exact Searches of a caller that holds many queries (not in the reference, whose REST endpoint is
        single-query, rest.go:371-487): top-k Searches with one K share ONE szg_search_topk call -- and with it
        sweeps on the matrix cores --, radius Searches one szg_search_radius_batch; each query keeps its own
        Filter.  Anything else (listing mode, mixed kinds) is answered Search by Search.  Same results as Search."""
        args_list = list(args_list)
        n_records = len(self._row_of)
        if not args_list:
            return []
        radius = args_list[0].Radius > 0
        same = all((a.Radius > 0) == radius and (radius or (a.K == args_list[0].K and a.K > 0)) for a in args_list)
        if not same or n_records == 0:
            return [self.Search(a) for a in args_list]
        if self._order_stale:
            self._resort()   # (a batch is not worth the tie bookkeeping: deterministic mode re-pages first)
        Q = np.stack([np.asarray(a.Vector, dtype=np.float64).reshape(-1) for a in args_list])
        if Q.shape[1] != self.DimensionCount:
            raise ValueError("query length %d != dimension %d" % (Q.shape[1], self.DimensionCount))
        allow = None
        if any(a.Filter is not None for a in args_list):
            words = (len(self._id_of) + 63) // 64
            allow = np.full((len(args_list), words), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
            for i, a in enumerate(args_list):
                if a.Filter is not None:
                    allow[i] = self._allow_mask(a.Filter, getattr(a, "FilterKey", None)).reshape(-1)
        out = []
        if radius:
            hits = self._index.search_radius_batch(Q, [a.Radius for a in args_list], allow=allow)
        else:
            r, d, c = self._index.search_topk(Q, args_list[0].K, allow=allow)
            hits = [(r[i, : c[i]], d[i, : c[i]]) for i in range(len(args_list))]
        for rows, dist in hits:
            res = [SearchResult(ID=self._id_of[int(row)], Metadata=self._meta[int(row)], Distance=float(dd))
                   for row, dd in zip(rows, dist)]
            out.append(SearchResults(Results=res, PercentSearched=100.0))
        return out
