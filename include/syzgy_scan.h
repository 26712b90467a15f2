/*
 * syzgy_scan.h -- C ABI of the MI355X-native brute-force scan that drops in
 * behind SyzgyDB's Collection.Search (Precision "exact").
 *
 * The reference (smhanov/syzgydb, Go) has no FFI seam today; the seam this
 * library replaces is the block
 *     collection.go:672-684   (IterateRecords + consider(), the HOT LOOP)
 * together with the per-record work it calls:
 *     collection.go:583-629   consider(): filter, distance, top-k / radius
 *     collection.go:768-794   decodeVector     quantization.go:25-36 dequantize
 *     collection.go:812-832   euclideanDistance / angularDistance
 *     collection.go:536-564   resultPriorityQueue (container/heap)
 * The Go side keeps SearchArgs / SearchResults / SearchResult
 * (collection.go:115-158) unchanged; go/syzgy_gpu.go shows the cgo binding and
 * INTEGRATION.md the edit to Collection.Search.
 *
 * Conventions: extern "C", plain pointers and explicit sizes, no exceptions
 * cross the boundary.  Every function returns SZG_OK (0) or a negative
 * SZG_E_* code.  All in/out buffers are caller-owned and only borrowed for the
 * duration of the call (cgo rule: no Go pointer is retained).  A handle may
 * be used by any number of threads concurrently for szg_search_* (the
 * reference runs Searches concurrently under RLock, collection.go:570);
 * load/append/tombstone/overwrite need exclusive access (the reference's
 * write lock, collection.go:428, :512).
 *
 * Rows are addressed by their position in the VISIT ORDER the caller loaded
 * them in (spanfile.go:521-560); the caller keeps the row -> document-id table.
 */
#ifndef SYZGY_SCAN_H
#define SYZGY_SCAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SZG_ABI_VERSION 4

/* DistanceMethod, collection.go:186-189 */
#define SZG_EUCLIDEAN 0
#define SZG_COSINE 1

#define SZG_OK 0
#define SZG_E_INVALID (-1)    /* bad argument (dim, quantization, metric, k, null pointer) */
#define SZG_E_NOMEM (-2)      /* host or device allocation failed */
#define SZG_E_DEVICE (-3)     /* HIP runtime error; szg_last_error() has the text */
#define SZG_E_TRUNCATED (-4)  /* radius search: more hits than `capacity`; *out_total says how many */
#define SZG_E_NODEVICE (-5)   /* no usable gfx950 device / HIP kernels not loadable */
#define SZG_E_RANGE (-6)      /* row index out of range */
#define SZG_E_UNSUPPORTED (-7)/* valid in the reference but outside this build's limits */

typedef struct szg_index szg_index;

/*
 * One handle per Collection (created where NewCollection pages the corpus,
 * collection.go:297-311; destroyed in Collection.Close, :408-421).
 *   dim         CollectionOptions.DimensionCount
 *   quant_bits  CollectionOptions.Quantization: 4, 8, 16, 32 or 64 (collection.go:796-811)
 *   metric      SZG_EUCLIDEAN or SZG_COSINE (collection.go:275-283)
 *   devices     HIP device ordinals to shard the rows over (contiguous row
 *               ranges, one per entry); NULL / n_devices==0 = current device.
 */
int szg_index_create(szg_index **out, int dim, int quant_bits, int metric,
                     const int *devices, int n_devices);
void szg_index_destroy(szg_index *ix);

/* Row byte size = getVectorSize(quant_bits, dim), collection.go:796-811; <0 on error. */
int64_t szg_row_bytes(int quant_bits, int dim);

/*
 * Replace the mirror's content with n_rows packed vectors in the reference's
 * on-disk element encoding (stream 1 of each span: 4-bit high-nibble-first,
 * 8-bit bytes, big-endian 16/32/64-bit; collection.go:713-743).  Copies.
 */
int szg_index_load(szg_index *ix, const uint8_t *rows, uint64_t n_rows);

/* AddDocument (collection.go:427-457): rows go to the end of the visit order. */
int szg_index_append(szg_index *ix, const uint8_t *rows, uint64_t n_rows);

/*
 * AddDocument for a block of float64 vectors (bulk ingest): quantize + pack on the
 * device exactly as encodeDocument / quantize do (collection.go:713-743,
 * quantization.go:5-23), rows go to the end of the visit order.
 */
int szg_index_append_f64(szg_index *ix, const double *vectors, uint64_t n_rows);

/* AddDocument on an existing id rewrites the record: replace one row in place
 * (from packed bytes, or from a float64 vector encoded on the device). */
int szg_index_overwrite_f64(szg_index *ix, uint64_t row, const double *vector);
int szg_index_overwrite(szg_index *ix, uint64_t row, const uint8_t *row_bytes);

/* removeDocument (collection.go:511-521): the row is skipped by every later scan. */
int szg_index_tombstone(szg_index *ix, uint64_t row);

/* rows loaded (tombstoned ones included) / rows still live */
uint64_t szg_index_rows(const szg_index *ix);
uint64_t szg_index_live_rows(const szg_index *ix);

/* Read rows back in the reference encoding (inverse of the page-in transform). */
int szg_index_read_rows(szg_index *ix, uint64_t first_row, uint64_t n_rows, uint8_t *out);

/*
 * Exact top-k, i.e. Search{Precision:"exact", K:k, Radius:0}
 * (collection.go:606-619, :672-684, :694-697) for n_queries independent
 * queries.
 *   queries     n_queries x dim float64, SearchArgs.Vector (never quantized)
 *   allow_bits  NULL, or n_queries(!) x ceil(rows/64) words: bit r of query
 *               q's mask = args.Filter(id(r), metadata(r)) (collection.go:592)
 *   out_rows    n_queries x k row indices, ascending distance (:694-697)
 *   out_dist    n_queries x k float64 distances, the reference's own values
 *   out_count   n_queries result counts (<= k; fewer when fewer rows qualify)
 * Unused tail entries of out_rows are UINT64_MAX.
 */
int szg_search_topk(szg_index *ix, const double *queries, int n_queries, int k,
                    const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist,
                    int32_t *out_count);

/*
 * Radius search, i.e. Search{Precision:"exact", Radius:radius>0} (K ignored,
 * collection.go:598-605).  Writes the min(total, capacity) closest hits in
 * ascending distance; *out_total is the full hit count.  Returns
 * SZG_E_TRUNCATED when total > capacity (retry with a larger buffer).
 */
int szg_search_radius(szg_index *ix, const double *query, double radius,
                      const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist,
                      uint64_t capacity, uint64_t *out_total);

/*
 * Radius searches for a batch of queries, each with its own radius: the collect sweeps of the batch are walked
 * query-major by one launch per 16 queries, several launches in flight (collection.go:598-605 per query).
 *   allow_bits   NULL, or n_queries x ceil(rows/64) words
 *   out_offsets  n_queries + 1 entries: the hits of query i are out_rows / out_dist [out_offsets[i], out_offsets[i+1]),
 *                ascending distance
 * When the hits do not fit `capacity` the call returns SZG_E_TRUNCATED with out_offsets complete (so
 * out_offsets[n_queries] is the capacity to retry with) and the first `capacity` entries written.
 */
int szg_search_radius_batch(szg_index *ix, const double *queries, int n_queries, const double *radii,
                            const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                            uint64_t *out_offsets);

/*
 * The reference's float64 distance (c.distance, collection.go:596, :812-832) from
 * one query to each listed row, bit-identical to the reference: the gather-by-row
 * primitive for re-ranking candidates of the LSH path (lshtree.go:283-351 calls
 * consider() per candidate id) or any other candidate generator.
 */
int szg_distances(szg_index *ix, const double *query, const uint64_t *rows, uint64_t n_rows,
                  double *out_dist);

/*
 * c.distance(doc_a.Vector, doc_b.Vector) for stored rows: the primitive under
 * computeAverageDistance (collection.go:348-400), which the Go side keeps (it owns the
 * math/rand pair selection and the in-order float64 sum, :372-398).  out_dist[i] is the
 * reference's float64 distance between the decoded rows rows_a[i] and rows_b[i].
 */
int szg_pair_distances(szg_index *ix, const uint64_t *rows_a, const uint64_t *rows_b, uint64_t n_pairs,
                       double *out_dist);

/*
 * Cross-shard result assembly for one-process-per-GPU sharding (host code, no
 * device work).  Every rank answers the batch on its own row range with
 * szg_search_topk asking for list_len = k+1 results (rows made global with
 * szg_index_set_row_base); the per-rank lists are exchanged with one RCCL
 * all-gather and merged here by replaying consider()'s top-k branch
 * (collection.go:606-619) over their union in visit order.
 *   rows/dist  [n_lists][n_queries][list_len], counts [n_lists][n_queries]
 *   out_*      [n_queries][k] (+ out_count[n_queries])
 *   out_history_dependent  nullable, [n_queries]: 1 when two of the best k+1
 *              distances are equal or NaN, i.e. the reference's answer depends
 *              on its whole heap history and a single-handle search over the
 *              unsharded corpus would take the exact-replay path.
 */
int szg_merge_topk(int k, int n_lists, int list_len, int n_queries, const uint64_t *rows,
                   const double *dist, const int32_t *counts, uint64_t *out_rows,
                   double *out_dist, int32_t *out_count, uint8_t *out_history_dependent);
/* The same merge straight from the exchanged buffer, no repacking on the caller's side:
 * records[n_lists][n_queries][2*list_len + 1] int64 = list_len rows | list_len float64 bit
 * patterns | count, i.e. what each rank contributes to the all-gather. */
int szg_merge_topk_records(int k, int n_lists, int list_len, int n_queries, const int64_t *records,
                           uint64_t *out_rows, double *out_dist, int32_t *out_count,
                           uint8_t *out_history_dependent);

/* ---- one process per GPU: the exchange inside the library ----------------- */

/*
 * The loop collection.go:672-684 visits independent records, so the rows shard over the GPUs of a node in
 * contiguous ranges, one process (and one handle) per GPU; szg_index_set_row_base makes the rows a rank returns
 * global.  A sharded search runs the rank's own exact search on its range, exchanges the per-rank results with ONE
 * all-gather per micro-batch -- RCCL (ncclAllGather over xGMI) on the rank's device -- and replays the reference's
 * selection over the union on every rank, so every rank returns the single-collection answer.
 *
 * Set-up: rank 0 calls szg_comm_unique_id and hands the 128 bytes to the other ranks by whatever means the host
 * has (a file, a socket, its own RPC); then EVERY rank calls szg_comm_create (collective: ncclCommInitRank) and
 * attaches the communicator to its handle.  All sharded calls are collective too: every rank makes the same
 * calls, in the same order, with the same queries.  One communicator serves any number of handles of the process.
 */
typedef struct szg_comm szg_comm;
#define SZG_COMM_ID_BYTES 128
int szg_comm_unique_id(uint8_t *id /* [SZG_COMM_ID_BYTES] */);
int szg_comm_create(szg_comm **out, const uint8_t *id, int rank, int world, int device);
/* The same with the host's own transport instead of RCCL: fn all-gathers bytes_per_rank bytes from every rank's
 * `send` into `recv` ([world][bytes_per_rank], rank order) and returns 0.  Host memory only, no device needed
 * (hosts with their own fabric; the tests: gloo on CPU, several ranks on one card). */
typedef int (*szg_allgather_fn)(void *user, const void *send, void *recv, uint64_t bytes_per_rank);
int szg_comm_create_host(szg_comm **out, szg_allgather_fn fn, void *user, int rank, int world);
void szg_comm_destroy(szg_comm *c);
/* Size the exchange staging for micro-batches of up to n_queries queries at this k ahead of time (it grows on
 * demand otherwise -- a hipHostMalloc inside the first call that needs it).  COLLECTIVE like the searches: every
 * rank calls it with the same arguments; the ranks confirm to each other that all of them hold the staging (one
 * status word over the staging that already exists), so a rank that cannot allocate makes EVERY rank return
 * SZG_E_NOMEM here instead of leaving its peers waiting in a later exchange. */
int szg_comm_reserve(szg_comm *c, int n_queries, int k);
/* The handle's sharded searches go through `c` (borrowed: destroy it after the handle, or attach NULL first). */
int szg_index_attach_comm(szg_index *ix, szg_comm *c);

/*
 * szg_search_topk over the sharded collection.  allow_bits covers THIS rank's rows (n_queries x ceil(local rows / 64)
 * words).  More than 128 queries are pipelined in micro-batches of 256: a worker thread sweeps the next one while
 * this thread exchanges and merges.
 * out_history_dependent (nullable, [n_queries]): 1 when two of the best k+1 merged distances are equal or NaN, i.e.
 * the reference's order depends on its whole heap history, which no single rank holds.  With tie_mode 0 (default)
 * such a query is then answered EXACTLY as the unsharded collection would: the heap travels rank 0 -> 1 -> ... in
 * visit order, each rank replaying consider() over its own rows (G more small all-gathers for the flagged queries
 * of the call together), so N > 1 returns what N = 1 returns, order included.
 *
 * Failure semantics of every collective call (this one, szg_search_radius_sharded, szg_comm_merge_*): a rank whose
 * own search or allocation fails still enters every exchange of the call and marks its contribution, so EVERY rank
 * returns an error and none waits for a peer that never comes.  The one case left to the host's own timeout is a
 * transport failure part-way through an exchange (HIP / RCCL error between enqueue and wait on one rank).
 */
int szg_search_topk_sharded(szg_index *ix, const double *queries, int n_queries, int k, const uint64_t *allow_bits,
                            uint64_t *out_rows, double *out_dist, int32_t *out_count, uint8_t *out_history_dependent);
/* szg_search_radius_batch over the sharded collection: an all-gather of the hit counts, one padded all-gather of
 * (row, distance) records, the reference's push-all / pop-all heap over the union in row order.
 * `capacity` may differ between ranks: SZG_E_TRUNCATED (out_offsets complete) is a LOCAL condition -- fetch the
 * answer again with szg_comm_last_radius and a buffer of out_offsets[n_queries] entries; never repeat the
 * collective call on the truncated ranks only. */
int szg_search_radius_sharded(szg_index *ix, const double *queries, int n_queries, const double *radii,
                              const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                              uint64_t *out_offsets);
/*
 * The exchange-and-merge halves on their own (what the two calls above do after the rank's own search): the rank's
 * exact top-(k+1) lists  rows / dist [n_queries][k+1], counts [n_queries]  (rows global)  ->  out_* [n_queries][k];
 * the rank's radius hits in CSR form (offsets [n_queries + 1], rows global, ascending distance per query) -> the
 * merged CSR.  Pure host code plus the transport.
 */
int szg_comm_merge_topk(szg_comm *c, int k, int n_queries, const uint64_t *rows, const double *dist,
                        const int32_t *counts, uint64_t *out_rows, double *out_dist, int32_t *out_count,
                        uint8_t *out_history_dependent);
int szg_comm_merge_radius(szg_comm *c, int n_queries, const uint64_t *offsets, const uint64_t *rows,
                          const double *dist, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                          uint64_t *out_offsets);
/* The merged answer of this communicator's LAST radius call again (not collective): what a rank whose buffer was too
 * small calls after SZG_E_TRUNCATED. */
int szg_comm_last_radius(szg_comm *c, int n_queries, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                         uint64_t *out_offsets);
/*
 * The heap chain on its own (what szg_search_topk_sharded runs for its flagged queries): `replay` continues the
 * reference's heap -- container/heap's array, heap_n entries of (row, distance), element for element -- over THIS
 * rank's rows of flagged query j in visit order (consider()'s top-k branch, collection.go:606-619) and returns 0.
 * Collective: rank g replays in round g.  out_* [n_flagged][k].
 */
typedef int (*szg_replay_fn)(void *user, int j, int k, uint64_t *heap_rows, double *heap_dist, int32_t *heap_n);
int szg_comm_chain_topk(szg_comm *c, int k, int n_flagged, szg_replay_fn replay, void *user, uint64_t *out_rows,
                        double *out_dist, int32_t *out_count);

typedef struct szg_comm_stats {
    uint64_t exchanges;   /* data all-gathers of the searches (ONE per top-k micro-batch, two per radius batch) */
    double exchange_us;   /* wall time inside all all-gathers (copies + collective + wait) */
    double host_us;       /* packing the records and merging the gathered lists */
    int rccl_ranks;       /* ncclCommCount of the communicator (0: host transport) */
    int zero_copy;        /* RCCL: the collective reads / writes the pinned host staging itself (no H2D / D2H copies) */
    uint64_t chained_replays; /* top-k queries whose merged answer held equal distances and was settled by the
                                 rank-to-rank heap chain (the reference's order, collection.go:606-619) */
    uint64_t status_rounds;   /* one-word all-gathers in which the ranks agreed on their staging (only when it grows) */
    uint64_t chain_rounds;    /* all-gathers of the heap chain (world per call that has flagged queries) */
} szg_comm_stats;
int szg_comm_get_stats(szg_comm *c, szg_comm_stats *out);
int szg_comm_reset_stats(szg_comm *c);

/* ---- diagnostics -------------------------------------------------------- */

const char *szg_strerror(int code);
/* Text of the last error raised on the calling thread ("" if none). */
const char *szg_last_error(void);
int szg_abi_version(void);

typedef struct szg_stats {
    uint64_t queries;          /* top-k + radius queries served */
    uint64_t scan_launches;    /* launches of the fused scan kernel */
    uint64_t escalations;      /* top-k queries whose first pass could not be certified */
    uint64_t scan_bytes;       /* algorithmic bytes swept: sum over launches of rows x row_bytes */
    double scan_ms;            /* HIP-event time of the scan kernel launches (timing on) */
    double total_ms;           /* HIP-event time of the whole per-query pipeline (timing on) */
    uint64_t timed_launches;   /* launches included in scan_ms */
    uint64_t full_replays;     /* top-k queries answered by the exact full-history replay
                                  (equal distances or NaN among the best k+1 candidates) */
    uint64_t mq_launches;      /* shared (multi-query) sweeps; each is also one scan launch */
    uint64_t mq_queries;       /* queries answered through shared sweeps */
    uint64_t mq_fallbacks;     /* shared-sweep batches redone through the score matrix (candidate buffer overflow) */
    /* host time of szg_search_topk outside the waits for results (always measured): query
     * preparation (swizzle / digit planes, first-k rows), the HIP calls that enqueue a batch
     * (copies, launches, events: these can block on a full queue), and result assembly
     * (gather, certification, heap replay) */
    double host_prep_us;
    double host_finish_us;
    double host_enqueue_us;
    uint64_t sketch_queries;   /* top-k queries answered through the 8-bit sketch pre-pass ("sketch" option) */
    uint64_t sketch_fallbacks; /* ... that it could not settle and handed to the full-precision path */
    uint64_t mq_bf16_sweeps;   /* shared sweeps that ran on the bfloat16 matrix cores (64- / 32- / 16-bit rows, top-k batches
                                  on 8-bit rows in whole 64-byte steps; their candidates are re-scored in float32 and
                                  re-ranked in float64 like every other path's) */
} szg_stats;

/* HIP-event timing on the library's own streams (off by default): 1 = events around the scan
 * launches (scan_ms / timed_launches), 2 = also around each batch's whole pipeline (total_ms). */
int szg_set_timing(szg_index *ix, int enabled);
int szg_get_stats(szg_index *ix, szg_stats *out);
int szg_reset_stats(szg_index *ix);

/*
 * Tunables (name, default, meaning): the sixteen a deployment could want.  All are safe to change between calls.
 * (Rounds 1-3 exposed another fifteen -- ring depths, waves per CU, stream placement, sweep kinds per row width --
 * whose values measurement settled; they are compile-time constants now, csrc/scan_internal.h, and A/B runs go
 * through `make variant`.)
 *
 *   answer semantics
 *     tie_mode            0   when two of the best k+1 distances are exactly equal, or one is NaN, the reference's
 *                             output depends on its whole heap history, so the query is re-answered by an exact
 *                             replay over every row; 1 = keep the fast answer (a valid top-k whose order among equal
 *                             distances may differ from the reference's)
 *     slack               16  extra candidates kept beyond k (at least; k/2 when larger)
 *   one sweep per query
 *     queries_per_launch  16  sweeps one scan launch walks back to back (query-major): no launch gap or chip-wide
 *                             tail between the sweeps of a batch
 *     query_batch         16  queries staged, merged, re-ranked and copied back together
 *     mask_dense          1   sweeps whose filter / tombstone masks pass at least half the rows read every row and
 *                             apply the masks at the row finish; selective masks (and 0) compact the row steps that
 *                             hold a passing row first
 *     serialize_scans     1   sweeps of one shard never overlap each other (every sweep has the whole HBM bandwidth)
 *     contexts            4   batches in flight per shard
 *   sketch pre-pass (float32 rows)
 *     sketch              0   1 = keep an 8-bit sketch of every row (+25 % memory, built on the device at the first
 *                             search after a load, kept up to date across appends / overwrites / tombstones) and
 *                             answer one-query-per-sweep searches by sweeping the sketch (a quarter of the bytes) for
 *                             k + sketch_extra candidates, re-ranking those on the float32 rows in float64 and
 *                             certifying with the triangle inequality of the reference's distance; unsettled queries
 *                             take the full sweep.  Same answers.  1M x 768 cosine k=10: 2.2 k -> 8.0 k queries/s
 *     sketch_extra        30  candidates beyond k; the pre-pass serves k + sketch_extra <= 64
 *     sketch_min_rows     4096  collections below this size always take the full sweep
 *   shared sweeps
 *     multi_query         1   batches of >= mq_min queries share ONE sweep of the corpus, the dot products on the
 *                             matrix cores: 64- / 32- / 16-bit rows on bfloat16 roundings (v_mfma_f32_16x16x32_bf16,
 *                             up to 96 queries per pass; candidates scored again in float32, certified against the
 *                             bfloat16 bound) -- and top-k batches of more than 48 queries on 8-bit rows in whole 64-byte steps, whose codes
 *                             are exact in bfloat16 (only the query is rounded) --, 4-bit rows, the other 8-bit
 *                             shapes and every radius batch on 8-bit rows in exact integer arithmetic
 *                             (v_mfma_i32_16x16x64_i8, 48 per pass, two passes per launch); 0 = one sweep per query
 *     mq_min              2   smallest batch worth a shared sweep
 *     mq_hits             1024 candidates per query the threshold from the prefix pass aims at
 *     coalesce            1   concurrent szg_search_topk calls with ONE query each -- the reference's Searches under
 *                             RLock -- are answered together, up to 96 per shared sweep, by whichever caller finds no
 *                             batch in flight; concurrent szg_search_radius callers likewise
 *     radius_mq           1   radius batches of two or more queries share ONE sweep of the corpus per up to 96
 *                             queries (the radius is the collect threshold); 0 = one collect sweep per query
 *     finish_thread       1   a call of three or more shared-sweep batches assembles its finished batches on a
 *                             second host thread while the caller's prepares and enqueues the next ones
 *   test hooks (paths that data takes by itself only rarely): force_escalate, force_matrix (the score-matrix form of
 *   the shared sweeps: small shards, candidate-buffer overflow), force_no_refine (their tail as separate launches: kp > 256)
 */
int szg_set_option(szg_index *ix, const char *name, int64_t value);

/* ---- bench / test utilities (not part of the drop-in surface) ----------- */

/*
 * Fill the mirror with n_rows synthetic vectors generated on the device:
 * element e of row r = U[-1,1) from splitmix64(seed + (first_row + r)*dim + e),
 * quantized and packed exactly as encodeDocument would (collection.go:713-743,
 * quantization.go:5-23).  Byte-identical to oracle/orc_synth_rows.
 */
int szg_index_synth(szg_index *ix, uint64_t n_rows, uint64_t seed, uint64_t first_row);

/* Global row index of this handle's row 0 (multi-process sharding); rows
 * returned by szg_search_* are local + base. */
int szg_index_set_row_base(szg_index *ix, uint64_t base);

/* Test hook: what = 1: the next `value` growths of this rank's exchange staging fail (SZG_E_NOMEM), as an allocation
 * failure on ONE rank of a job would. */
int szg_comm_debug_inject(szg_comm *c, int what, int value);

#ifdef __cplusplus
}
#endif
#endif /* SYZGY_SCAN_H */
