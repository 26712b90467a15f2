/*
 * syzgy_oracle.h -- CPU restatement of the SyzgyDB brute-force scan path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (syzgydb_amd/, include/) never links, imports or calls it.
 *
 * Parity pinning status: the reference is Go and there is no Go toolchain in
 * this image, so the reference itself cannot be run here.  The oracle is
 * pinned against every known-answer value the reference's own tests hold for
 * this path (collection_test.go:12-21, :549-612, dump_test.go round trips)
 * and the byte-level worked examples of SURVEY.md Appendix A
 * (tests/test_oracle_golden.py).  Everything beyond those KATs rests on this
 * file being a line-by-line restatement of the cited reference lines.
 *
 * Build: gcc -O2 -ffp-contract=off (Go on amd64 never fuses a*b+c).
 */
#ifndef SYZGY_ORACLE_H
#define SYZGY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_EUCLIDEAN 0 /* collection.go:186-189 */
#define ORC_COSINE 1

/* quantization.go:5-23 / :25-36 */
uint64_t orc_quantize(double value, int bits);
double orc_dequantize(uint64_t value, int bits);

/* collection.go:796-811; returns -1 where the reference panics */
int64_t orc_vector_size(int bits, int dim);

/* collection.go:713-743 (encodeDocument) / :768-794 (decodeVector) */
void orc_encode_vector(const double *vec, int dim, int bits, uint8_t *out);
void orc_decode_vector(const uint8_t *data, int dim, int bits, double *out);

/* collection.go:812-819 / :821-832 */
double orc_euclidean(const double *a, const double *b, int n);
double orc_angular(const double *a, const double *b, int n);

/* Go stdlib math.Acos (asin.go / atan.go, Cephes-derived), restated */
double orc_go_acos(double x);

/*
 * collection.go:569-711, Precision=="exact" branch with the `consider`
 * closure (:583-629) and the container/heap max-heap (:536-564).
 *
 * rows      n_rows x orc_vector_size() bytes, reference element encoding,
 *           in VISIT order (spanfile.go:521-560).
 * allow     NULL, or one byte per row: 0 = args.Filter returned false.
 * k, radius as SearchArgs.K / SearchArgs.Radius (radius > 0 wins over k).
 * out_*     caller buffers of `capacity` entries, filled in the order the
 *           reference returns Results (ascending by heap pops, :694-697).
 * returns   number of results the reference would return (may exceed
 *           capacity; only min(ret, capacity) entries are written, the best
 *           ones first), or -1 on bad arguments.
 * points_searched  receives pointsSearched (:589).
 */
int64_t orc_search_exact(const uint8_t *rows, uint64_t n_rows, int dim, int bits,
                         int metric, const double *query, int k, double radius,
                         const uint8_t *allow, uint64_t *out_rows, double *out_dist,
                         uint64_t capacity, uint64_t *points_searched);

/* distance of every row to the query, no selection (test helper) */
void orc_all_distances(const uint8_t *rows, uint64_t n_rows, int dim, int bits,
                       int metric, const double *query, double *out_dist);

/* distance for a list of rows only (full-size spot checks) */
void orc_distances_for_rows(const uint8_t *rows, const uint64_t *row_ids, uint64_t n_ids,
                            int dim, int bits, int metric, const double *query,
                            double *out_dist);

/*
 * Visit order of IterateSortedRecords (spanfile.go:540-560): sort.Strings over
 * the decimal spellings of the ids.  perm[i] = index into ids of the i-th
 * visited record.
 */
void orc_sorted_id_order(const uint64_t *ids, uint64_t n, uint64_t *perm);

/*
 * Synthetic data: counter-based splitmix64.  Element e of the
 * stream `seed` is U[-1,1) = (mix(seed + e) >> 11) * 2^-52 - 1.
 */
uint64_t orc_splitmix64(uint64_t x);
double orc_synth_value(uint64_t seed, uint64_t index);
void orc_synth_vectors(uint64_t seed, uint64_t first_row, uint64_t n_rows, int dim, double *out);
/* rows first_row..first_row+n_rows-1 of the synthetic corpus, encoded */
void orc_synth_rows(uint64_t seed, uint64_t first_row, uint64_t n_rows, int dim, int bits,
                    uint8_t *out);

/*
 * CPU baseline driver for bench.py: runs n_queries exact top-k searches over
 * the given rows with `threads` worker threads (one query per thread at a
 * time, mirroring concurrent Search calls under RLock, collection.go:570).
 * Returns wall seconds.
 */
double orc_bench_topk(const uint8_t *rows, uint64_t n_rows, int dim, int bits, int metric,
                      const double *queries, int n_queries, int k, int threads,
                      uint64_t *out_rows /* n_queries*k */);

/*
 * "Faithful" CPU baseline: the per-record work the reference does on EVERY visited
 * record of an exact search, not just the arithmetic -- getDocument -> ReadRecord ->
 * parseSpan with a CRC32-IEEE over the whole span (spanfile.go:730-818, :757),
 * stream walk, decodeVector into a freshly allocated []float64
 * (collection.go:470-484, :768-794), then distance and heap.  orc_spans_build
 * serialises the given rows as spans (spanfile.go:1-22 grammar, metadata stream of
 * meta_len bytes) into `out` and returns the bytes used (0 if out_cap is too small);
 * offsets[i] receives the start of record i's span.
 */
uint64_t orc_spans_build(const uint8_t *rows, uint64_t n_rows, int dim, int bits, int meta_len,
                         uint8_t *out, uint64_t out_cap, uint64_t *offsets);
/* One exact top-k search over those spans, single thread; returns wall seconds, or < 0
 * if a span fails its checksum / parse. */
double orc_bench_topk_faithful(const uint8_t *spans, const uint64_t *offsets, uint64_t n_rows, int dim,
                               int bits, int metric, const double *queries, int n_queries, int k,
                               uint64_t *out_rows /* n_queries*k */);

/*
 * LSH path: the reference's default search, Search{Precision:"medium"} ->
 * lshTree.search (lshtree.go:283-351) with consider() (collection.go:583-629) per candidate.
 * The forest is built by the reference's insert / split rules (lshtree.go:102-251) from a
 * documented splitmix64 stream in place of Go's math/rand (which cannot be reproduced here):
 * the forest is an input of the search, and parity of the search is defined on a given forest.
 * Documents are identified by their row.  PARITY UNPINNED beyond the restatement: the
 * reference's tests hold no known answers for this path: its one check of it,
 * TestCosineDistancePrecisionComparison (collection_test.go:23-103), asserts properties only (equal result
 * counts, distances within 100 % of the exact ones, PercentSearched < 100) -- restated in
 * tests/test_gpu_lsh.py::test_cosine_distance_precision_comparison_20000x3.
 */
typedef struct orc_lsh orc_lsh;
orc_lsh *orc_lsh_build(const uint8_t *rows /* borrowed until orc_lsh_free */, uint64_t n_rows, int dim, int bits,
                       int metric, int threshold /* 100, collection.go:292 */, int num_trees /* 5 */, uint64_t seed);
void orc_lsh_free(orc_lsh *t);
void orc_lsh_sizes(const orc_lsh *t, int64_t *n_nodes, int64_t *n_ids);
/* flat arrays: roots[num_trees], left/right[n_nodes] (-1 = leaf), normals[n_nodes*dim], b[n_nodes],
 * ids_off/ids_cnt[n_nodes] into ids[n_ids] */
void orc_lsh_export(const orc_lsh *t, int32_t *roots, int32_t *left, int32_t *right, double *normals, double *b,
                    int64_t *ids_off, int32_t *ids_cnt, uint64_t *ids);
int64_t orc_lsh_search(const orc_lsh *t, uint64_t n_rows, const double *query, int k, double radius,
                       const uint8_t *allow, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                       uint64_t *points_searched, uint64_t *visit_order /* nullable, n_rows */);

#ifdef __cplusplus
}
#endif
#endif
