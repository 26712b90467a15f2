// scan_internal.h -- types and helpers shared by the translation units that implement
// include/syzgy_scan.h (the C ABI).  Not installed; nothing here crosses the boundary.
//
//   api_common.cpp    error text, container/heap replay, cross-shard merge (host only)
//   scan_query.cpp    query preparation (swizzle / digit planes) and the key error bounds
//   scan_handle.cpp   handle lifetime, contexts, mutations, options, statistics
//   scan_topk.cpp     one-sweep-per-query pipeline, certification, escalation, exact replay
//   scan_mq.cpp       shared (multi-query) sweeps on the matrix cores
//   scan_sketch.cpp   8-bit sketch pre-pass
//   scan_radius.cpp   radius search (single, batch, coalesced)
//   scan_comm.cpp     one-process-per-GPU exchange (RCCL all-gather + merge)
//   scan_api.cpp      remaining C entry points (top-k with caller coalescing, distances)
#pragma once
#include "../../include/syzgy_scan.h"
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace szgi {

constexpr int kMaxBatch = 96;  // queries one batch may stage (szg::kMqMaxQueries: a bfloat16 shared sweep)

extern thread_local std::string g_last_error;
int fail(int code, const char *what, hipError_t e = hipSuccess);

#define HIPCHK(expr)                                                    \
    do {                                                                \
        hipError_t e__ = (expr);                                        \
        if (e__ != hipSuccess) return fail(SZG_E_DEVICE, #expr, e__);   \
    } while (0)

// No exception crosses the C boundary: std::vector / std::string growth inside an entry point
// becomes SZG_E_NOMEM.
#define SZG_TRY try {
#define SZG_CATCH                                                          \
    }                                                                      \
    catch (const std::bad_alloc &) { return fail(SZG_E_NOMEM, "out of memory (host)"); } \
    catch (...) { return fail(SZG_E_DEVICE, "unexpected exception"); }

// SZG_DEBUG_TIMERS=1: host time per call site of the enqueue path, printed when a handle is
// destroyed (development aid: which HIP call blocks)
struct SiteTimers {
    static constexpr int N = 12;
    double us[N] = {0};
    uint64_t n[N] = {0};
    const char *name[N] = {"h2d queries", "ev_up+wait", "ev_scan0", "scan launches", "ev_scan1", "ev_done+wait",
                           "merges", "rerank", "d2h", "sentinels", "ev_all", "other"};
    bool on = getenv("SZG_DEBUG_TIMERS") != nullptr;
};
extern SiteTimers g_sites;
struct SiteScope {
    int i;
    std::chrono::steady_clock::time_point t0;
    explicit SiteScope(int i_) : i(i_) { if (g_sites.on) t0 = std::chrono::steady_clock::now(); }
    ~SiteScope()
    {
        if (!g_sites.on) return;
        g_sites.us[i] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        g_sites.n[i]++;
    }
};

int64_t row_bytes_of(int bits, int dim);  // getVectorSize, collection.go:796-811

// ---- container/heap replay (Go stdlib heap.Push / heap.Pop over the
// resultPriorityQueue of collection.go:536-564: max-heap on distance) ---------
struct HeapItem {
    uint64_t row;
    double priority;
};
struct GoHeap {
    std::vector<HeapItem> a;
    bool less(size_t i, size_t j) const { return a[i].priority > a[j].priority; }
    void up(size_t j)
    {
        for (;;) {
            const size_t i = j == 0 ? 0 : (j - 1) / 2;
            if (i == j || !less(j, i)) break;
            std::swap(a[i], a[j]);
            j = i;
        }
    }
    void down(size_t i0, size_t n)
    {
        size_t i = i0;
        for (;;) {
            const size_t j1 = 2 * i + 1;
            if (j1 >= n) break;
            size_t j = j1;
            const size_t j2 = j1 + 1;
            if (j2 < n && less(j2, j1)) j = j2;
            if (!less(j, i)) break;
            std::swap(a[i], a[j]);
            i = j;
        }
    }
    void push(const HeapItem &it)
    {
        a.push_back(it);
        up(a.size() - 1);
    }
    HeapItem pop()
    {
        const size_t n = a.size() - 1;
        std::swap(a[0], a[n]);
        down(0, n);
        HeapItem it = a[n];
        a.pop_back();
        return it;
    }
    // consider()'s top-k branch for one visited record (collection.go:606-619)
    void consider_topk(uint64_t row, double dist, int k)
    {
        if ((int)a.size() <= k) {
            if ((int)a.size() < k || a[0].priority > dist) {
                push(HeapItem{row, dist});
                if ((int)a.size() > k) pop();
            }
        }
    }
    // the pop loop of collection.go:694-697: results in ascending order
    void drain(std::vector<HeapItem> *out)
    {
        out->assign(a.size(), HeapItem{});
        for (size_t i = out->size(); i-- > 0;) (*out)[i] = pop();
    }
};

// per-query constants of the prepared query
struct QMeta {
    double qnorm = 0;   // norm of the prepared (normalised / scaled) query, float paths' error bound
    double m1 = 0;      // sum q_i^2 of the caller's query (zero-query detection)
    double qscale = 0;  // integer paths: prepared query ~ qscale * Q
    double qconst = 0;  // integer paths: sum Q_i
    double qnorm2 = 0;  // euclid: sum g_i^2 of the prepared query g
    bool mq = false;    // answered by the shared float32 MFMA sweep (its own error bound)
    // the int8 shared sweep (8- and 4-bit rows): the query as kMqPlanes int8 digit planes of
    // Q_i = round(v_i / mq_qscale), |Q| <= kMqQmax (the single-query path's own planes stay
    // in qscale / qconst for the escalation sweep)
    bool mq_int = false;
    double mq_qscale = 0, mq_qconst = 0;
    bool mq_bf16 = false;  // (with mq) the shared sweep multiplied bfloat16 roundings of rows and query
};

struct Cand {
    uint64_t row;  // index-level row
    double dist;   // reference float64 distance
    float key;     // the scan's ranking key for this row
    double ub;     // key + the error bound of the arithmetic that produced it: the real-number key is <= ub
};

// ---- one in-flight batch of queries on one shard --------------------------------
struct Ctx {
    hipStream_t stream = nullptr;
    hipStream_t work = nullptr;    // where this batch's uploads and post-processing go: `stream`, or the shard's scan
                                   // stream for a call that is a single batch (set at acquire time)
    int early_n = 0;               // a short call (one batch on the scan stream): the merges, re-rank and copy-back of its
                                   // first early_n queries go onto `stream` behind one event and run -- like the host's
                                   // assembly of those queries -- beside the call's last sweeps (0 = one tail for all)
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;  // HIP-event timing of the early part's sweeps
    bool timed_part = false;
    int timed_part_n = 0;
    hipEvent_t ev_scan0 = nullptr, ev_scan1 = nullptr, ev_all0 = nullptr, ev_all1 = nullptr;
    float mq_qsum[128] = {};             // bfloat16 sweep of 8-bit rows: the sum of each staged query's rounded image values
    hipEvent_t ev_scan_done = nullptr;   // this batch's scans have finished (scan stream)
    hipEvent_t ev_up = nullptr;          // this batch's uploads have finished (ctx stream)
    // pinned host staging, kMaxBatch queries
    uint8_t *h_qsw = nullptr;      // swizzled queries for the scan
    double *h_q64 = nullptr;       // float64 queries for the rerank
    szg::RerankOut *h_out = nullptr;
    size_t h_out_cap = 0;
    uint64_t *h_allow = nullptr;
    size_t h_allow_cap = 0;        // words
    uint32_t *h_count = nullptr;
    // device scratch
    uint8_t *d_qsw = nullptr;
    double *d_q64 = nullptr;
    uint64_t *d_lists_a = nullptr, *d_lists_b = nullptr;
    size_t lists_cap = 0;          // entries per buffer
    szg::RerankOut *d_out = nullptr;
    size_t d_out_cap = 0;
    uint64_t *d_allow = nullptr;
    size_t allow_cap = 0;          // words
    uint64_t *d_collect = nullptr;
    size_t collect_cap = 0;        // entries
    uint32_t *d_count = nullptr;   // hit counters of collect sweeps, one 128-byte line per sweep of a launch
    size_t radius_cap = 0;         // radius batches: entries per sweep the next batch's buffers get (follows the hit counts seen)
    // multi-query sweep: LDS image of the batch, score matrix
    uint8_t *h_mq = nullptr, *d_mq = nullptr;
    int32_t *h_mqQ = nullptr;      // 4-bit int8 sweep: the queries as integers (kMaxBatch x dim)
    size_t h_mq_cap = 0, d_mq_cap = 0;
    float *d_keys = nullptr;
    size_t keys_cap = 0;           // floats
    // fused selection of the shared sweep: thresholds, candidate buffers, hit counts
    float *d_thr = nullptr;
    float *h_thr = nullptr;        // (pinned) the prefix thresholds of a two-stage batch, for certification
    double *h_qscale = nullptr, *d_qscale = nullptr;  // [128] float32-query scale per staged query (re-score)
    int kp_used = 0;               // candidates per query in h_out for the batch in flight
    int out_stride = 0;            // entries per query in h_out: kp_used, + sent_n when the sentinels ride along
    bool sent_deferred = false;    // the sentinel rows are staged (d_sent) but their distances not yet enqueued
    bool sent_in_out = false;      // their distances are entries [kp_used, out_stride) of each query in h_out
    bool sent_own_stream = false;  // their re-rank runs on `stream` while the batch runs on the scan stream (`work`)
    bool mq_band_used = false;     // bfloat16 sweep refined by the band form: h_thr[128 + q] = the band's edge
    bool mq_stage2 = false;        // bfloat16 sweep -> float32 re-score of its candidates -> selection
    bool mq_bf16_used = false;     // the list keys of this batch are bfloat16-sweep keys (matrix form)
    uint64_t *d_cand = nullptr;
    size_t cand_cap_total = 0;     // entries
    uint32_t *d_cand_count = nullptr, *h_cand_count = nullptr;
    bool mq_fused_used = false;
    uint32_t mq_cand_cap = 0;
    int mq_nb = 0;
    bool mq_has_allow = false;
    bool timed_scan = false;
    int timed_n = 0;               // scan launches between ev_scan0 and ev_scan1
    // the first k eligible rows of each staged query in visit order (those consider() pushes
    // whatever their distance, collection.go:608) and their float64 distances: a NaN there
    // poisons the reference's heap, so the query takes the exact replay
    uint64_t *h_sent = nullptr, *d_sent = nullptr;
    size_t h_sent_cap = 0, d_sent_cap = 0;
    szg::RerankOut *h_sent_out = nullptr, *d_sent_out = nullptr;
    size_t h_sent_out_cap = 0, d_sent_out_cap = 0;
    int sent_n = 0;                // entries per query (0 = none staged)
    QMeta meta[kMaxBatch];         // constants of the staged queries
};

struct Shard {
    int device = 0;
    uint64_t first = 0;        // index-level row of this shard's row 0
    uint64_t n_rows = 0;
    uint64_t cap_rows = 0;
    uint64_t n_live = 0;
    uint8_t *rows = nullptr;
    uint64_t *live_bits = nullptr;
    uint64_t bits_cap = 0;     // words
    std::vector<uint64_t> live_host;  // host copy of live_bits (tombstone / append bookkeeping, first-k rows)
    bool has_dead = false;
    int cu_count = 256;
    std::vector<Ctx *> free_ctx;
    std::vector<Ctx *> parked_ctx;   // contexts taken out of rotation ("contexts" option)
    std::vector<Ctx *> all_ctx;
    std::mutex mu;
    std::condition_variable cv;
    // All scan launches of a shard go back to back onto ONE stream: each sweep
    // gets the whole HBM bandwidth and the blocks of a launch stay in lockstep
    // (that is what keeps DRAM pages hot); uploads and the small merge/rerank/
    // copy work of other batches overlap them on the contexts' own streams.
    std::mutex chain_mu;
    hipStream_t scan_stream = nullptr;
    uint8_t *zero16 = nullptr;   // 16 zero bytes idle lanes of the multi-query sweep read
    // resident float32 row norms (16-bit rows, shared bfloat16 sweep): rows [0, norm_valid) are up to date; load /
    // synth reset it, appended rows are caught up before the next shared sweep, an overwritten row at once
    float *row_norm = nullptr;
    uint64_t norm_cap = 0, norm_valid = 0;
    std::mutex norm_mu;
    // device staging of the mutation entry points (load / append / overwrite / read-back): kept
    // between calls, so AddDocument in a loop pays no hipMalloc / hipFree per row
    uint8_t *stage = nullptr;
    size_t stage_cap = 0;
    std::mutex stage_mu;         // szg_index_read_rows may run beside other readers (szg_pair_distances)
    // second stage of the sketch pre-pass: queries | candidate lists | distances, kept between calls
    uint8_t *sk_buf = nullptr;
    size_t sk_buf_cap = 0;
    std::mutex sk_buf_mu;
};

}  // namespace szgi

// One caller of szg_search_topk(n_queries == 1) or szg_search_radius waiting to be answered as part of a batch.
struct PendingSearch {
    const double *query;
    const uint64_t *allow;  // the caller's filter mask, or nullptr
    int k;                  // top-k search (radius == 0)
    double radius = 0;      // > 0: radius search (k ignored, collection.go:598-605)
    uint64_t *out_rows;
    double *out_dist;
    int32_t *out_count = nullptr;   // top-k
    uint64_t capacity = 0;          // radius: room in out_rows / out_dist
    uint64_t *out_total = nullptr;  // radius: the full hit count
    int rc = 0;
    bool done = false;
    bool lead = false;  // told to take over as the batch leader
    std::condition_variable cv;
};

struct szg_index {
    int dim = 0, bits = 0, metric = 0;
    uint32_t row_bytes = 0, pitch = 0;
    szg::RowLayout layout{};  // of every shard's mirror (linear, or 16-row x 64-byte-step tiles)
    szg::RowMap map{};
    size_t qsw_bytes = 0;
    double norm_bias = 0;     // integer paths: sum n^2 = 4(SQ+SV) + norm_bias (padding removed)
    uint64_t row_base = 0;
    std::vector<szgi::Shard *> shards;
    // 8-bit sketch pre-pass for float32 cosine collections ("sketch" option, sketch_sync / search_topk_sketch)
    szg_index *sketch = nullptr;         // an internal 8-bit cosine index over the same rows, same shard ranges
    int sketch_on = 0;
    int sketch_extra = 30;               // sketch neighbours asked for beyond k: k = 10 -> 40, which keeps the sketch
                                         // sweep's lists in registers (kp <= 64); the pre-pass serves k <= 34
    int sketch_min_rows = 4096;          // smaller collections are not worth a second index
    std::mutex sk_mu;                    // the sync
    uint64_t gen = 1, sk_gen = 0;        // mutation counter / the value the sketch was synced at
    bool sk_need_full = true;            // load / synth / reset since the last sync
    bool sk_live_dirty = true;           // tombstones since the last sync
    std::vector<uint64_t> sk_dirty_rows; // rows overwritten since the last sync (index-level)
    double sk_max_ang = 0.0;             // max over the rows of d(row, its sketch), the reference's angular distance
    double sk_gscale = 0.0;              // Euclidean collections: the sketch of a row is sk_gscale * n / 255 (0: cosine)
    std::vector<uint64_t> sk_exc;        // rows without a usable sketch (zero rows, non-finite elements): always re-ranked
    bool sk_disabled = false;            // too many such rows
    std::vector<std::pair<std::string, int64_t>> opt_log;  // tunables set so far (replayed on the sketch index)
    // tunables (szg_set_option; include/syzgy_scan.h lists them)
    int slack_min = 16;
    int n_ctx = 4;            // contexts (and streams) per shard (bfloat16 batches of 1M x 768: 3 -> 4 = 155 -> 162 k queries/s; 6: no more)
    int n_ctx_active = 4;
    int query_batch = 16;     // queries per scan launch
    int radius_mq = 1;        // radius batches of 2+ queries share one sweep of the corpus (the shared sweeps' collect form)
    int finish_thread = 1;    // shared-sweep calls of 3+ batches: a second host thread assembles the finished batches
                              // while the caller's prepares and enqueues the next ones (0 = one thread does both)
    int queries_per_launch = 16;  // sweeps one scan launch walks back to back (query-major)
    int tie_mode = 0;         // 0: exact full replay on ties/NaN, 1: keep the fast answer
    int serialize_scans = 1;  // scan launches of a shard never overlap each other
    int multi_query = 1;      // share one sweep between the queries of a batch (MFMA path)
    int mask_dense = 1;       // masked sweeps whose masks pass most rows use the dense phase
    int coalesce = 1;         // concurrent single-query calls share sweeps (see Combiner)
    int mq_min = 2;           // smallest batch worth a shared sweep (measured: 2 queries already break even)
    int mq_hits = 1024;       // fused selection: candidates per query the full sweep is expected to collect
                              // (sets the prefix: n_rows * kp / mq_hits rows)
    // test hooks: paths that occur by themselves only on particular data
    int force_escalate = 0;   // treat every first pass as uncertified
    int force_matrix = 0;     // shared sweeps: the score-matrix form (what an overflowing candidate buffer falls back to)
    int force_no_refine = 0;  // shared sweeps: the batch's tail as separate re-score / select / rerank launches (kp > 256)
    // settled by measurement (rounds 1-3): compile-time facts since round 4, A/B through -D and `make variant`
    static constexpr int blocks_per_cu = 0;     // 0 = waves per CU chosen from the row format (scan_geometry)
    static constexpr int block_threads = 256;
    static constexpr int first_batch = 4;       // queries of a call's first launch: the card starts sweeping sooner
    static constexpr int short_call = 32;       // calls of up to this many one-sweep queries are ONE batch on the scan stream
    static constexpr int radius_sort = 1;       // a radius batch's re-ranked hits are sorted by distance on the device
    static constexpr int shape_kernels = 1;     // row-shape-specialised kernels where they exist
    static constexpr int ring = 0;              // 8 = always the deep piece ring
    static constexpr int mq_i8 = 1;             // 8- / 4-bit rows: exact integer shared sweep (v_mfma_i32_16x16x64_i8)
    static constexpr int mq_i8_groups = 2;      // ... one launch walks the passes of up to two groups of 48 queries
    static constexpr int mq_bf16 = 1;           // 64- / 32- / 16-bit rows: shared sweep on bfloat16 roundings, certified
                                                // against its own bound and re-ranked in float64 like every other path
    static constexpr int mq_overlap = 1;        // a batch's threshold pass and tail on the context's stream beside the
                                                // neighbouring batches' sweeps
    static constexpr int mq_bf16_slack = 246;   // candidates kept beyond k where the lists hold bfloat16 keys themselves
    static constexpr int mq_tail_overlap = 0;
    static constexpr int mq_blocks_max = 6;     // query blocks of 16 per shared sweep (LDS image permitting)
    int timing = 0;           // 0 off, 1 HIP events around the scan launches, 2 + around the whole per-batch pipeline
    std::mutex stats_mu;
    // coalescing of concurrent single-query searches (szg_search_topk, n_queries == 1)
    std::mutex comb_mu;
    std::deque<struct PendingSearch *> comb_waiting;
    bool comb_leader = false;
    szg_stats stats{};
    szg_comm *comm = nullptr;    // one process per GPU: the attached communicator (borrowed; scan_comm.cpp)
};

namespace szgi {

struct LaunchGeom {
    int grid, block;
};

// ---- api_common.cpp
double now_us();
// consider()'s top-k branch replayed over the candidates in visit order, then the pop loop
void replay_topk(std::vector<Cand> &cands, int k, std::vector<HeapItem> *result);
// a NaN distance, or two exactly equal ones among the best k+1: the reference's answer depends on its heap history
bool history_dependent(const double *dist, size_t n, int k);

// ---- scan_query.cpp
szg::RowMap choose_map(int r16, bool tiled = false);
void prep_query(const szg_index *ix, const double *q, uint8_t *out_sw, QMeta *meta);
void prep_query_meta(const szg_index *ix, const double *q, QMeta *meta);  // the constants only (shared sweeps)
double key_eps(const szg_index *ix, double key, const QMeta &m);
// (radius: the batch is a radius batch -- tiled 8-bit rows then stay on the exact int8 sweep: the bfloat16 sweep's
// band around every radius would collect several times the hits)
// nq: the queries of the batch in question -- 8-bit rows take the bfloat16 sweep only for MORE than 48 of them (up to 48
// fit one int8 pass, which measures 7-10 % faster than a bfloat16 pass of three query blocks)
bool mq_uses_i8(const szg_index *ix, bool radius = false, int nq = 1 << 30);
bool mq_uses_bf16(const szg_index *ix, bool radius = false, int nq = 1 << 30);
uint16_t bf16_rne(float f);
double mq_int_scale(const szg_index *ix, double m1);
void prep_mq_int(const szg_index *ix, const double *q, QMeta *meta, int32_t *Qout);
int mq_blocks(const szg_index *ix, int nq, bool radius = false);
// key threshold of a radius search: surely contains every row with distance <= radius
float radius_key_threshold(const szg_index *ix, double radius, const QMeta &meta);

// ---- scan_handle.cpp
int ctx_alloc(szg_index *ix, Shard *sh, Ctx **out);
void ctx_free(Ctx *c);
Ctx *ctx_acquire(Shard *sh);
Ctx *ctx_try_acquire(Shard *sh);
void ctx_release(Shard *sh, Ctx *c);
int shard_stage(Shard *sh, size_t bytes, uint8_t **out);
int upload_rows(szg_index *ix, Shard *sh, uint64_t dst_row, const uint8_t *rows, uint64_t n);
int shard_reserve(szg_index *ix, Shard *sh, uint64_t rows_needed);
int shard_set_live(Shard *sh, uint64_t lo, uint64_t hi);
void split_rows(const szg_index *ix, uint64_t n_rows, std::vector<uint64_t> *counts);
Shard *shard_of(szg_index *ix, uint64_t row, uint64_t *local);
int reset_shards(szg_index *ix, const std::vector<uint64_t> &counts);
Shard *append_target(szg_index *ix);
void note_overwritten(szg_index *ix, uint64_t row);

// ---- scan_topk.cpp
LaunchGeom scan_geometry(const szg_index *ix, const Shard *sh, int kp, bool plain_topk = false);
size_t shard_words(const Shard *sh);
int enqueue_queries(szg_index *ix, Shard *sh, Ctx *c, const double *q, int nq, const uint64_t *const *masks,
                    bool with_single_form = true);
void fill_scan_args(const szg_index *ix, const Shard *sh, const Ctx *c, bool has_allow, int slot, int nq,
                    szg::ScanArgs *a);
// (after: the stream that goes on once the sweeps are done -- default c->work; part 1: the early part of a short call,
// timed with its own event pair)
int launch_scans_chained(szg_index *ix, Shard *sh, Ctx *c, const std::vector<szg::ScanArgs> &a, const LaunchGeom &g,
                         hipStream_t after = nullptr, int part = 0);
int enqueue_topk(szg_index *ix, Shard *sh, Ctx *c, int kp, int nq, bool has_allow);
// float64 distances of the staged sentinel rows (d_sent) on `stream`, results to h_sent_out
int launch_sentinel_rerank(szg_index *ix, Shard *sh, Ctx *c, int nq, hipStream_t stream);
int finish_timing(szg_index *ix, Ctx *c);
int run_collect(szg_index *ix, Shard *sh, Ctx *c, int slot, float thr_key, bool has_allow, std::vector<Cand> *cands);
// fraction of the shard's rows that staged query `slot` may visit (tombstones, and a sample of its filter mask's words)
double mask_pass_rate(const Shard *sh, const Ctx *c, bool has_allow, int slot);
void first_eligible_rows(const szg_index *ix, const uint64_t *allow, int k, std::vector<uint64_t> *rows_out);
int search_topk_impl(szg_index *ix, const double *queries, int n_queries, int k, const uint64_t *allow_bits,
                     uint64_t *out_rows, double *out_dist, int32_t *out_count,
                     const uint64_t *const *allow_ptrs = nullptr);
// consider()'s top-k branch over every row of the handle in visit order, continuing the heap *h (rows + row_base)
int replay_rows_into_heap(szg_index *ix, const double *query, const uint64_t *allow, int k, GoHeap *h);

// ---- scan_radius.cpp
// radius searches for a batch of queries (own radius and filter mask each): results[i] = the hits of query i, ascending
int search_radius_impl(szg_index *ix, const double *queries, int n_queries, const double *radii,
                       const uint64_t *const *masks, std::vector<std::vector<HeapItem>> *results);

#ifndef SZG_BF16_8BIT_DEFAULT
#define SZG_BF16_8BIT_DEFAULT 1  // tiled 8-bit rows take the bfloat16 sweep for top-k batches (SZG_BF16_8BIT=0: the int8 sweep)
#endif
// ---- scan_mq.cpp
// the shard's resident row norms are complete (16-bit rows; no-op otherwise): called before a shared sweep is enqueued
int ensure_row_norms(szg_index *ix, Shard *sh);
int enqueue_topk_mq(szg_index *ix, Shard *sh, Ctx *c, int kp, int kp_wide, int nq, int nb, bool has_allow,
                    bool force_matrix = false);

// the batch's tail will compute the sentinel rows' distances itself (stage them, do not launch their own rerank)
bool mq_tail_takes_sentinels(const szg_index *ix, const Shard *sh, int kp, int kp_wide, int nq, int nb);
// radius batches: ONE shared sweep collects every (query, row) pair at or below the query's key threshold
// (thr[q], host) into c->d_collect (cap entries per query, counts in c->d_count)
int enqueue_collect_mq(szg_index *ix, Shard *sh, Ctx *c, int nq, int nb, bool has_allow, const float *thr, size_t cap);

// ---- scan_sketch.cpp
int search_topk_any(szg_index *ix, const double *queries, int n_queries, int k, const uint64_t *allow_bits,
                    uint64_t *out_rows, double *out_dist, int32_t *out_count,
                    const uint64_t *const *allow_ptrs = nullptr);

struct CtxGuard {  // returns a borrowed context on every exit path
    Shard *sh;
    Ctx *c;
    ~CtxGuard() { ctx_release(sh, c); }
};

template <typename T>
int ensure_dev(T **p, size_t *cap, size_t need)
{
    if (*cap >= need) return SZG_OK;
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t n = std::max(need, (size_t)64);
    HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
    *cap = n;
    return SZG_OK;
}
template <typename T>
int ensure_host(T **p, size_t *cap, size_t need)
{
    if (*cap >= need) return SZG_OK;
    if (*p) HIPCHK(hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t n = std::max(need, (size_t)64);
    HIPCHK(hipHostMalloc((void **)p, n * sizeof(T), hipHostMallocDefault));
    *cap = n;
    return SZG_OK;
}


struct Ticket {
    int first = 0, nq = 0;       // queries [first, first+nq) of the call
    std::vector<Ctx *> ctx;      // one per shard
    std::vector<QMeta> meta;
    int kp = 0, kp_wide = 0;
    bool failed = false;         // enqueueing failed part-way: drain and release only
    bool any_mask = false;       // some query of the batch carries a filter mask
    bool lazy_single = false;    // shared sweep: the queries' single-query form (h_qsw / d_qsw) has not been built
    szg_index *owner = nullptr;
    Ticket() = default;
    Ticket(Ticket &&) = default;
    Ticket(const Ticket &) = delete;
    Ticket &operator=(const Ticket &) = delete;
    // a ticket dropped with contexts still attached (an exception unwinding the call) drains
    // and returns them, so later calls do not wait for contexts that never come back
    ~Ticket()
    {
        if (!owner) return;
        for (size_t s = 0; s < ctx.size(); s++) {
            if (!ctx[s]) continue;
            (void)hipSetDevice(owner->shards[s]->device);
            (void)hipStreamSynchronize(ctx[s]->work);
            (void)hipStreamSynchronize(ctx[s]->stream);
            ctx[s]->mq_fused_used = false;
            ctx_release(owner->shards[s], ctx[s]);
        }
    }
};

}  // namespace szgi
