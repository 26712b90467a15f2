// scan_api.cpp -- implementation of include/syzgy_scan.h (the C ABI).
//
// Host side of the drop-in: owns the HBM mirror of a Collection's packed
// vectors and runs, per batch of queries, the pipeline
//     H2D queries -> fused scan (query-major, one launch per batch)
//                 -> list merges -> float64 rerank -> D2H
// on pooled HIP streams, then does the reference's result assembly on the few
// survivors: certification of the candidate set, the container/heap replay of
// consider() (collection.go:598-619) and the ascending pop loop (:694-697).
//
// There is no CPU scan in here: without a usable gfx950 device every entry
// point that computes fails with SZG_E_NODEVICE.
#include "../../include/syzgy_scan.h"
#include "kernels.h"

#include <algorithm>
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
#include <vector>

namespace {

constexpr int kMaxBatch = 96;  // queries one batch may stage (szg::kMqMaxQueries: a bfloat16 shared sweep)

thread_local std::string g_last_error;

int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof(buf), "%s", what);
    g_last_error = buf;
    return code;
}

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
SiteTimers g_sites;
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

int64_t row_bytes_of(int bits, int dim)
{   // getVectorSize, collection.go:796-811
    switch (bits) {
    case 4: return ((int64_t)dim + 1) / 2;
    case 8: return dim;
    case 16: return (int64_t)dim * 2;
    case 32: return (int64_t)dim * 4;
    case 64: return (int64_t)dim * 8;
    default: return -1;
    }
}

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
    hipEvent_t ev_scan0 = nullptr, ev_scan1 = nullptr, ev_all0 = nullptr, ev_all1 = nullptr;
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
    uint32_t *d_count = nullptr;
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

}  // namespace

// One caller of szg_search_topk(n_queries == 1) waiting to be answered as part of a batch.
struct PendingSearch {
    const double *query;
    const uint64_t *allow;  // the caller's filter mask, or nullptr
    int k;
    uint64_t *out_rows;
    double *out_dist;
    int32_t *out_count;
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
    std::vector<Shard *> shards;
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
    // tunables
    int slack_min = 16;
    int n_ctx = 3;            // contexts (and streams) per shard
    int n_ctx_active = 3;
    int blocks_per_cu = 0;    // 0 = choose from the row format (scan_geometry)
    int block_threads = 256;
    int query_batch = 16;     // queries per scan launch
    int shape_kernels = 1;    // use the row-shape-specialised scan kernels where they exist
    int ring = 0;             // tuning hook: 8 = always the deep piece ring
    int queries_per_launch = 16;  // sweeps one scan launch walks back to back (query-major)
    int force_escalate = 0;   // test hook: treat every first pass as uncertified
    int tie_mode = 0;         // 0: exact full replay on ties/NaN, 1: keep the fast answer
    int serialize_scans = 1;  // scan launches of a shard never overlap each other
    int multi_query = 1;      // share one sweep between the queries of a batch (MFMA path)
    int mask_dense = 1;       // masked sweeps whose masks pass most rows use the dense phase
    int coalesce = 1;         // concurrent single-query calls share sweeps (see Combiner)
    int mq_fused = 1;         // shared sweep: threshold-collect selection instead of a score matrix
    int mq_i8 = 1;            // 8-bit rows: exact integer shared sweep (v_mfma_i32_16x16x64_i8)
    int mq_i8_groups = 2;     // int8 sweeps: query groups of 48 one launch walks (1 or 2)
    int mq_bf16 = 1;          // 32-bit rows: shared sweep on bfloat16 roundings (v_mfma_f32_16x16x32_bf16), certified
                              // against its own bound and re-ranked in float64 like every other path
    int mq_overlap = 1;       // bfloat16 sweeps: a batch's threshold pass and its post-processing run on the context's
                              // stream beside the neighbouring batches' sweeps (the sweep is a bare stream of the rows)
    int mq_bf16_slack = 118;  // candidates kept beyond k by a bfloat16 sweep (its band holds more rows)
    int mq_tail_overlap = 0;  // shared sweep: post-processing of a batch beside the next batch's sweep
    int mq_min = 2;           // smallest batch worth a shared sweep (measured: 2 queries already break even)
    int mq_blocks_max = 6;    // query blocks of 16 per shared sweep (LDS image permitting; 3 at most for the
                              // float32 and int8 sweeps, 6 for the bfloat16 sweep)
    int mq_hits = 1024;       // fused selection: candidates per query the full sweep is expected to collect
                              // (sets the prefix: n_rows * kp / mq_hits rows)
    int timing = 0;           // 0 off, 1 HIP events around the scan launches, 2 + around the whole per-batch pipeline
    std::mutex stats_mu;
    // coalescing of concurrent single-query searches (szg_search_topk, n_queries == 1)
    std::mutex comb_mu;
    std::deque<struct PendingSearch *> comb_waiting;
    bool comb_leader = false;
    szg_stats stats{};
};

namespace {

szg::RowMap choose_map(int r16, bool tiled = false)
{
    if (tiled) return szg::RowMap{r16, 4, r16 / 4, 16, 1, 1};  // one 64-byte step of 16 rows per load   // Groups of L lanes per row, P pieces per lane.
    // 1) Exact power-of-two groups (L*P == r16): every lane always holds a piece (the
    //    kernel's dense phase), reductions are DPP.  The SMALLEST such L >= 8 wins: a
    //    group still reads whole 128-byte lines per load, and the fewer lanes share a
    //    row, the more pieces each walks between two row finishes (measured on
    //    1M x 768 f32: L=8 6.80 TB/s, L=16 6.73, L=32 6.47, L=64 6.44; 8-bit: L=8 5.9 vs
    //    L=16 5.75 vs L=32 3.8).  L=4 (64-byte segments) only when nothing wider is exact.
    for (int L : {8, 16, 32, 64, 4})
        if (r16 % L == 0) return szg::RowMap{r16, L, r16 / L, 64 / L, 1, 1};
    // 2) Otherwise maximise lane utilisation, with a bonus for power-of-two groups.
    szg::RowMap best{r16, 64, (r16 + 63) / 64, 1, 1, 0};
    double best_score = -1;
    const int pmax = std::max(1, (r16 + 63) / 64 + 8);
    for (int P = 1; P <= pmax; P++) {
        const int need = (r16 + P - 1) / P;  // lanes a row needs at P pieces per lane
        if (need > 64) continue;
        int cand[2] = {need, 1};
        while (cand[1] < need) cand[1] <<= 1;  // next power of two
        for (int L : cand) {
            if (L > 64) continue;
            const int gpw = 64 / L;
            const double util = (double)gpw * r16 / (64.0 * P);
            const bool pow2 = (L & (L - 1)) == 0;
            const double score = util * (pow2 ? 1.3 : 1.0);
            if (score > best_score + 1e-9) {
                best_score = score;
                best = szg::RowMap{r16, L, P, gpw, pow2 ? 1 : 0, 0};
            }
        }
    }
    best.dense = (best.L * best.P == r16 && best.gpw * best.L == 64) ? 1 : 0;
    return best;
}

// round to nearest (ties away from zero) without a libm call; NaN -> 0, clamped to +-lim.
// Any rounding rule serves: Q only has to be within 1/2 of v/qscale (key_eps).
inline long long round_clamp(double t, double lim)
{
    if (!(t == t)) return 0;
    if (t > lim) t = lim;
    if (t < -lim) t = -lim;
    return (long long)(t + (t >= 0 ? 0.5 : -0.5));
}

// Query as the scan wants it (see RowAcc in kernels_scan.hip):
//  * 16/32/64-bit rows: float (double for 64-bit), pre-normalised for cosine,
//    pre-scaled by maxInt for 16-bit euclid, laid out [chunk][piece][4];
//  * 8/4-bit rows: the prepared real query v (q/|q| for cosine, maxInt*q for
//    euclid) quantized to integers Q_i = round(v_i / qscale) and split into
//    balanced digit planes (3 x int8 radix 128, or 5 x int4 radix 16), one
//    16-byte plane word per 16-byte piece of the row.
void prep_query(const szg_index *ix, const double *q, uint8_t *out_sw, QMeta *meta)
{
    const int dim = ix->dim, bits = ix->bits;
    const int E = 128 / bits;
    const int r16 = ix->map.r16;
    memset(out_sw, 0, ix->qsw_bytes);
    *meta = QMeta{};
    double m1 = 0.0;
    for (int i = 0; i < dim; i++) m1 += q[i] * q[i];
    meta->m1 = m1;
    double scale = 1.0;
    if (ix->metric == SZG_COSINE) {
        scale = m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0;
    } else if (bits <= 16) {
        scale = (double)((1u << bits) - 1u);
    }
    double nrm = 0.0, vmax = 0.0;
    for (int e = 0; e < dim; e++) {
        const double v = q[e] * scale;
        nrm += v * v;
        vmax = std::max(vmax, std::fabs(v));
    }
    meta->qnorm = std::sqrt(nrm);
    meta->qnorm2 = nrm;
    if (bits == 8 || bits == 4) {
        const double Qmax = bits == 8 ? 1000000.0 : szg::kQmax4;
        const double qs = (vmax > 0 && std::isfinite(vmax)) ? vmax / Qmax : 1.0;
        meta->qscale = qs;
        double sumQ = 0.0;
        uint32_t *planes = reinterpret_cast<uint32_t *>(out_sw);
        for (int e = 0; e < dim; e++) {
            long long Q = round_clamp(q[e] * scale / qs, Qmax);
            sumQ += (double)Q;
            const int j = e / E, i = e % E;
            if (bits == 8) {
                const int d = i / 4, kb = i % 4;
                for (int x = 2; x >= 0; x--) {  // planes: 0 = h (x16384), 1 = m (x128), 2 = l
                    long long dig;
                    if (x > 0) {
                        dig = ((Q + 64) & 127) - 64;
                        Q = (Q - dig) >> 7;
                    } else {
                        dig = Q;
                    }
                    planes[((size_t)x * r16 + j) * 4 + d] |= (uint32_t)((uint8_t)(int8_t)dig) << (8 * kb);
                }
            } else {
                // byte b of the piece holds element 2b in its high nibble, 2b+1 in the low one
                const int bb = i / 2, d = bb / 4, kb = bb % 4;
                const int t4 = 2 * kb + ((i % 2 == 0) ? 1 : 0);
                for (int x = 0; x < szg::kPlanes4; x++) {  // plane x carries the digit of weight 16^x
                    long long dig;
                    if (x < szg::kPlanes4 - 1) {
                        dig = ((Q + 8) & 15) - 8;
                        Q = (Q - dig) >> 4;
                    } else {
                        dig = Q;
                    }
                    planes[((size_t)x * r16 + j) * 4 + d] |= (uint32_t)(dig & 0xF) << (4 * t4);
                }
            }
        }
        meta->qconst = sumQ;
        return;
    }
    for (int e = 0; e < dim; e++) {
        const double v = q[e] * scale;
        const int j = e / E, i = e % E;
        if (bits == 64) {
            reinterpret_cast<double *>(out_sw)[(size_t)j * 2 + i] = v;
        } else {
            const int c = i / 4, m = i % 4;
            reinterpret_cast<float *>(out_sw)[((size_t)c * r16 + j) * 4 + m] = (float)v;
        }
    }
}

// Bound on |scan key - real-number key| (see DESIGN.md "certification").
double key_eps(const szg_index *ix, double key, const QMeta &m)
{
    const double k = std::fabs(key);
    if (m.mq) {
        // shared sweep: float32 everywhere (quantized rows decode to exact integers first).
        // cosine: dot and norm each carry <= (dim+16) u relative error.  euclid: the key is
        // |x|^2 - 2 x.g + |g|^2, three float32 sums whose magnitudes are bounded by
        // (|x| + |g|)^2 <= (2|g| + sqrt(key))^2 -- an absolute bound, far looser than the
        // difference form's when rows sit far from the origin; certification then simply
        // escalates more often.
        const double u = 0x1p-24, n = (double)ix->dim + 16.0;
        if (m.mq_bf16) {
            // bfloat16 sweep: each operand is rounded to 8 significant bits (relative error <= 2^-9,
            // the query once more from float32), the products are exact in float32 and summed by the
            // matrix core in float32.  |sum bf(x_i) bf(g_i) - sum x_i g_i| <= c |x| |g| (Cauchy-Schwarz)
            // with c = (1 + 2^-9)^2 (1 + 2^-24) - 1 < 1.01 * 2^-8; the float32 part of the bound is
            // doubled (the accumulation order and rounding of the matrix core are its own).
            const double c = 1.01 * 0x1p-8;
            if (ix->metric == SZG_COSINE) return c + 4.0 * n * u + 1e-6;
            // euclid: the key moves by 2 c |x| |g|, and |x| <= |g| + d with d^2 <= key + 2 c |x| |g|
            // gives |x| <= 1.1 |g| + sqrt(key) for this c; the last term keeps key - eps(key) monotone
            const double s = 2.0 * m.qnorm + std::sqrt(k);
            return 2.0 * c * m.qnorm * (1.1 * m.qnorm + std::sqrt(k)) + c * c * m.qnorm2 + 3.0 * n * u * s * s + 1e-30;
        }
        if (ix->metric == SZG_COSINE) return 2.0 * n * u;
        const double s = 2.0 * m.qnorm + std::sqrt(k);
        return 1.5 * n * u * s * s + 1e-30;
    }
    if (m.mq_int) {
        // as the integer branch below with the sweep's own quantization step; the row operand is
        // v' = v - 128 (8-bit rows) or the nibble x in 0..15 (4-bit rows)
        const double M = (double)((1u << ix->bits) - 1u);
        const double V = ix->bits == 8 ? 128.0 : 16.0, Qmax = szg::kMqQmax;
        const double fl = 16.0 * 0x1p-24 * m.mq_qscale * Qmax * V * (double)ix->dim;
        if (ix->metric == SZG_COSINE)
            return 0.5 * m.mq_qscale * std::sqrt((double)ix->dim) + fl / std::sqrt((double)ix->dim) + 0x1p-21;
        return m.mq_qscale * M * (double)ix->dim + 2.0 * fl +
               0x1p-21 * (k + m.qnorm2 + M * M * (double)ix->dim) + 1e-30;
    }
    if (ix->bits == 8 || ix->bits == 4) {
        // integer paths: the per-lane sums are exact.  What is left is (a) the query's
        // quantization, |v_i - qscale*Q_i| <= qscale/2, and (b) the float32 roundings of
        // the row finish: the plane combination and the reduction over the lanes act on
        // terms bounded by sum |Q_i||v'_i| <= Qmax*V*dim (V = 128 resp. 8), i.e. an
        // absolute error <= 16*2^-24 * qscale*Qmax*V*dim in units of sum v n.
        const double M = (double)((1u << ix->bits) - 1u);
        const double V = ix->bits == 8 ? 128.0 : 8.0;
        const double Qmax = ix->bits == 8 ? 1000000.0 : szg::kQmax4;
        const double fl = 16.0 * 0x1p-24 * m.qscale * Qmax * V * (double)ix->dim;
        if (ix->metric == SZG_COSINE)  // divided by |n| >= sqrt(dim) (every n is odd)
            return 0.5 * m.qscale * std::sqrt((double)ix->dim) + fl / std::sqrt((double)ix->dim) + 0x1p-21;
        return m.qscale * M * (double)ix->dim + 2.0 * fl +
               0x1p-21 * (k + m.qnorm2 + M * M * (double)ix->dim) + 1e-30;
    }
    const double u = ix->bits == 64 ? 0x1p-53 : 0x1p-24;
    const double n = (double)ix->dim + 16.0;
    if (ix->metric == SZG_COSINE) {
        return 2.0 * n * u + (ix->bits == 64 ? 0x1p-22 : 0.0);
    }
    return 2.0 * n * u * k + 8.0 * u * m.qnorm * std::sqrt(k) + (ix->bits == 64 ? 0x1p-22 * k : 0.0) +
           1e-37;
}

int ctx_alloc(szg_index *ix, Shard *sh, Ctx **out)
{
    Ctx *c = new Ctx();
    *out = c;
    HIPCHK(hipSetDevice(sh->device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&c->ev_scan0));
    HIPCHK(hipEventCreate(&c->ev_scan1));
    HIPCHK(hipEventCreate(&c->ev_all0));
    HIPCHK(hipEventCreate(&c->ev_all1));
    HIPCHK(hipEventCreateWithFlags(&c->ev_scan_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    const size_t B = kMaxBatch;
    HIPCHK(hipHostMalloc((void **)&c->h_qsw, B * ix->qsw_bytes, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_q64, B * sizeof(double) * ix->dim, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_count, sizeof(uint32_t) * 4, hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&c->d_qsw, B * ix->qsw_bytes));
    HIPCHK(hipMalloc((void **)&c->d_q64, B * sizeof(double) * ix->dim));
    HIPCHK(hipMalloc((void **)&c->d_count, sizeof(uint32_t) * 4));
    return SZG_OK;
}

void ctx_free(Ctx *c)
{
    if (!c) return;
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (hipEvent_t e : {c->ev_scan0, c->ev_scan1, c->ev_all0, c->ev_all1, c->ev_scan_done, c->ev_up})
        if (e) (void)hipEventDestroy(e);
    (void)hipHostFree(c->h_qsw);
    (void)hipHostFree(c->h_q64);
    (void)hipHostFree(c->h_out);
    (void)hipHostFree(c->h_allow);
    (void)hipHostFree(c->h_count);
    (void)hipFree(c->d_qsw);
    (void)hipFree(c->d_q64);
    (void)hipFree(c->d_lists_a);
    (void)hipFree(c->d_lists_b);
    (void)hipFree(c->d_out);
    (void)hipFree(c->d_allow);
    (void)hipFree(c->d_collect);
    (void)hipFree(c->d_count);
    (void)hipHostFree(c->h_mq);
    free(c->h_mqQ);
    (void)hipFree(c->d_mq);
    (void)hipFree(c->d_thr);
    (void)hipHostFree(c->h_thr);
    (void)hipHostFree(c->h_qscale);
    (void)hipFree(c->d_qscale);
    (void)hipFree(c->d_cand);
    (void)hipFree(c->d_cand_count);
    (void)hipHostFree(c->h_cand_count);
    (void)hipFree(c->d_keys);
    (void)hipHostFree(c->h_sent);
    (void)hipFree(c->d_sent);
    (void)hipHostFree(c->h_sent_out);
    (void)hipFree(c->d_sent_out);
    delete c;
}

Ctx *ctx_acquire(Shard *sh)
{
    std::unique_lock<std::mutex> lk(sh->mu);
    sh->cv.wait(lk, [&] { return !sh->free_ctx.empty(); });
    Ctx *c = sh->free_ctx.back();
    sh->free_ctx.pop_back();
    return c;
}
Ctx *ctx_try_acquire(Shard *sh)
{
    std::lock_guard<std::mutex> lk(sh->mu);
    if (sh->free_ctx.empty()) return nullptr;
    Ctx *c = sh->free_ctx.back();
    sh->free_ctx.pop_back();
    return c;
}
void ctx_release(Shard *sh, Ctx *c)
{
    {
        std::lock_guard<std::mutex> lk(sh->mu);
        sh->free_ctx.push_back(c);
    }
    sh->cv.notify_one();
}

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

struct LaunchGeom {
    int grid, block;
};

LaunchGeom scan_geometry(const szg_index *ix, const Shard *sh, int kp, bool plain_topk = false)
{
    int block = ix->block_threads;
    // keep query + per-wave lists within 64 KiB of LDS
    while (block > 64 && szg::scan_lds_bytes(ix->bits, ix->map, kp, block) > 64u * 1024u) block >>= 1;
    const int nwaves = block / 64;
    const uint64_t rows_per_block = (uint64_t)nwaves * ix->map.gpw;
    uint64_t need = (sh->n_rows + rows_per_block - 1) / rows_per_block;
    int waves_per_cu = ix->blocks_per_cu * nwaves;
    if (ix->blocks_per_cu <= 0) {
        // Measured on MI355X (scripts/dev_bpc.sh, scripts/readbw): HBM streams fastest with
        // 6-8 MB of reads in flight; more requests only lengthen the DRAM queues.  8 waves
        // per CU for float rows of >= 1 KB and for LDS-resident candidate lists (kp > 64);
        // the integer / 16-bit decodes and short rows need 12 to hide their ALU work.
        // (4 waves per CU is another 0.5 % faster on 3 KB rows at 1M rows but 10 % slower
        // on a 125 K-row shard, where the sweep's ramp-up and tail weigh more.)
        (void)plain_topk;
        // Collect sweeps (kp == 0: radius search, escalation) keep no lists; on short 4-bit rows (cfg5's 192 bytes)
        // they stream best with 8 (same-box A/B, scripts/ab_opts.sh: 6.2-6.7 -> 6.85-6.91 TB/s; top-k on the same rows
        // wants its 12: 6.8-6.9 against 6.5).
        const bool short_collect = kp == 0 && ix->bits == 4 && ix->row_bytes <= 256;
        if (kp > 64 || short_collect || (ix->bits >= 32 && ix->row_bytes >= 1024) || (ix->bits == 8 && ix->layout.tiled))
            waves_per_cu = 8;
        else
            waves_per_cu = 12;
    }
    uint64_t grid = (uint64_t)sh->cu_count * (uint64_t)std::max(1, waves_per_cu / nwaves);
    if (need < grid) grid = need;
    if (grid < 1) grid = 1;
    return LaunchGeom{(int)grid, block};
}

size_t shard_words(const Shard *sh) { return (size_t)((sh->n_rows + 63) / 64); }

// Enqueue H2D of nq prepared queries (+ their masks) on the ctx stream.
// masks: nullptr (no query of the batch is filtered), or nq pointers to index-level masks
// ((total_rows + 63) / 64 words each); a null entry allows every row.
int enqueue_queries(szg_index *ix, Shard *sh, Ctx *c, const double *q, int nq, const uint64_t *const *masks)
{
    HIPCHK(hipSetDevice(sh->device));
    memcpy(c->h_q64, q, sizeof(double) * ix->dim * nq);
    if (ix->timing >= 2) {
        SiteScope t_(10);
        HIPCHK(hipEventRecord(c->ev_all0, c->stream));
    }
    {
        SiteScope t_(0);
        HIPCHK(hipMemcpyAsync(c->d_qsw, c->h_qsw, ix->qsw_bytes * nq, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_q64, c->h_q64, sizeof(double) * ix->dim * nq, hipMemcpyHostToDevice,
                              c->stream));
    }
    if (masks) {
        const size_t words = shard_words(sh);
        int rc = ensure_dev(&c->d_allow, &c->allow_cap, words * nq);
        if (rc) return rc;
        rc = ensure_host(&c->h_allow, &c->h_allow_cap, words * nq);
        if (rc) return rc;
        for (int i = 0; i < nq; i++) {
            if (masks[i])
                memcpy(c->h_allow + (size_t)i * words, masks[i] + sh->first / 64, words * sizeof(uint64_t));
            else
                memset(c->h_allow + (size_t)i * words, 0xFF, words * sizeof(uint64_t));
        }
        HIPCHK(hipMemcpyAsync(c->d_allow, c->h_allow, words * nq * sizeof(uint64_t),
                              hipMemcpyHostToDevice, c->stream));
    }
    // what the sweeps wait for ends here: work enqueued on this stream afterwards (the first-k
    // rows' distances) runs beside the sweeps
    HIPCHK(hipEventRecord(c->ev_up, c->stream));
    return SZG_OK;
}

// scan arguments for queries [slot, slot+nq) of the ctx's staged batch
void fill_scan_args(const szg_index *ix, const Shard *sh, const Ctx *c, bool has_allow, int slot,
                    int nq, szg::ScanArgs *a)
{
    memset(a, 0, sizeof(*a));
    a->rows = sh->rows;
    a->n_rows = (uint32_t)sh->n_rows;
    a->pitch = ix->pitch;
    a->tiled = ix->layout.tiled;
    a->steps = ix->layout.steps;
    a->dim = ix->dim;
    a->map = ix->map;
    a->live_bits = sh->has_dead ? sh->live_bits : nullptr;
    a->allow_stride = (uint32_t)shard_words(sh);
    a->allow_bits = has_allow ? c->d_allow + (size_t)slot * a->allow_stride : nullptr;
    a->query_stride = (uint32_t)ix->qsw_bytes;
    a->query = c->d_qsw + (size_t)slot * ix->qsw_bytes;
    a->n_queries = nq;
    for (int j = 0; j < nq && j < szg::kMaxSweepsPerLaunch; j++) {
        a->qscale[j] = (float)c->meta[slot + j].qscale;
        a->qconst[j] = (float)c->meta[slot + j].qconst;
        a->qnorm2[j] = (float)c->meta[slot + j].qnorm2;
    }
    a->norm_bias = ix->norm_bias;
    a->no_shape_kernels = ix->shape_kernels ? 0 : 1;
    a->ring = ix->ring;
}

// Launch the fused scan for each of the batch's queries (n = a->size()) as the
// next links of the shard's scan chain; the ctx stream resumes after the last.
int launch_scans_chained(szg_index *ix, Shard *sh, Ctx *c, const std::vector<szg::ScanArgs> &a,
                         const LaunchGeom &g)
{
    const int n = (int)a.size();
    {
        std::lock_guard<std::mutex> lk(sh->chain_mu);
        hipStream_t st = ix->serialize_scans ? sh->scan_stream : c->stream;
        if (st != c->stream) {
            SiteScope t_(1);
            HIPCHK(hipStreamWaitEvent(st, c->ev_up, 0));  // recorded by enqueue_queries
        }
        if (ix->timing) {
            SiteScope t_(2);
            HIPCHK(hipEventRecord(c->ev_scan0, st));
        }
        {
            SiteScope t_(3);
            for (int j = 0; j < n; j++)
                HIPCHK(szg::launch_scan(ix->bits, ix->metric, a[j], g.grid, g.block, st));
        }
        if (ix->timing) {
            SiteScope t_(4);
            HIPCHK(hipEventRecord(c->ev_scan1, st));
            c->timed_scan = true;
            c->timed_n = n;
        }
        if (st != c->stream) {
            SiteScope t_(5);
            HIPCHK(hipEventRecord(c->ev_scan_done, st));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_scan_done, 0));
        }
    }
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    uint64_t sweeps = 0;
    for (const szg::ScanArgs &x : a) sweeps += (uint64_t)x.n_queries;
    ix->stats.scan_launches += n;
    ix->stats.scan_bytes += sweeps * sh->n_rows * (uint64_t)ix->row_bytes;
    return SZG_OK;
}

// top-k pass for the nq staged queries of one shard: scan -> merges -> rerank -> D2H (async)
int enqueue_topk(szg_index *ix, Shard *sh, Ctx *c, int kp, int nq, bool has_allow)
{
    c->kp_used = kp;
    c->mq_stage2 = false;
    c->mq_bf16_used = false;
    HIPCHK(hipSetDevice(sh->device));
    const LaunchGeom g = scan_geometry(ix, sh, kp, !has_allow && !sh->has_dead);
    const size_t need = (size_t)nq * g.grid * kp;
    if (c->lists_cap < need) {  // both ping-pong buffers grow together
        if (c->d_lists_a) HIPCHK(hipFree(c->d_lists_a));
        if (c->d_lists_b) HIPCHK(hipFree(c->d_lists_b));
        c->d_lists_a = c->d_lists_b = nullptr;
        c->lists_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_lists_a, need * sizeof(uint64_t)));
        HIPCHK(hipMalloc((void **)&c->d_lists_b, need * sizeof(uint64_t)));
        c->lists_cap = need;
    }
    int rc = ensure_dev(&c->d_out, &c->d_out_cap, (size_t)nq * kp);
    if (rc) return rc;
    rc = ensure_host(&c->h_out, &c->h_out_cap, (size_t)nq * kp);
    if (rc) return rc;

    // Masked sweeps: when most rows pass (a few tombstones, a mild filter) every row is read and
    // the masks decide at the row finish -- the predicate-free dense phase; a selective filter
    // keeps the form that tests a row before issuing its loads.  Pass rates are estimated from
    // a sample of each mask's words.
    auto pass_rate = [&](int j) -> double {
        double live = sh->n_rows ? (double)sh->n_live / (double)sh->n_rows : 1.0;
        if (!has_allow) return live;
        const size_t words = shard_words(sh);
        const uint64_t *m = c->h_allow + (size_t)j * words;
        const size_t step = std::max<size_t>(1, words / 256);
        uint64_t ones = 0, seen = 0;
        for (size_t w = 0; w < words; w += step) {
            ones += (uint64_t)__builtin_popcountll(m[w]);
            seen += 64;
        }
        return live * (seen ? (double)ones / (double)seen : 1.0);
    };
    const bool masked = has_allow || sh->has_dead;
    const int qpl = std::max(1, ix->queries_per_launch);
    std::vector<szg::ScanArgs> args((nq + qpl - 1) / qpl);
    for (int j = 0; j < nq; j += qpl) {  // one sweep per query, results side by side
        szg::ScanArgs &a = args[j / qpl];
        fill_scan_args(ix, sh, c, has_allow, j, std::min(qpl, nq - j), &a);
        if (masked && ix->mask_dense) {
            double lowest = 1.0;
            for (int i = j; i < std::min(nq, j + qpl); i++) lowest = std::min(lowest, pass_rate(i));
            a.mask_dense = lowest >= 0.5 ? 1 : 0;
        }
        a.kp = kp;
        a.block_lists = c->d_lists_a + (size_t)j * g.grid * kp;
    }
    rc = launch_scans_chained(ix, sh, c, args, g);
    if (rc) return rc;

    int n_lists = g.grid;
    uint64_t *src = c->d_lists_a, *dst = c->d_lists_b;
    const int fan = szg::merge_fan(kp);
    {
        SiteScope t_(6);
        while (n_lists > 1) {
            HIPCHK(szg::launch_merge(src, n_lists, kp, nq, dst, c->stream));
            n_lists = (n_lists + fan - 1) / fan;
            std::swap(src, dst);
        }
    }
    {
        SiteScope t_(7);
        HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64, src,
                                  nullptr, (uint32_t)kp, nq, c->d_out, c->stream));
    }
    {
        SiteScope t_(8);
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * kp * nq,
                              hipMemcpyDeviceToHost, c->stream));
    }
    if (ix->timing >= 2) {
        SiteScope t_(10);
        HIPCHK(hipEventRecord(c->ev_all1, c->stream));
    }
    return SZG_OK;
}

// ---- multi-query sweep (32-bit rows, cosine): B queries share one pass ------------

bool mq_uses_i8(const szg_index *ix) { return (ix->bits == 8 || ix->bits == 4) && ix->mq_i8; }
// 32-bit rows of whole 64-byte steps: the bfloat16 sweep
bool mq_uses_bf16(const szg_index *ix)
{
    return ix->bits == 32 && ix->mq_bf16 && ix->map.r16 % 4 == 0 && ix->dim == ix->map.r16 * 4 && !ix->layout.tiled;
}

// round to nearest even, as v_cvt_pk_bf16_f32 does (NaN stays NaN)
uint16_t bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// the prepared real query of the integer sweeps: q/|q| (cosine) or maxInt*q (euclid)
double mq_int_scale(const szg_index *ix, double m1)
{
    if (ix->metric == SZG_COSINE) return m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0;
    return (double)((1u << ix->bits) - 1u);
}

// int8 sweep: quantization step, the integer query Q (dim values) and its digit sum
void prep_mq_int(const szg_index *ix, const double *q, QMeta *meta, int32_t *Qout)
{
    const double scale = mq_int_scale(ix, meta->m1);
    double vmax = 0.0;
    for (int e = 0; e < ix->dim; e++) vmax = std::max(vmax, std::fabs(q[e] * scale));
    const double Qmax = szg::kMqQmax;
    const double qs = (vmax > 0 && std::isfinite(vmax)) ? vmax / Qmax : 1.0;
    const double inv = scale / qs;
    long long sumQ = 0;
    for (int e = 0; e < ix->dim; e++) {
        const long long Q = round_clamp(q[e] * inv, Qmax);
        Qout[e] = (int32_t)Q;
        sumQ += Q;
    }
    meta->mq_int = true;
    meta->mq_qscale = qs;
    meta->mq_qconst = (double)sumQ;
}

int mq_blocks(const szg_index *ix, int nq)
{   // query blocks of 16 the batch gets (nq = the queries left in the call), or 0 when the shared sweep does not apply
    if (!ix->multi_query || ix->bits == 64 || nq < ix->mq_min) return 0;
    const bool bf16 = mq_uses_bf16(ix);
    int nb = std::min((nq + 15) / 16, std::min(ix->mq_blocks_max, bf16 ? 6 : 3));
    auto fits = [&](int n) {  // the image (+ tables, hit buffers, staging) must fit LDS
        if (bf16) return szg::mq_bf16_lds_bytes(ix->map.r16, n) <= 160u * 1024u;
        return (mq_uses_i8(ix) ? szg::mq_i8_lds_bytes(ix->bits, ix->map.r16, n)
                               : szg::mq_lds_bytes(ix->bits, ix->map.r16, n)) <= 150u * 1024u;
    };
    while (nb > 0 && !fits(nb)) nb--;
    return nb;
}

// top-k pass for the nq staged queries through ONE shared sweep:
// score matrix -> per-query selection -> merges -> rerank -> D2H (async)
// kp_wide: the list length when the lists hold bfloat16-sweep keys themselves (matrix form: small
// shards, overflow reruns), whose error band needs more candidates than kp
int enqueue_topk_mq(szg_index *ix, Shard *sh, Ctx *c, int kp, int kp_wide, int nq, int nb, bool has_allow,
                    bool force_matrix = false)
{
    HIPCHK(hipSetDevice(sh->device));
    const int r16 = ix->map.r16;
    const bool i8 = mq_uses_i8(ix);
    const bool bf16 = mq_uses_bf16(ix);
    // int8 sweeps: up to two groups of 16 * nb queries per launch (the kernel walks their passes back to back)
    const int groups = i8 ? (nq + 16 * nb - 1) / (16 * nb) : 1;
    const size_t group_stride = i8 ? ((szg::mq_i8_lds_bytes(ix->bits, r16, nb) + 255) & ~(size_t)255) : 0;
    const size_t img = bf16 ? szg::mq_bf16_image_bytes(r16, nb)
                            : i8 ? group_stride * groups : szg::mq_lds_bytes(ix->bits, r16, nb);
    int rc = ensure_host(&c->h_mq, &c->h_mq_cap, img);
    if (rc) return rc;
    rc = ensure_dev(&c->d_mq, &c->d_mq_cap, img);
    if (rc) return rc;
    memset(c->h_mq, 0, img);
    if (bf16) {
        // [32-element step][query block][lane = k-group*16 + query][8 bf16 = elements 8*k-group + 0..7
        // of the step]; cosine: q/|q|.
        uint16_t *im = reinterpret_cast<uint16_t *>(c->h_mq);
        for (int q = 0; q < nq; q++) {
            const double *src = c->h_q64 + (size_t)q * ix->dim;
            const double m1 = c->meta[q].m1;
            double scale = m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0;
            if (ix->metric != SZG_COSINE) scale = 1.0;
            const int b = q / 16, qi = q % 16;
            for (int e = 0; e < ix->dim; e++) {
                const int S = e >> 5, w = e & 31;
                im[((((size_t)S * nb + b) * 64) + (w >> 3) * 16 + qi) * 8 + (w & 7)] = bf16_rne((float)(src[e] * scale));
            }
        }
    } else if (i8) {
        // [64-byte step][digit plane h..l][T halves][query block][lane = chunk*16 + query][16 bytes] of the
        // int8 digits of Q = round(v / mq_qscale), then the table [qscale | qconst | qnorm2][48].
        // 8-bit rows (T = 1): byte i of a lane's word belongs to element 16*piece + i; the row
        // operand is v' = v - 128 and n = 2v' + 1, so sum Q n = 2 sum Q v' + sum Q.
        // 4-bit rows (T = 2: even | odd elements): byte bi belongs to element 32*piece + 2*bi
        // (+1 for the odd half); the operand is the nibble x and n = 2x - 15.
        const int NP = szg::kMqPlanes, T = ix->bits == 4 ? 2 : 1;
        const int epp = ix->bits == 4 ? 32 : 16;  // elements per 16-byte piece
        const size_t plane = (size_t)T * nb * 64 * 16;  // bytes between digit planes of a step
        for (int q = 0; q < nq; q++) {
            const int32_t *Qv = c->h_mqQ + (size_t)q * ix->dim;
            const int ql = q % (16 * nb);  // position inside its group
            uint8_t *im8 = c->h_mq + (size_t)(q / (16 * nb)) * group_stride;
            const int b = ql / 16, qi = ql % 16;
            for (int e = 0; e < ix->dim; e++) {
                int Q = Qv[e];
                const int j = e / epp, i = e % epp;
                const int bi = T == 2 ? i >> 1 : i, half = T == 2 ? i & 1 : 0;
                const int s = j >> 2, ch = j & 3;
                uint8_t *dst = im8 + ((((size_t)s * NP * T + half) * nb + b) * 64 + ch * 16 + qi) * 16 + bi;
                for (int p = NP - 1; p > 0; p--) {   // low digits first, balanced in [-64, 63]
                    const int dig = ((Q + 64) & 127) - 64;
                    Q = (Q - dig) >> 7;
                    dst[(size_t)p * plane] = (uint8_t)(int8_t)dig;
                }
                dst[0] = (uint8_t)(int8_t)Q;         // plane 0 = the top digit
            }
        }
        for (int q = 0; q < nq; q++) {
            const int ql = q % (16 * nb);
            float *tab = reinterpret_cast<float *>(c->h_mq + (size_t)(q / (16 * nb)) * group_stride +
                                                   szg::mq_i8_image_bytes(ix->bits, r16, nb));
            tab[ql] = (float)c->meta[q].mq_qscale;
            tab[48 + ql] = (float)((ix->bits == 4 ? -15.0 : 1.0) * c->meta[q].mq_qconst);
            tab[96 + ql] = (float)c->meta[q].qnorm2;
        }
    } else {
    // LDS image [piece j][query block][group of 4 elements][query 16][4 floats].  Cosine:
    // the normalised queries (q / |q|, so the key is -cos; quantized rows decode to
    // n = maxInt * d and the common factor cancels).  Euclid: maxInt * q for quantized
    // rows (key = |n - maxInt q|^2 = maxInt^2 |d - q|^2, the single-query path's unit).
    float *im = reinterpret_cast<float *>(c->h_mq);
    const int E = 128 / ix->bits, G4 = E / 4;
    for (int q = 0; q < nq; q++) {
        const double *src = c->h_q64 + (size_t)q * ix->dim;
        const double m1 = c->meta[q].m1;
        double scale = m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0;
        if (ix->metric != SZG_COSINE) scale = ix->bits <= 16 ? (double)((1u << ix->bits) - 1u) : 1.0;
        const int b = q / 16, qi = q % 16;
        for (int e = 0; e < ix->dim; e++) {
            const int j = e / E, i = e % E, g4 = i / 4, m = i % 4;
            im[((((size_t)j * nb + b) * G4 + g4) * 16 + qi) * 4 + m] = (float)(src[e] * scale);
        }
    }
    }
    HIPCHK(hipMemcpyAsync(c->d_mq, c->h_mq, img, hipMemcpyHostToDevice, c->stream));

    // Fused selection: sweep a prefix of the rows into a small score matrix, take each
    // query's kp-th best key there as its threshold, then sweep everything and collect the
    // (query, row) pairs at or below their threshold -- about `hits` per query -- instead of
    // writing and re-reading n_rows x batch keys.  Every row outside a query's buffer has a
    // key above the threshold, which is >= the kp-th kept key: certification is unchanged.
    // bfloat16 sweep: the collected candidates are scored again in float32 before the selection
    // (two stages); rows outside the buffer are bounded by the bfloat16 threshold, rows inside it by
    // the float32 keys.  In matrix form its lists hold bfloat16 keys and are kp_wide long.
    const uint64_t hits = std::max<uint64_t>((uint64_t)ix->mq_hits, 16ull * kp);
    uint64_t prefix = ((sh->n_rows * (uint64_t)kp + hits - 1) / hits + 15) & ~15ull;
    prefix = std::max<uint64_t>(prefix, 16ull * kp);
    const bool fused = ix->mq_fused && !force_matrix && prefix * 4 <= sh->n_rows;
    const bool stage2 = bf16 && fused;
    if (bf16 && !fused) kp = std::max(kp, kp_wide);
    const uint32_t cand_cap = (uint32_t)(4 * hits);
    const size_t key_stride = fused ? (size_t)prefix : (((size_t)sh->n_rows + 3) & ~(size_t)3);

    const int sb = 16;  // select blocks per query
    const size_t need = (size_t)nq * sb * kp;
    if (c->lists_cap < need) {
        if (c->d_lists_a) HIPCHK(hipFree(c->d_lists_a));
        if (c->d_lists_b) HIPCHK(hipFree(c->d_lists_b));
        c->d_lists_a = c->d_lists_b = nullptr;
        c->lists_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_lists_a, need * sizeof(uint64_t)));
        HIPCHK(hipMalloc((void **)&c->d_lists_b, need * sizeof(uint64_t)));
        c->lists_cap = need;
    }
    rc = ensure_dev(&c->d_out, &c->d_out_cap, (size_t)nq * kp);
    if (rc) return rc;
    rc = ensure_host(&c->h_out, &c->h_out_cap, (size_t)nq * kp);
    if (rc) return rc;
    rc = ensure_dev(&c->d_keys, &c->keys_cap, key_stride * nq);
    if (rc) return rc;
    if (fused) {
        if (!c->d_thr) HIPCHK(hipMalloc((void **)&c->d_thr, 128 * sizeof(float)));
        if (!c->d_cand_count) HIPCHK(hipMalloc((void **)&c->d_cand_count, 128 * szg::kCandCountStride * sizeof(uint32_t)));
        if (!c->h_cand_count)
            HIPCHK(hipHostMalloc((void **)&c->h_cand_count, 128 * szg::kCandCountStride * sizeof(uint32_t), hipHostMallocDefault));
        rc = ensure_dev(&c->d_cand, &c->cand_cap_total, (size_t)cand_cap * nq);
        if (rc) return rc;
    }
    if (stage2) {
        if (!c->h_thr) HIPCHK(hipHostMalloc((void **)&c->h_thr, 128 * sizeof(float), hipHostMallocDefault));
        if (!c->h_qscale) HIPCHK(hipHostMalloc((void **)&c->h_qscale, 128 * sizeof(double), hipHostMallocDefault));
        if (!c->d_qscale) HIPCHK(hipMalloc((void **)&c->d_qscale, 128 * sizeof(double)));
        for (int q = 0; q < nq; q++) {
            const double m1 = c->meta[q].m1;
            c->h_qscale[q] = ix->metric == SZG_COSINE ? (m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0) : 1.0;
        }
        HIPCHK(hipMemcpyAsync(c->d_qscale, c->h_qscale, sizeof(double) * nq, hipMemcpyHostToDevice, c->stream));
    }
    c->mq_fused_used = fused;
    c->mq_cand_cap = cand_cap;
    c->mq_nb = nb;
    c->mq_has_allow = has_allow;
    c->kp_used = kp;
    c->mq_stage2 = stage2;
    c->mq_bf16_used = bf16 && !stage2;

    szg::MqArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = sh->rows;
    a.n_rows = (uint32_t)sh->n_rows;
    a.pitch = ix->pitch;
    a.tiled = ix->layout.tiled;
    a.steps = ix->layout.steps;
    a.r16 = r16;
    a.dim = ix->dim;
    a.queries = c->d_mq;
    a.n_queries = nq;
    a.n_groups = groups;
    a.group_stride = (uint32_t)group_stride;
    a.metric = ix->metric;
    for (int q = 0; q < nq && q < szg::kMqMaxQueries; q++) a.qnorm2[q] = (float)c->meta[q].qnorm2;
    a.keys = c->d_keys;
    a.key_stride = key_stride;
    a.zero16 = sh->zero16;
    a.norm_bias = (float)ix->norm_bias;
    // The sweep wants every CU to itself (one 1024-thread block and up to 144 KiB of
    // LDS per CU), so the whole batch -- sweep, selection, merges, rerank, copy --
    // goes onto the shard's scan stream, one batch after the other; only uploads
    // overlap on the context's stream.
    {
        std::lock_guard<std::mutex> lk(sh->chain_mu);
        hipStream_t st = ix->serialize_scans ? sh->scan_stream : c->stream;
        const bool overlap = (bf16 || i8) && ix->mq_overlap && st != c->stream;  // the HBM-bound sweeps
        // (overlap: the threshold pass goes ahead on the context's stream, the sweep follows on the scan stream)
        hipStream_t head = overlap ? c->stream : st;
        if (st != c->stream && !overlap) {
            HIPCHK(hipEventRecord(c->ev_up, c->stream));
            HIPCHK(hipStreamWaitEvent(st, c->ev_up, 0));
        }
        auto launch_score = [&](const szg::MqArgs &x, hipStream_t s2) -> hipError_t {
            if (bf16) return szg::launch_mq_score_bf16(x, nb, sh->cu_count, s2);
            return i8 ? szg::launch_mq_score_i8(ix->bits, x, nb, sh->cu_count, s2)
                      : szg::launch_mq_score(ix->bits, x, nb, sh->cu_count, s2);
        };
        // score matrix of rows [0, n_sel) -> per-query sorted list of kp (returns its buffer)
        auto select_chain = [&](uint32_t n_sel, size_t kstride, hipStream_t s2, uint64_t **out) -> hipError_t {
            hipError_t e = szg::launch_mq_select(c->d_keys, kstride, n_sel, sh->has_dead ? sh->live_bits : nullptr,
                                                 has_allow ? c->d_allow : nullptr, (uint32_t)shard_words(sh),
                                                 kp, nq, sb, c->d_lists_a, s2);
            int n_lists = sb;
            uint64_t *src = c->d_lists_a, *dst = c->d_lists_b;
            const int fan = szg::merge_fan(kp);
            while (e == hipSuccess && n_lists > 1) {
                e = szg::launch_merge(src, n_lists, kp, nq, dst, s2);
                n_lists = (n_lists + fan - 1) / fan;
                std::swap(src, dst);
            }
            *out = src;
            return e;
        };
        uint64_t *src = nullptr;
        if (fused) {
            szg::MqArgs pa = a;  // the prefix, into the (small) score matrix
            pa.n_rows = (uint32_t)prefix;
            HIPCHK(launch_score(pa, head));
            // one block per query selects over the prefix's keys, publishes the query's
            // threshold and zeroes its hit counter
            HIPCHK(szg::launch_mq_select(c->d_keys, key_stride, (uint32_t)prefix,
                                         sh->has_dead ? sh->live_bits : nullptr, has_allow ? c->d_allow : nullptr,
                                         (uint32_t)shard_words(sh), kp, nq, 1, c->d_lists_a, head, c->d_thr,
                                         c->d_cand_count));
            a.collect = 1;
            a.thr = c->d_thr;
            a.cand_buf = c->d_cand;
            a.cand_count = c->d_cand_count;
            a.cand_cap = cand_cap;
            a.live_bits = sh->has_dead ? sh->live_bits : nullptr;
            a.allow_bits = has_allow ? c->d_allow : nullptr;
            a.allow_stride = (uint32_t)shard_words(sh);
        }
        if (overlap) {
            HIPCHK(hipEventRecord(c->ev_up, c->stream));
            HIPCHK(hipStreamWaitEvent(st, c->ev_up, 0));
        }
        if (ix->timing) HIPCHK(hipEventRecord(c->ev_scan0, st));  // the full sweep (not the prefix pass)
        HIPCHK(launch_score(a, st));
        if (ix->timing) {
            HIPCHK(hipEventRecord(c->ev_scan1, st));
            c->timed_scan = true;
            c->timed_n = groups;  // (passes: an int8 launch may walk two)
        }
        // The selection, merges, rerank and copy-back of this batch either follow on the scan
        // stream (default) or, with "mq_tail_overlap", on the context's stream, where they run
        // beside the NEXT batch's sweep (the sweep is MFMA-bound and leaves wave slots free).
        hipStream_t tail = st;
        if ((ix->mq_tail_overlap || overlap) && st != c->stream) {
            HIPCHK(hipEventRecord(c->ev_scan_done, st));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_scan_done, 0));
            tail = c->stream;
        }
        if (stage2) {
            HIPCHK(szg::launch_cand_rescore(ix->metric, sh->rows, ix->pitch, ix->dim, c->d_q64, c->d_qscale, c->d_cand,
                                            c->d_cand_count, cand_cap, nq, tail));
            HIPCHK(hipMemcpyAsync(c->h_thr, c->d_thr, 128 * sizeof(float), hipMemcpyDeviceToHost, tail));
        }
        if (fused) {
            HIPCHK(szg::launch_cand_select(c->d_cand, c->d_cand_count, cand_cap, kp, nq, c->d_lists_a, tail));
            src = c->d_lists_a;
            HIPCHK(hipMemcpyAsync(c->h_cand_count, c->d_cand_count, 128 * szg::kCandCountStride * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, tail));
        } else {
            HIPCHK(select_chain((uint32_t)sh->n_rows, key_stride, tail, &src));
        }
        HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64, src,
                                  nullptr, (uint32_t)kp, nq, c->d_out, tail));
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * kp * nq,
                              hipMemcpyDeviceToHost, tail));
        if (tail != c->stream) {
            HIPCHK(hipEventRecord(c->ev_scan_done, st));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_scan_done, 0));
        }
    }
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->stats.scan_launches += (uint64_t)groups;
        ix->stats.scan_bytes += (uint64_t)groups * sh->n_rows * (uint64_t)ix->row_bytes;  // ONE pass per group of the batch
        ix->stats.mq_launches += (uint64_t)groups;
        ix->stats.mq_queries += (uint64_t)nq;
        ix->stats.mq_bf16_sweeps += bf16 ? 1 : 0;
    }
    if (ix->timing >= 2) HIPCHK(hipEventRecord(c->ev_all1, c->stream));
    return SZG_OK;
}

int finish_timing(szg_index *ix, Ctx *c)
{
    if (!ix->timing) return SZG_OK;
    float ms_scan = 0, ms_all = 0;
    if (c->timed_scan) HIPCHK(hipEventElapsedTime(&ms_scan, c->ev_scan0, c->ev_scan1));
    if (ix->timing >= 2) HIPCHK(hipEventElapsedTime(&ms_all, c->ev_all0, c->ev_all1));
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    if (c->timed_scan) {
        ix->stats.scan_ms += ms_scan;
        ix->stats.timed_launches += c->timed_n;
    }
    ix->stats.total_ms += ms_all;
    c->timed_scan = false;
    return SZG_OK;
}

// candidates of staged query `slot` from a finished top-k pass, each with the upper bound of its
// real-number key; *lb = a lower bound of the real-number key of every eligible row of the shard that
// is NOT among them (+inf if every eligible row is).  `m` = the query's constants with the flags
// of the path the ticket was prepared for; the shard's context says which arithmetic actually
// produced the keys.
void gather_topk(const szg_index *ix, const Shard *sh, const Ctx *c, const QMeta &m, int slot,
                 std::vector<Cand> *cands, double *lb)
{
    const int kp = c->kp_used;
    QMeta lm = m;  // class of the list's keys
    if (m.mq) lm.mq_bf16 = c->mq_bf16_used;
    int valid = 0;
    float worst = -INFINITY;
    for (int i = 0; i < kp; i++) {
        const szg::RerankOut &r = c->h_out[(size_t)slot * kp + i];
        if (r.row == 0xFFFFFFFFu) continue;
        valid++;
        const float key = szg::key_from_ordered(r.ukey);
        worst = std::max(worst, key);
        double ub = (double)key + key_eps(ix, key, lm);
        // a row forced in (key -2: float32 norm under- or overflowed) carries no information in its key;
        // its float64 distance does: -cos(pi d) is the real-number key
        if (ix->metric == SZG_COSINE && key <= -1.5f && !std::isnan(r.dist)) ub = -std::cos(M_PI * r.dist) + 1e-9;
        cands->push_back(Cand{sh->first + r.row, r.dist, key, ub});
    }
    *lb = valid == kp ? (double)worst - key_eps(ix, worst, lm) : INFINITY;
    if (c->mq_stage2) {
        // rows the bfloat16 sweep did not collect: bfloat16 key above the prefix threshold
        const float thr = c->h_thr[slot];
        if (thr < 3.0e38f) {
            QMeta bm = m;
            bm.mq_bf16 = true;
            *lb = std::min(*lb, (double)thr - key_eps(ix, thr, bm));
        }
    }
}

// collect pass (radius search / escalation) for staged query `slot`: every row
// with key <= thr_key, reranked exactly.  Synchronous; grows the buffer and
// reruns on overflow.
int run_collect(szg_index *ix, Shard *sh, Ctx *c, int slot, float thr_key, bool has_allow,
                std::vector<Cand> *cands)
{
    HIPCHK(hipSetDevice(sh->device));
    if (sh->n_rows == 0) return SZG_OK;
    size_t want = std::max<size_t>(c->collect_cap, 1u << 16);
    for (;;) {
        int rc = ensure_dev(&c->d_collect, &c->collect_cap, want);
        if (rc) return rc;
        if (ix->timing >= 2) HIPCHK(hipEventRecord(c->ev_all0, c->stream));
        HIPCHK(hipMemsetAsync(c->d_count, 0, sizeof(uint32_t), c->stream));
        HIPCHK(hipEventRecord(c->ev_up, c->stream));  // the sweep must see the zeroed counter
        std::vector<szg::ScanArgs> a(1);
        fill_scan_args(ix, sh, c, has_allow, slot, 1, &a[0]);
        a[0].collect = 1;
        a[0].thr_ukey = szg::ordered_key(thr_key);
        a[0].collect_buf = c->d_collect;
        a[0].collect_cap = (uint32_t)std::min<size_t>(c->collect_cap, 0xFFFFFFFFu);
        a[0].collect_count = c->d_count;
        const LaunchGeom g = scan_geometry(ix, sh, 0);
        rc = launch_scans_chained(ix, sh, c, a, g);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->h_count, c->d_count, sizeof(uint32_t), hipMemcpyDeviceToHost,
                              c->stream));
        if (ix->timing >= 2) HIPCHK(hipEventRecord(c->ev_all1, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rc = finish_timing(ix, c);
        if (rc) return rc;
        const uint32_t count = c->h_count[0];
        if (count > c->collect_cap) {
            want = (size_t)count + count / 8 + 1024;
            continue;
        }
        if (count == 0) return SZG_OK;
        rc = ensure_dev(&c->d_out, &c->d_out_cap, (size_t)count);
        if (rc) return rc;
        rc = ensure_host(&c->h_out, &c->h_out_cap, (size_t)count);
        if (rc) return rc;
        HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim,
                                  c->d_q64 + (size_t)slot * ix->dim, c->d_collect, nullptr, count, 1,
                                  c->d_out, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * count,
                              hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        cands->reserve(cands->size() + count);
        for (uint32_t i = 0; i < count; i++) {
            const szg::RerankOut &r = c->h_out[i];
            cands->push_back(Cand{sh->first + r.row, r.dist, szg::key_from_ordered(r.ukey), 0.0});
        }
        return SZG_OK;
    }
}

// consider()'s top-k branch replayed over the candidates in visit order
// (collection.go:606-619), then the ascending pop loop (:694-697).
void replay_topk(std::vector<Cand> &cands, int k, std::vector<HeapItem> *result)
{
    std::sort(cands.begin(), cands.end(), [](const Cand &x, const Cand &y) { return x.row < y.row; });
    GoHeap h;
    for (const Cand &c : cands) h.consider_topk(c.row, c.dist, k);
    h.drain(result);
}

// True when the reference's answer may depend on its whole heap history: a NaN
// distance, or two exactly equal distances among the best k+1 candidates.
bool history_dependent(const double *dist, size_t n, int k)
{
    std::vector<double> d;
    d.reserve(n);
    for (size_t i = 0; i < n; i++) {
        if (std::isnan(dist[i])) return true;
        d.push_back(dist[i]);
    }
    const size_t m = std::min(d.size(), (size_t)k + 1);
    std::partial_sort(d.begin(), d.begin() + m, d.end());
    for (size_t i = 1; i < m; i++)
        if (d[i] == d[i - 1]) return true;
    return false;
}

// Exact replay of the reference loop over EVERY row (collection.go:672-684 with
// consider(), :583-629): float64 distances for all rows on the device, then the
// heap on the host in visit order.  Bit-faithful in every case, used only when
// history_dependent() says the fast answer could differ.
int run_full_replay(szg_index *ix, std::vector<Ctx *> &ctx, int slot, const uint64_t *allow, int k,
                    std::vector<HeapItem> *res)
{
    GoHeap h;
    for (size_t s = 0; s < ix->shards.size(); s++) {
        Shard *sh = ix->shards[s];
        if (sh->n_rows == 0) continue;
        Ctx *c = ctx[s];
        HIPCHK(hipSetDevice(sh->device));
        const size_t n = sh->n_rows;
        int rc = ensure_dev(&c->d_out, &c->d_out_cap, n);
        if (rc) return rc;
        rc = ensure_host(&c->h_out, &c->h_out_cap, n);
        if (rc) return rc;
        HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim,
                                  c->d_q64 + (size_t)slot * ix->dim, nullptr, nullptr, (uint32_t)n, 1,
                                  c->d_out, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * n, hipMemcpyDeviceToHost,
                              c->stream));
        std::vector<uint64_t> live((n + 63) / 64, ~0ull);
        if (sh->has_dead)
            HIPCHK(hipMemcpyAsync(live.data(), sh->live_bits, live.size() * sizeof(uint64_t),
                                  hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const uint64_t *aw = allow ? allow + sh->first / 64 : nullptr;
        for (size_t r = 0; r < n; r++) {
            if (!((live[r >> 6] >> (r & 63)) & 1)) continue;       // removed record
            if (aw && !((aw[r >> 6] >> (r & 63)) & 1)) continue;   // collection.go:592-594
            h.consider_topk(sh->first + r, c->h_out[r].dist, k);
        }
    }
    h.drain(res);
    return SZG_OK;
}

// The first k eligible rows of a query in visit order are pushed by consider() whatever their
// distance (collection.go:608, `len < K`); a NaN among them -- an antipodal or parallel row
// under the unclamped acos (:831), a NaN / Inf element -- sits in the reference's heap and
// decides what is accepted afterwards.  Such a row need not be anywhere near the best keys,
// so the scan's candidates do not show it: the exact distances of these k rows are computed
// beside every batch and a NaN sends the query to the exact replay.
// rows_out: index-level rows, ascending; at most k.
void first_eligible_rows(const szg_index *ix, const uint64_t *allow, int k, std::vector<uint64_t> *rows_out)
{
    rows_out->clear();
    for (const Shard *sh : ix->shards) {
        if ((int)rows_out->size() >= k) break;
        if (sh->n_rows == 0) continue;
        if (!allow && !sh->has_dead) {
            for (uint64_t r = 0; r < sh->n_rows && (int)rows_out->size() < k; r++) rows_out->push_back(sh->first + r);
            continue;
        }
        const uint64_t words = (sh->n_rows + 63) / 64;
        const uint64_t *aw = allow ? allow + sh->first / 64 : nullptr;
        for (uint64_t w = 0; w < words && (int)rows_out->size() < k; w++) {
            uint64_t m = sh->live_host[w];
            if (aw) m &= aw[w];
            const uint64_t left = sh->n_rows - w * 64;
            if (left < 64) m &= (1ull << left) - 1ull;
            while (m && (int)rows_out->size() < k) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                rows_out->push_back(sh->first + w * 64 + (uint64_t)b);
            }
        }
    }
}

// Stage the sentinel rows of the batch that fall into this shard and enqueue their float64
// distances on the ctx stream (lists: one vector of index-level rows per staged query).
int enqueue_sentinels(szg_index *ix, Shard *sh, Ctx *c, const std::vector<std::vector<uint64_t>> &lists, int nq)
{
    c->sent_n = 0;
    size_t most = 0;
    for (int j = 0; j < nq; j++) {
        size_t n = 0;
        for (uint64_t r : lists[j]) n += (r >= sh->first && r < sh->first + sh->n_rows) ? 1 : 0;
        most = std::max(most, n);
    }
    if (most == 0) return SZG_OK;
    SiteScope t_(9);
    HIPCHK(hipSetDevice(sh->device));
    const size_t total = most * (size_t)nq;
    int rc = ensure_host(&c->h_sent, &c->h_sent_cap, total);
    if (rc) return rc;
    rc = ensure_dev(&c->d_sent, &c->d_sent_cap, total);
    if (rc) return rc;
    rc = ensure_host(&c->h_sent_out, &c->h_sent_out_cap, total);
    if (rc) return rc;
    rc = ensure_dev(&c->d_sent_out, &c->d_sent_out_cap, total);
    if (rc) return rc;
    for (int j = 0; j < nq; j++) {
        size_t n = 0;
        for (uint64_t r : lists[j])
            if (r >= sh->first && r < sh->first + sh->n_rows) c->h_sent[(size_t)j * most + n++] = r - sh->first;
        for (; n < most; n++) c->h_sent[(size_t)j * most + n] = szg::kInvalidCand;
    }
    HIPCHK(hipMemcpyAsync(c->d_sent, c->h_sent, total * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64, c->d_sent, nullptr,
                              (uint32_t)most, nq, c->d_sent_out, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_sent_out, c->d_sent_out, total * sizeof(szg::RerankOut), hipMemcpyDeviceToHost,
                          c->stream));
    c->sent_n = (int)most;
    return SZG_OK;
}

double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Ticket {
    int first = 0, nq = 0;       // queries [first, first+nq) of the call
    std::vector<Ctx *> ctx;      // one per shard
    std::vector<QMeta> meta;
    int kp = 0, kp_wide = 0;
    bool failed = false;         // enqueueing failed part-way: drain and release only
    bool any_mask = false;       // some query of the batch carries a filter mask
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
            (void)hipStreamSynchronize(ctx[s]->stream);
            ctx[s]->mq_fused_used = false;
            ctx_release(owner->shards[s], ctx[s]);
        }
    }
};

// allow_bits: n_queries masks back to back, or nullptr; allow_ptrs (used instead when given):
// one mask pointer per query, null entries unfiltered.
int search_topk_impl(szg_index *ix, const double *queries, int n_queries, int k,
                     const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist,
                     int32_t *out_count, const uint64_t *const *allow_ptrs = nullptr)
{
    const size_t n_sh = ix->shards.size();
    uint64_t total_rows = 0;
    for (Shard *s : ix->shards) total_rows += s->n_rows;
    const size_t allow_stride = (total_rows + 63) / 64;
    auto mask_of = [&](int qi) -> const uint64_t * {
        if (allow_ptrs) return allow_ptrs[qi];
        return allow_bits ? allow_bits + (size_t)qi * allow_stride : nullptr;
    };
    int kp = k + std::max(ix->slack_min, k / 2);
    // The reference bounds K by nothing (collection.go:606-619).  The fused selection keeps kp
    // candidates per wave in LDS; beyond that (kp > 4096 or 64 KiB of lists) every query of the
    // call takes the exact replay: float64 distances of all rows
    // on the device, consider() over them on the host.
    bool replay_all = false;
    for (Shard *s : ix->shards) {
        if (s->n_rows == 0 || replay_all) continue;
        const LaunchGeom g = scan_geometry(ix, s, kp);
        if (szg::scan_lds_bytes(ix->bits, ix->map, kp, g.block) > 64u * 1024u || kp > 4096) replay_all = true;
    }
    if (replay_all) kp = 1;  // the batches only stage their queries

    // result assembly for one finished batch
    auto finish = [&](Ticket &t) -> int {
        int rc = SZG_OK;
        if (t.failed) {
            for (size_t s = 0; s < n_sh; s++) {
                if (!t.ctx[s]) continue;
                (void)hipSetDevice(ix->shards[s]->device);
                (void)hipStreamSynchronize(t.ctx[s]->stream);
                t.ctx[s]->mq_fused_used = false;
                ctx_release(ix->shards[s], t.ctx[s]);
            }
            t.ctx.assign(n_sh, nullptr);
            return SZG_OK;  // the enqueue error is already the call's return code
        }
        for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
            Shard *sh = ix->shards[s];
            if (sh->n_rows == 0) continue;
            hipError_t e = hipSetDevice(sh->device);
            if (e == hipSuccess) e = hipStreamSynchronize(t.ctx[s]->stream);
            if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipStreamSynchronize", e);
            if (rc == SZG_OK) rc = finish_timing(ix, t.ctx[s]);
            // fused selection: a query whose candidate buffer overflowed (threshold from the
            // prefix too loose: duplicates, sorted corpora) sends the batch down the matrix path
            Ctx *c = t.ctx[s];
            if (rc == SZG_OK && c->mq_fused_used) {
                bool overflow = false;
                for (int j = 0; j < t.nq; j++) overflow |= c->h_cand_count[j * szg::kCandCountStride] > c->mq_cand_cap;
                c->mq_fused_used = false;
                if (overflow) {
                    {
                        std::lock_guard<std::mutex> lk(ix->stats_mu);
                        ix->stats.mq_launches -= (uint64_t)((t.nq + 16 * c->mq_nb - 1) / (16 * c->mq_nb));  // counted again by the rerun
                        ix->stats.mq_queries -= (uint64_t)t.nq;
                        ix->stats.mq_bf16_sweeps -= (c->mq_stage2 || c->mq_bf16_used) ? 1 : 0;
                        ix->stats.mq_fallbacks += 1;
                    }
                    rc = enqueue_topk_mq(ix, sh, c, t.kp, t.kp_wide, t.nq, c->mq_nb, c->mq_has_allow, true);
                    if (rc == SZG_OK) {
                        e = hipStreamSynchronize(c->stream);
                        if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipStreamSynchronize", e);
                    }
                    if (rc == SZG_OK) rc = finish_timing(ix, c);
                }
            }
        }
        const double t_fin0 = now_us();
        double t_dev = 0;  // time spent waiting on escalation / replay passes (device work)
        // gather every query's candidates first: the escalation and replay paths
        // below reuse the contexts' output buffers
        std::vector<std::vector<Cand>> all(t.nq);
        std::vector<double> thr_min(t.nq, INFINITY);
        std::vector<uint8_t> nan_first(t.nq, 0);  // a NaN distance among the query's first k eligible rows
        if (rc == SZG_OK && !replay_all) {
            for (int j = 0; j < t.nq; j++) {
                for (size_t s = 0; s < n_sh; s++) {
                    Shard *sh = ix->shards[s];
                    if (sh->n_rows == 0) continue;
                    double lb;
                    gather_topk(ix, sh, t.ctx[s], t.meta[j], j, &all[j], &lb);
                    thr_min[j] = std::min(thr_min[j], lb);
                    const Ctx *c = t.ctx[s];
                    for (int i = 0; i < c->sent_n; i++) {
                        const szg::RerankOut &r = c->h_sent_out[(size_t)j * c->sent_n + i];
                        if (r.row != 0xFFFFFFFFu && std::isnan(r.dist)) nan_first[j] = 1;
                    }
                }
            }
        }
        for (int j = 0; j < t.nq && rc == SZG_OK; j++) {
            const int qi = t.first + j;
            const uint64_t *allow = mask_of(qi);
            std::vector<Cand> &cands = all[j];
            std::vector<HeapItem> res;
            if (replay_all) {
                const double td = now_us();
                rc = run_full_replay(ix, t.ctx, j, allow, k, &res);
                t_dev += now_us() - td;
                if (rc == SZG_OK) {
                    std::lock_guard<std::mutex> lk(ix->stats_mu);
                    ix->stats.full_replays++;
                }
            } else {
            // A NaN distance outside the query's first k eligible rows never enters the reference's heap
            // (`distance < worst` is false, collection.go:608-619); rows with an Inf / NaN element are forced
            // into the lists by the kernels (their float32 norm is not finite) and leave here.  A NaN among
            // the first k rows is the sentinels' business (nan_first: exact replay).
            auto drop_nan = [&](std::vector<Cand> &v) {
                if (nan_first[j]) return;
                v.erase(std::remove_if(v.begin(), v.end(), [](const Cand &c) { return std::isnan(c.dist); }), v.end());
            };
            drop_nan(cands);
            replay_topk(cands, k, &res);
            // certification: every row outside the lists has a real-number key >= thr_min (the
            // lists' own lower bound), so the result is final once the upper bound of its worst
            // key stays below that
            bool certified = true;
            double kmax = -INFINITY;  // upper bound of the real-number key of the worst result
            const bool zero_query = ix->metric == SZG_COSINE && t.meta[j].m1 == 0;  // all distances 1.0
            if (thr_min[j] < INFINITY && !zero_query) {
                std::vector<std::pair<uint64_t, double>> by_row;  // cands are sorted by row now
                by_row.reserve(cands.size());
                for (const Cand &c : cands) by_row.emplace_back(c.row, c.ub);
                for (const HeapItem &h : res) {
                    auto it = std::lower_bound(by_row.begin(), by_row.end(),
                                               std::make_pair(h.row, (double)-INFINITY));
                    kmax = std::max(kmax, it->second);
                }
                certified = (int)res.size() == k && kmax < thr_min[j];
            }
            if (ix->force_escalate && thr_min[j] < INFINITY) certified = false;
            if (nan_first[j] && ix->tie_mode == 0) certified = true;  // answered by the replay below
            if (!certified) {
                {
                    std::lock_guard<std::mutex> lk(ix->stats_mu);
                    ix->stats.escalations++;
                }
                double thr = INFINITY;
                if ((int)res.size() == k && std::isfinite(kmax) && !zero_query) {
                    // kmax bounds the worst result's real-number key; the collect sweep (always
                    // the single-query kernel) adds its own error on the rows it tests
                    QMeta single = t.meta[j];
                    single.mq = false;
                    single.mq_int = false;
                    single.mq_bf16 = false;
                    const double e2 = key_eps(ix, kmax, single);
                    thr = kmax + 1.05 * e2 + 0.05 * std::fabs(kmax) * 0x1p-20;
                }
                const float thr_f = !(thr < 3.0e38) ? 3.0e38f : std::nextafter((float)thr, INFINITY);
                cands.clear();
                const double td = now_us();
                for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
                    Shard *sh = ix->shards[s];
                    if (sh->n_rows == 0) continue;
                    rc = run_collect(ix, sh, t.ctx[s], j, thr_f, t.any_mask, &cands);
                }
                t_dev += now_us() - td;
                if (rc == SZG_OK) {
                    drop_nan(cands);
                    replay_topk(cands, k, &res);
                }
            }
            if (rc == SZG_OK && ix->tie_mode == 0) {
                std::vector<double> d(cands.size());
                for (size_t i = 0; i < cands.size(); i++) d[i] = cands[i].dist;
                if (nan_first[j] || history_dependent(d.data(), d.size(), k)) {
                    {
                        std::lock_guard<std::mutex> lk(ix->stats_mu);
                        ix->stats.full_replays++;
                    }
                    const double td = now_us();
                    rc = run_full_replay(ix, t.ctx, j, allow, k, &res);
                    t_dev += now_us() - td;
                }
            }
            }
            if (rc) break;
            for (int i = 0; i < k; i++) {
                const bool have = i < (int)res.size();
                out_rows[(size_t)qi * k + i] = have ? res[i].row + ix->row_base : UINT64_MAX;
                out_dist[(size_t)qi * k + i] = have ? res[i].priority : 0.0;
            }
            if (out_count) out_count[qi] = (int32_t)res.size();
        }
        {
            std::lock_guard<std::mutex> lk(ix->stats_mu);
            ix->stats.host_finish_us += now_us() - t_fin0 - t_dev;
        }
        for (size_t s = 0; s < n_sh; s++)
            if (t.ctx[s]) ctx_release(ix->shards[s], t.ctx[s]);
        t.ctx.assign(n_sh, nullptr);
        if (rc == SZG_OK) {
            std::lock_guard<std::mutex> lk(ix->stats_mu);
            ix->stats.queries += t.nq;
        }
        return rc;
    };

    std::deque<Ticket> inflight;
    int rc = SZG_OK;
    const int B1 = std::max(1, std::min(ix->query_batch, kMaxBatch));
    for (int q0 = 0; q0 < n_queries && rc == SZG_OK;) {
        Ticket t;
        t.owner = ix;
        t.first = q0;
        // batches of up to 32 share one sweep when the multi-query path applies
        const int left = n_queries - q0;
        const int nb = replay_all ? 0 : mq_blocks(ix, left);
        // (int8 sweeps: two groups of 48 per launch when that many queries are waiting)
        const int groups = nb == 3 && mq_uses_i8(ix) && ix->mq_i8_groups > 1 && left > 48 &&
                                   szg::mq_i8_lds_bytes(ix->bits, ix->map.r16, 3, 2) <= 160u * 1024u   // both images in LDS
                               ? 2 : 1;
        t.nq = nb ? std::min(left, 16 * nb * groups) : std::min(B1, left);
        const bool bf16_sweep = nb > 0 && mq_uses_bf16(ix);
        t.kp = kp;
        // lists of bfloat16-sweep keys (matrix form): the error band holds more rows than the
        // float32 one's, keep enough candidates for the k-th result to clear it
        t.kp_wide = bf16_sweep ? std::min(4096, std::max(kp, k + std::max(ix->mq_bf16_slack, k / 2))) : kp;
        t.ctx.assign(n_sh, nullptr);
        t.meta.assign(t.nq, QMeta{});
        // one context per shard; never block while holding in-flight work
        bool got = true;
        for (size_t s = 0; s < n_sh; s++) {
            if (ix->shards[s]->n_rows == 0) continue;
            Ctx *c = inflight.empty() ? ctx_acquire(ix->shards[s]) : ctx_try_acquire(ix->shards[s]);
            if (!c) {
                got = false;
                break;
            }
            t.ctx[s] = c;
        }
        if (!got) {
            for (size_t s = 0; s < n_sh; s++)
                if (t.ctx[s]) ctx_release(ix->shards[s], t.ctx[s]);
            t.ctx.assign(n_sh, nullptr);
            rc = finish(inflight.front());
            inflight.pop_front();
            continue;
        }
        const double *q = queries + (size_t)q0 * ix->dim;
        std::vector<const uint64_t *> masks(t.nq);
        for (int j = 0; j < t.nq; j++) {
            masks[j] = mask_of(q0 + j);
            t.any_mask |= masks[j] != nullptr;
        }
        const uint64_t *const *mptr = t.any_mask ? masks.data() : nullptr;
        const double t_prep0 = now_us();
        // the batch is prepared ONCE (swizzled / digit-plane forms, constants) into the first
        // shard's staging buffers; the other shards get copies
        Ctx *c0 = nullptr;
        const bool int_planes = nb > 0 && mq_uses_i8(ix);
        for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
            if (ix->shards[s]->n_rows == 0) continue;
            Ctx *cx = t.ctx[s];
            if (int_planes && !cx->h_mqQ) {
                cx->h_mqQ = (int32_t *)malloc(sizeof(int32_t) * (size_t)kMaxBatch * ix->dim);
                if (!cx->h_mqQ) {
                    rc = fail(SZG_E_NOMEM, "host scratch");  // the ticket is still finished below
                    break;
                }
            }
            if (!c0) {
                c0 = cx;
                for (int j = 0; j < t.nq; j++) {
                    prep_query(ix, q + (size_t)j * ix->dim, cx->h_qsw + (size_t)j * ix->qsw_bytes, &t.meta[j]);
                    t.meta[j].mq = nb > 0 && !mq_uses_i8(ix);  // the integer sweeps keep the integer bound
                    t.meta[j].mq_bf16 = bf16_sweep;
                    if (int_planes)
                        prep_mq_int(ix, q + (size_t)j * ix->dim, &t.meta[j], cx->h_mqQ + (size_t)j * ix->dim);
                    cx->meta[j] = t.meta[j];
                }
            } else {
                memcpy(cx->h_qsw, c0->h_qsw, ix->qsw_bytes * (size_t)t.nq);
                if (int_planes) memcpy(cx->h_mqQ, c0->h_mqQ, sizeof(int32_t) * (size_t)t.nq * ix->dim);
                for (int j = 0; j < t.nq; j++) cx->meta[j] = t.meta[j];
            }
        }
        // rows consider() pushes unconditionally: the first k eligible ones per query
        std::vector<std::vector<uint64_t>> sent;
        if (rc == SZG_OK && ix->tie_mode == 0 && !replay_all) {
            sent.resize(t.nq);
            for (int j = 0; j < t.nq; j++) {
                if (j > 0 && !masks[j] && !masks[j - 1]) sent[j] = sent[j - 1];
                else first_eligible_rows(ix, masks[j], k, &sent[j]);
            }
        }
        const double t_enq0 = now_us();
        for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
            Shard *sh = ix->shards[s];
            if (sh->n_rows == 0) continue;
            t.ctx[s]->sent_n = 0;
            rc = enqueue_queries(ix, sh, t.ctx[s], q, t.nq, mptr);
            // (before the sweeps: on the context's stream this runs while the scan stream sweeps)
            if (rc == SZG_OK && !sent.empty()) rc = enqueue_sentinels(ix, sh, t.ctx[s], sent, t.nq);
            if (rc == SZG_OK && !replay_all)
                rc = nb ? enqueue_topk_mq(ix, sh, t.ctx[s], t.kp, t.kp_wide, t.nq, nb, t.any_mask)
                        : enqueue_topk(ix, sh, t.ctx[s], kp, t.nq, t.any_mask);
            if (rc == SZG_OK && replay_all && ix->timing >= 2) {
                const hipError_t e = hipEventRecord(t.ctx[s]->ev_all1, t.ctx[s]->stream);
                if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipEventRecord", e);
            }
        }
        {
            std::lock_guard<std::mutex> lk(ix->stats_mu);
            const double t_end = now_us();
            ix->stats.host_prep_us += t_enq0 - t_prep0;
            ix->stats.host_enqueue_us += t_end - t_enq0;
        }
        t.failed = rc != SZG_OK;  // nothing to gather: finish() only drains and releases
        inflight.push_back(std::move(t));
        q0 += inflight.back().nq;
    }
    while (!inflight.empty()) {
        int r2 = finish(inflight.front());
        if (rc == SZG_OK) rc = r2;
        inflight.pop_front();
    }
    return rc;
}

// ---- 8-bit sketch pre-pass (float32 rows, cosine) ---------------------------------------------------------------
//
// The reference's "cosine" distance IS the angle (acos(cos)/pi, collection.go:821-832), a metric on directions:
// |d(q, x) - d(q, s)| <= d(x, s) for any sketch s of the row x.  The library keeps an 8-bit sketch of every float32
// row (a quarter of the bytes) as an internal 8-bit index, sweeps THAT for the K' nearest sketches (its own exact,
// certified answer: the whole machinery of this file on 8-bit rows), re-ranks those rows -- plus the query's first k
// rows and the rows that have no usable sketch -- on the float32 rows in float64, and replays consider() over them.
// With A = max over the rows of d(row, sketch) (measured when the sketch is built) and D = the K'-th sketch
// distance, every row that is not a candidate has d(q, sketch) >= D, hence d(q, x) >= D - A: the answer is final
// when its k-th distance is below that.  Otherwise -- and for equal distances or a NaN among the first k rows,
// where the reference's answer depends on its heap history -- the query takes the float32 path.
int reset_shards(szg_index *ix, const std::vector<uint64_t> &counts);
int shard_reserve(szg_index *ix, Shard *sh, uint64_t rows_needed);
int shard_set_live(Shard *sh, uint64_t lo, uint64_t hi);

bool sketch_applies(const szg_index *ix, int k)
{
    if (!ix->sketch_on || ix->sk_disabled || ix->bits != 32) return false;
    uint64_t n = 0;
    for (const Shard *sh : ix->shards) n += sh->n_rows;
    // the sketch sweep must keep its lists short: 8-bit rows pass four times as fast as float32 rows, and with
    // LDS-resident lists of hundreds (k = 100: 1.35 ms per sweep) the pre-pass is slower than the sweep it replaces
    const int kk = k + ix->sketch_extra;
    return n >= (uint64_t)ix->sketch_min_rows && kk + std::max(ix->slack_min, kk / 2) <= 96;
}

// bring the sketch index up to date with the rows (callers hold ix->sk_mu)
int sketch_sync(szg_index *ix)
{
    if (ix->sk_gen == ix->gen && ix->sketch) return SZG_OK;
    if (!ix->sketch) {
        std::vector<int> devs;
        for (Shard *sh : ix->shards) devs.push_back(sh->device);
        int rc = szg_index_create(&ix->sketch, ix->dim, 8, ix->metric, devs.data(), (int)devs.size());
        if (rc) return rc;
        ix->sk_need_full = true;
        ix->sketch->timing = ix->timing;
        for (const auto &o : ix->opt_log) (void)szg_set_option(ix->sketch, o.first.c_str(), o.second);
    }
    szg_index *sk = ix->sketch;
    const size_t n_sh = ix->shards.size();
    bool full = ix->sk_need_full;
    for (size_t s = 0; s < n_sh && !full; s++) {
        const Shard *a = ix->shards[s], *b = sk->shards[s];
        if (b->n_rows > a->n_rows || (b->n_rows && b->first != a->first)) full = true;
    }
    if (ix->sk_dirty_rows.size() > 4096) full = true;
    const bool euclid = ix->metric != SZG_COSINE;
    // Euclidean collections share ONE scale (the largest |x_i|): rows beyond it force a rebuild
    auto max_abs = [&](bool only_new, double *out) -> int {
        double g = 0.0;
        for (size_t s = 0; s < n_sh; s++) {
            Shard *a = ix->shards[s], *b = sk->shards[s];
            const uint64_t have = only_new ? b->n_rows : 0;
            if (a->n_rows <= have) continue;
            HIPCHK(hipSetDevice(a->device));
            unsigned long long *d_max = nullptr, bits = 0;
            HIPCHK(hipMalloc((void **)&d_max, 16));
            hipError_t e = hipMemset(d_max, 0, 16);
            if (e == hipSuccess)
                e = szg::launch_sketch_build(a->rows, ix->layout, ix->dim, nullptr, sk->layout, have, a->n_rows - have,
                                             nullptr, d_max, nullptr, nullptr, 0, 0.0, 1, nullptr);
            if (e == hipSuccess) e = hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost);
            (void)hipFree(d_max);
            if (e != hipSuccess) return fail(SZG_E_DEVICE, "sketch scale pass", e);
            const uint32_t fb = (uint32_t)bits;
            float f;
            memcpy(&f, &fb, 4);
            g = std::max(g, (double)f);
        }
        *out = g;
        return SZG_OK;
    };
    if (euclid && !full) {
        double g = 0.0;
        int rc = max_abs(true, &g);
        if (rc) return rc;
        if (g > ix->sk_gscale) full = true;
    }
    if (full) {
        std::vector<uint64_t> counts;
        for (Shard *sh : ix->shards) counts.push_back(sh->n_rows);
        int rc = reset_shards(sk, counts);
        if (rc) return rc;
        ix->sk_max_ang = 0.0;
        ix->sk_exc.clear();
        ix->sk_dirty_rows.clear();
        ix->sk_live_dirty = true;
        ix->sk_gscale = 0.0;
        if (euclid) {
            double g = 0.0;
            rc = max_abs(false, &g);
            if (rc) return rc;
            ix->sk_gscale = g > 0.0 ? g : 1.0;
        }
    }
    const uint32_t exc_cap = 4096;
    for (size_t s = 0; s < n_sh; s++) {
        Shard *a = ix->shards[s], *b = sk->shards[s];
        if (a->n_rows == 0) continue;
        HIPCHK(hipSetDevice(a->device));
        if (b->n_rows == 0) b->first = a->first;
        const uint64_t have = full ? 0 : b->n_rows;
        std::vector<uint32_t> list;  // overwritten rows of this shard that already had a sketch
        if (!full)
            for (uint64_t r : ix->sk_dirty_rows)
                if (r >= a->first && r < a->first + have) list.push_back((uint32_t)(r - a->first));
        if (have == a->n_rows && list.empty()) continue;
        int rc = shard_reserve(sk, b, a->n_rows);
        if (rc) return rc;
        unsigned long long *d_ang = nullptr;
        uint32_t *d_exc = nullptr, *d_list = nullptr;
        HIPCHK(hipMalloc((void **)&d_ang, 16));
        HIPCHK(hipMalloc((void **)&d_exc, (exc_cap + 1) * sizeof(uint32_t)));
        HIPCHK(hipMemset(d_ang, 0, 16));
        HIPCHK(hipMemset(d_exc, 0, (exc_cap + 1) * sizeof(uint32_t)));
        hipError_t e = hipSuccess;
        if (a->n_rows > have)
            e = szg::launch_sketch_build(a->rows, ix->layout, ix->dim, b->rows, sk->layout, have, a->n_rows - have, nullptr,
                                         d_ang, d_exc + 1, d_exc, exc_cap, ix->sk_gscale, 0, nullptr);
        if (e == hipSuccess && !list.empty()) {
            e = hipMalloc((void **)&d_list, list.size() * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemcpy(d_list, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e == hipSuccess)
                e = szg::launch_sketch_build(a->rows, ix->layout, ix->dim, b->rows, sk->layout, 0, list.size(), d_list,
                                             d_ang, d_exc + 1, d_exc, exc_cap, ix->sk_gscale, 0, nullptr);
        }
        unsigned long long ang_bits = 0;
        std::vector<uint32_t> exc(exc_cap + 1, 0);
        if (e == hipSuccess) e = hipMemcpy(&ang_bits, d_ang, sizeof(ang_bits), hipMemcpyDeviceToHost);  // (synchronises)
        if (e == hipSuccess) e = hipMemcpy(exc.data(), d_exc, exc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
        (void)hipFree(d_ang);
        (void)hipFree(d_exc);
        if (d_list) (void)hipFree(d_list);
        if (e != hipSuccess) return fail(SZG_E_DEVICE, "sketch build", e);
        double ang;
        memcpy(&ang, &ang_bits, sizeof(ang));
        ix->sk_max_ang = std::max(ix->sk_max_ang, ang);
        if (exc[0] > exc_cap || ix->sk_exc.size() + exc[0] > exc_cap) {
            ix->sk_disabled = true;  // a collection of zero / non-finite rows: nothing to gain
            return SZG_OK;
        }
        for (uint32_t i = 0; i < exc[0]; i++) {
            const uint64_t r = a->first + exc[1 + i];
            if (std::find(ix->sk_exc.begin(), ix->sk_exc.end(), r) == ix->sk_exc.end()) ix->sk_exc.push_back(r);
        }
        if (a->n_rows > b->n_rows) {
            rc = shard_set_live(b, b->n_rows, a->n_rows);
            if (rc) return rc;
            b->n_live += a->n_rows - b->n_rows;
            b->n_rows = a->n_rows;
            ix->sk_live_dirty = true;
        }
    }
    if (ix->sk_live_dirty) {  // tombstones: the sketch shards take the rows' live bits over
        for (size_t s = 0; s < n_sh; s++) {
            Shard *a = ix->shards[s], *b = sk->shards[s];
            if (a->n_rows == 0) continue;
            HIPCHK(hipSetDevice(a->device));
            const uint64_t words = (a->n_rows + 63) / 64;
            for (uint64_t w = 0; w < words; w++) b->live_host[w] = a->live_host[w];
            HIPCHK(hipMemcpy(b->live_bits, b->live_host.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice));
            b->has_dead = a->has_dead;
            b->n_live = a->n_live;
        }
        ix->sk_live_dirty = false;
    }
    std::sort(ix->sk_exc.begin(), ix->sk_exc.end());
    ix->sk_dirty_rows.clear();
    ix->sk_need_full = false;
    ix->sk_gen = ix->gen;
    // the sketch index answers with the caller's tunables where they matter for correctness
    sk->tie_mode = ix->tie_mode;
    return SZG_OK;
}

// float64 distances of per-query candidate lists (index-level rows) in one rerank launch per shard
int sketch_exact_distances(szg_index *ix, const double *queries, int nq, const std::vector<std::vector<uint64_t>> &cand,
                           std::vector<std::vector<double>> *dist)
{
    dist->assign(nq, {});
    size_t most = 0;
    for (int j = 0; j < nq; j++) {
        (*dist)[j].assign(cand[j].size(), 0.0);
        most = std::max(most, cand[j].size());
    }
    if (most == 0) return SZG_OK;
    std::vector<uint64_t> local((size_t)nq * most);
    std::vector<uint32_t> where((size_t)nq * most);
    for (Shard *sh : ix->shards) {
        if (sh->n_rows == 0) continue;
        size_t width = 0;
        for (int j = 0; j < nq; j++) {
            size_t n = 0;
            for (size_t i = 0; i < cand[j].size(); i++) {
                const uint64_t r = cand[j][i];
                if (r >= sh->first && r < sh->first + sh->n_rows) {
                    local[(size_t)j * most + n] = r - sh->first;
                    where[(size_t)j * most + n] = (uint32_t)i;
                    n++;
                }
            }
            width = std::max(width, n);
            for (; n < most; n++) local[(size_t)j * most + n] = szg::kInvalidCand;
        }
        if (width == 0) continue;
        HIPCHK(hipSetDevice(sh->device));
        std::lock_guard<std::mutex> bl(sh->sk_buf_mu);
        const size_t q_bytes = (sizeof(double) * (size_t)nq * ix->dim + 255) & ~(size_t)255;
        const size_t c_bytes = (sizeof(uint64_t) * local.size() + 255) & ~(size_t)255;
        const size_t o_bytes = sizeof(szg::RerankOut) * local.size();
        if (sh->sk_buf_cap < q_bytes + c_bytes + o_bytes) {
            if (sh->sk_buf) (void)hipFree(sh->sk_buf);
            sh->sk_buf = nullptr;
            sh->sk_buf_cap = 0;
            const size_t want = (q_bytes + c_bytes + o_bytes) * 2;
            if (hipMalloc((void **)&sh->sk_buf, want) != hipSuccess) return fail(SZG_E_NOMEM, "hipMalloc(sketch re-rank)");
            sh->sk_buf_cap = want;
        }
        double *d_q = reinterpret_cast<double *>(sh->sk_buf);
        uint64_t *d_c = reinterpret_cast<uint64_t *>(sh->sk_buf + q_bytes);
        szg::RerankOut *d_o = reinterpret_cast<szg::RerankOut *>(sh->sk_buf + q_bytes + c_bytes);
        std::vector<szg::RerankOut> h_o((size_t)nq * most);
        hipError_t e = hipMemcpy(d_q, queries, sizeof(double) * (size_t)nq * ix->dim, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_c, local.data(), sizeof(uint64_t) * local.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, d_q, d_c, nullptr, (uint32_t)most,
                                   nq, d_o, nullptr);
        if (e == hipSuccess) e = hipMemcpy(h_o.data(), d_o, sizeof(szg::RerankOut) * h_o.size(), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(SZG_E_DEVICE, "sketch re-rank", e);
        for (int j = 0; j < nq; j++)
            for (size_t n = 0; n < most; n++) {
                if (local[(size_t)j * most + n] == szg::kInvalidCand) break;
                (*dist)[j][where[(size_t)j * most + n]] = h_o[(size_t)j * most + n].dist;
            }
    }
    return SZG_OK;
}

int search_topk_sketch(szg_index *ix, const double *queries, int n_queries, int k, const uint64_t *allow_bits,
                       uint64_t *out_rows, double *out_dist, int32_t *out_count, const uint64_t *const *allow_ptrs)
{
    int rc;
    {   // (mutations come under the caller's write lock: after the sync, searches run side by side)
        std::lock_guard<std::mutex> lk(ix->sk_mu);
        rc = sketch_sync(ix);
    }
    if (rc) return rc;
    if (ix->sk_disabled) return search_topk_impl(ix, queries, n_queries, k, allow_bits, out_rows, out_dist, out_count, allow_ptrs);
    szg_index *sk = ix->sketch;
    uint64_t total_rows = 0;
    for (Shard *sh : ix->shards) total_rows += sh->n_rows;
    const size_t allow_stride = (total_rows + 63) / 64;
    auto mask_of = [&](int qi) -> const uint64_t * {
        if (allow_ptrs) return allow_ptrs[qi];
        return allow_bits ? allow_bits + (size_t)qi * allow_stride : nullptr;
    };
    auto eligible = [&](const uint64_t *m, uint64_t r) -> bool {
        if (m && !((m[r >> 6] >> (r & 63)) & 1)) return false;
        for (const Shard *sh : ix->shards)
            if (r >= sh->first && r < sh->first + sh->n_rows) {
                const uint64_t l = r - sh->first;
                return (sh->live_host[l >> 6] >> (l & 63)) & 1;
            }
        return false;
    };
    const int kk = k + ix->sketch_extra;
    const double gs = ix->sk_gscale;  // Euclidean: sketch distances are in units of gs (the sketch index sees q / gs)
    const double slack = ix->sk_max_ang * (1.0 + 1e-9) + (gs > 0.0 ? 0.0 : 1e-7);  // (+ the rounding of the computed angles, ~1e-9 near 0)
    std::vector<double> q_scaled;
    std::vector<int> redo;  // queries that go to the float32 path
    const int chunk = 512;
    std::vector<uint64_t> s_rows((size_t)chunk * kk);
    std::vector<double> s_dist((size_t)chunk * kk);
    std::vector<int32_t> s_count(chunk);
    for (int q0 = 0; q0 < n_queries; q0 += chunk) {
        const int nq = std::min(chunk, n_queries - q0);
        const double *q = queries + (size_t)q0 * ix->dim;
        std::vector<const uint64_t *> masks(nq);
        bool any_mask = false;
        for (int j = 0; j < nq; j++) {
            masks[j] = mask_of(q0 + j);
            any_mask |= masks[j] != nullptr;
        }
        const double *q_sk = q;
        if (gs > 0.0) {
            q_scaled.resize((size_t)nq * ix->dim);
            for (size_t i = 0; i < q_scaled.size(); i++) q_scaled[i] = q[i] / gs;
            q_sk = q_scaled.data();
        }
        rc = search_topk_impl(sk, q_sk, nq, kk, nullptr, s_rows.data(), s_dist.data(), s_count.data(),
                              any_mask ? masks.data() : nullptr);
        if (rc) return rc;
        // candidates: the sketch neighbours, the query's first k eligible rows, the rows without a sketch
        std::vector<std::vector<uint64_t>> cand(nq);
        std::vector<std::vector<uint64_t>> firstk(nq);
        for (int j = 0; j < nq; j++) {
            if (j > 0 && !masks[j] && !masks[j - 1]) firstk[j] = firstk[j - 1];
            else first_eligible_rows(ix, masks[j], k, &firstk[j]);
            std::vector<uint64_t> &c = cand[j];
            c.assign(s_rows.begin() + (size_t)j * kk, s_rows.begin() + (size_t)j * kk + s_count[j]);
            c.insert(c.end(), firstk[j].begin(), firstk[j].end());
            for (uint64_t r : ix->sk_exc)
                if (eligible(masks[j], r)) c.push_back(r);
            std::sort(c.begin(), c.end());
            c.erase(std::unique(c.begin(), c.end()), c.end());
        }
        std::vector<std::vector<double>> dist;
        rc = sketch_exact_distances(ix, q, nq, cand, &dist);
        if (rc) return rc;
        for (int j = 0; j < nq; j++) {
            const int qi = q0 + j;
            std::vector<Cand> cs;
            bool nan_first = false;
            for (size_t i = 0; i < cand[j].size(); i++) {
                const double d = dist[j][i];
                if (std::isnan(d)) {
                    // outside the first k rows a NaN never enters the heap; among them it decides everything
                    if (std::binary_search(firstk[j].begin(), firstk[j].end(), cand[j][i])) nan_first = true;
                    continue;
                }
                cs.push_back(Cand{cand[j][i], d, 0.0f, 0.0});
            }
            std::vector<HeapItem> res;
            replay_topk(cs, k, &res);
            bool ok = !nan_first || ix->tie_mode != 0;
            if (ok && s_count[j] == kk) {  // rows exist that were not re-ranked: d(q, row) >= D - A for all of them
                double D = s_dist[(size_t)j * kk + kk - 1];
                if (gs > 0.0) D = D * gs * (1.0 - 1e-9);
                ok = (int)res.size() == k && res.back().priority * (1.0 + 1e-9) < D - slack;
            }
            if (ok && ix->tie_mode == 0) {
                std::vector<double> d(cs.size());
                for (size_t i = 0; i < cs.size(); i++) d[i] = cs[i].dist;
                if (history_dependent(d.data(), d.size(), k)) ok = false;  // the float32 path replays every row
            }
            if (!ok) {
                redo.push_back(qi);
                continue;
            }
            for (int i = 0; i < k; i++) {
                const bool have = i < (int)res.size();
                out_rows[(size_t)qi * k + i] = have ? res[i].row + ix->row_base : UINT64_MAX;
                out_dist[(size_t)qi * k + i] = have ? res[i].priority : 0.0;
            }
            if (out_count) out_count[qi] = (int32_t)res.size();
        }
    }
    {
        std::lock_guard<std::mutex> sl(ix->stats_mu);
        ix->stats.sketch_queries += (uint64_t)n_queries - redo.size();
        ix->stats.sketch_fallbacks += redo.size();
        ix->stats.queries += (uint64_t)n_queries - redo.size();
    }
    if (!redo.empty()) {
        const int m = (int)redo.size();
        std::vector<double> q2((size_t)m * ix->dim);
        std::vector<const uint64_t *> m2(m);
        std::vector<uint64_t> r2((size_t)m * k);
        std::vector<double> d2((size_t)m * k);
        std::vector<int32_t> c2(m);
        bool any = false;
        for (int i = 0; i < m; i++) {
            memcpy(&q2[(size_t)i * ix->dim], queries + (size_t)redo[i] * ix->dim, sizeof(double) * ix->dim);
            m2[i] = mask_of(redo[i]);
            any |= m2[i] != nullptr;
        }
        rc = search_topk_impl(ix, q2.data(), m, k, nullptr, r2.data(), d2.data(), c2.data(), any ? m2.data() : nullptr);
        if (rc) return rc;
        for (int i = 0; i < m; i++) {
            memcpy(out_rows + (size_t)redo[i] * k, &r2[(size_t)i * k], sizeof(uint64_t) * k);
            memcpy(out_dist + (size_t)redo[i] * k, &d2[(size_t)i * k], sizeof(double) * k);
            if (out_count) out_count[redo[i]] = c2[i];
        }
    }
    return SZG_OK;
}

int search_topk_any(szg_index *ix, const double *queries, int n_queries, int k, const uint64_t *allow_bits,
                    uint64_t *out_rows, double *out_dist, int32_t *out_count, const uint64_t *const *allow_ptrs = nullptr)
{
    // (a batch that shares one sweep on the matrix cores is cheaper per query than any pre-pass)
    const bool shared = ix->multi_query && n_queries >= ix->mq_min;
    if (!shared && sketch_applies(ix, k))
        return search_topk_sketch(ix, queries, n_queries, k, allow_bits, out_rows, out_dist, out_count, allow_ptrs);
    return search_topk_impl(ix, queries, n_queries, k, allow_bits, out_rows, out_dist, out_count, allow_ptrs);
}

// the shard's staging buffer, at least `bytes` large (kept up to 64 MiB between calls)
int shard_stage(Shard *sh, size_t bytes, uint8_t **out)
{
    if (sh->stage_cap < bytes) {
        if (sh->stage) (void)hipFree(sh->stage);
        sh->stage = nullptr;
        sh->stage_cap = 0;
        const size_t want = std::max<size_t>(bytes, 4096);
        hipError_t e = hipMalloc((void **)&sh->stage, want);
        if (e != hipSuccess) return fail(SZG_E_NOMEM, "hipMalloc(staging)", e);
        sh->stage_cap = want;
    }
    *out = sh->stage;
    return SZG_OK;
}

int upload_rows(szg_index *ix, Shard *sh, uint64_t dst_row, const uint8_t *rows, uint64_t n)
{
    if (n == 0) return SZG_OK;
    HIPCHK(hipSetDevice(sh->device));
    const uint64_t chunk_rows = std::max<uint64_t>(1, (64ull << 20) / ix->row_bytes);
    const uint64_t cr = std::min(chunk_rows, n);
    uint8_t *stage = nullptr;
    std::lock_guard<std::mutex> stage_lock(sh->stage_mu);
    int rc = shard_stage(sh, cr * ix->row_bytes, &stage);
    if (rc) return rc;
    // copies and the page-in kernel share the null stream: a chunk's copy waits for the previous
    // chunk's kernel, one synchronisation at the end
    hipError_t e = hipSuccess;
    for (uint64_t off = 0; off < n && e == hipSuccess; off += cr) {
        const uint64_t m = std::min(cr, n - off);
        e = hipMemcpy(stage, rows + off * ix->row_bytes, m * ix->row_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = szg::launch_repack(ix->bits, stage, ix->row_bytes, sh->rows, ix->layout, dst_row + off, m, 0,
                                   nullptr);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(SZG_E_DEVICE, "upload_rows", e);
    return SZG_OK;
}

int shard_reserve(szg_index *ix, Shard *sh, uint64_t rows_needed)
{
    HIPCHK(hipSetDevice(sh->device));
    if (rows_needed > 0xFFFFFFF0ull) return fail(SZG_E_UNSUPPORTED, "more than 2^32 rows per shard");
    if (rows_needed > sh->cap_rows) {
        uint64_t cap = std::max<uint64_t>(rows_needed, sh->cap_rows + sh->cap_rows / 2);
        cap = (cap + 63) & ~63ull;
        uint8_t *nr = nullptr;
        hipError_t e = hipMalloc((void **)&nr, szg::layout_bytes(ix->layout, cap) + 64);
        if (e != hipSuccess) return fail(SZG_E_NOMEM, "hipMalloc(corpus)", e);
        if (sh->rows && sh->n_rows) {
            e = hipMemcpy(nr, sh->rows, szg::layout_bytes(ix->layout, sh->n_rows), hipMemcpyDeviceToDevice);
            if (e != hipSuccess) {
                (void)hipFree(nr);
                return fail(SZG_E_DEVICE, "hipMemcpy(corpus)", e);
            }
        }
        if (sh->rows) (void)hipFree(sh->rows);
        sh->rows = nr;
        sh->cap_rows = cap;
    }
    const uint64_t words = (sh->cap_rows + 63) / 64;
    if (words > sh->bits_cap) {
        uint64_t *nb = nullptr;
        hipError_t e = hipMalloc((void **)&nb, words * sizeof(uint64_t));
        if (e != hipSuccess) return fail(SZG_E_NOMEM, "hipMalloc(live bits)", e);
        e = hipMemset(nb, 0, words * sizeof(uint64_t));
        if (e == hipSuccess && sh->live_bits && sh->bits_cap)
            e = hipMemcpy(nb, sh->live_bits, sh->bits_cap * sizeof(uint64_t), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(nb);
            return fail(SZG_E_DEVICE, "hipMemcpy(live bits)", e);
        }
        if (sh->live_bits) (void)hipFree(sh->live_bits);
        sh->live_bits = nb;
        sh->bits_cap = words;
    }
    if (sh->live_host.size() < words) sh->live_host.resize(words, 0);
    return SZG_OK;
}

// set live bits for rows [lo, hi) of a shard: the host copy is the master, the touched words
// follow it to the device (no read-back)
int shard_set_live(Shard *sh, uint64_t lo, uint64_t hi)
{
    if (hi <= lo) return SZG_OK;
    HIPCHK(hipSetDevice(sh->device));
    const uint64_t w0 = lo / 64, w1 = (hi - 1) / 64;
    if (sh->live_host.size() <= w1) return fail(SZG_E_RANGE, "live bitmap smaller than the shard");
    for (uint64_t r = lo; r < hi;) {
        const uint64_t w = r / 64;
        const uint64_t end = std::min(hi, (w + 1) * 64);
        const uint64_t nb = end - r;
        const uint64_t mask = (nb == 64 ? ~0ull : ((1ull << nb) - 1ull)) << (r % 64);
        sh->live_host[w] |= mask;
        r = end;
    }
    HIPCHK(hipMemcpy(sh->live_bits + w0, sh->live_host.data() + w0, (w1 - w0 + 1) * sizeof(uint64_t),
                     hipMemcpyHostToDevice));
    return SZG_OK;
}

// rows of the index are split over shards in contiguous ranges whose boundaries
// are multiples of 64 (so filter words slice cleanly)
void split_rows(const szg_index *ix, uint64_t n_rows, std::vector<uint64_t> *counts)
{
    const size_t g = ix->shards.size();
    counts->assign(g, 0);
    uint64_t per = (n_rows + g - 1) / g;
    per = (per + 63) & ~63ull;
    uint64_t left = n_rows;
    for (size_t s = 0; s < g; s++) {
        const uint64_t m = std::min(per, left);
        (*counts)[s] = m;
        left -= m;
    }
}

Shard *shard_of(szg_index *ix, uint64_t row, uint64_t *local)
{
    for (Shard *s : ix->shards) {
        if (row >= s->first && row < s->first + s->n_rows) {
            *local = row - s->first;
            return s;
        }
    }
    return nullptr;
}

int reset_shards(szg_index *ix, const std::vector<uint64_t> &counts)
{
    uint64_t first = 0;
    for (size_t s = 0; s < ix->shards.size(); s++) {
        Shard *sh = ix->shards[s];
        HIPCHK(hipSetDevice(sh->device));
        HIPCHK(hipDeviceSynchronize());
        sh->first = first;
        sh->n_rows = 0;
        sh->n_live = 0;
        sh->has_dead = false;
        int rc = shard_reserve(ix, sh, counts[s]);
        if (rc) return rc;
        HIPCHK(szg::launch_fill_bits(sh->live_bits, counts[s], sh->bits_cap, nullptr));
        HIPCHK(hipDeviceSynchronize());
        std::fill(sh->live_host.begin(), sh->live_host.end(), 0ull);
        for (uint64_t w = 0; w * 64 < counts[s]; w++)
            sh->live_host[w] = counts[s] - w * 64 >= 64 ? ~0ull : ((1ull << (counts[s] - w * 64)) - 1ull);
        first += counts[s];
    }
    return SZG_OK;
}

// consider()'s top-k branch over the union of the lists in visit order; get(l, q, i, &row, &dist)
template <typename Count, typename Get>
int merge_lists(int k, int n_lists, int list_len, int n_queries, Count count_of, Get get, uint64_t *out_rows,
                double *out_dist, int32_t *out_count, uint8_t *out_history_dependent)
{
    std::vector<Cand> cands;
    std::vector<HeapItem> res;
    std::vector<double> d;
    for (int q = 0; q < n_queries; q++) {
        cands.clear();
        for (int l = 0; l < n_lists; l++) {
            const int n = std::min<int>(std::max<int>(count_of(l, q), 0), list_len);
            for (int i = 0; i < n; i++) {
                Cand c{0, 0.0, 0.0f};
                get(l, q, i, &c.row, &c.dist);
                cands.push_back(c);
            }
        }
        replay_topk(cands, k, &res);
        if (out_history_dependent) {
            d.resize(cands.size());
            for (size_t i = 0; i < cands.size(); i++) d[i] = cands[i].dist;
            out_history_dependent[q] = history_dependent(d.data(), d.size(), k) ? 1 : 0;
        }
        for (int i = 0; i < k; i++) {
            const bool have = i < (int)res.size();
            out_rows[(size_t)q * k + i] = have ? res[i].row : UINT64_MAX;
            out_dist[(size_t)q * k + i] = have ? res[i].priority : 0.0;
        }
        if (out_count) out_count[q] = (int32_t)res.size();
    }
    return SZG_OK;
}

}  // namespace

// ============================================================== C ABI ==========

extern "C" {

int szg_abi_version(void) { return SZG_ABI_VERSION; }

const char *szg_last_error(void) { return g_last_error.c_str(); }

const char *szg_strerror(int code)
{
    switch (code) {
    case SZG_OK: return "ok";
    case SZG_E_INVALID: return "invalid argument";
    case SZG_E_NOMEM: return "out of memory";
    case SZG_E_DEVICE: return "HIP runtime error";
    case SZG_E_TRUNCATED: return "result truncated: more hits than capacity";
    case SZG_E_NODEVICE: return "no usable gfx950 device";
    case SZG_E_RANGE: return "row index out of range";
    case SZG_E_UNSUPPORTED: return "outside this build's limits";
    default: return "unknown error";
    }
}

int64_t szg_row_bytes(int quant_bits, int dim)
{
    if (dim <= 0) return -1;
    return row_bytes_of(quant_bits, dim);
}

int szg_index_create(szg_index **out, int dim, int quant_bits, int metric, const int *devices,
                     int n_devices)
{
    SZG_TRY
    if (!out) return fail(SZG_E_INVALID, "out is null");
    *out = nullptr;
    if (dim <= 0 || dim > (1 << 20)) return fail(SZG_E_INVALID, "dim out of range");
    const int64_t rb = row_bytes_of(quant_bits, dim);
    if (rb < 0) return fail(SZG_E_INVALID, "unsupported quantization (reference panics, collection.go:809)");
    if (metric != SZG_EUCLIDEAN && metric != SZG_COSINE)
        return fail(SZG_E_INVALID, "unsupported distance method (collection.go:282)");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(SZG_E_NODEVICE, "hipGetDeviceCount", e);

    szg_index *ix = new szg_index();
    ix->dim = dim;
    ix->bits = quant_bits;
    ix->metric = metric;
    ix->row_bytes = (uint32_t)rb;
    ix->pitch = (uint32_t)((rb + 15) & ~15ll);
    // 4- and 8-bit rows of whole 64-byte steps live in 16-row tiles (kernels.h, RowLayout): their
    // single-query walk and the shared sweeps then read 1 KiB runs instead of 64-byte segments
    // (+8-12 % on 4-bit rows, +2.5 % on 8-bit rows; float rows measured -1..0 % and stay linear:
    // scripts/dev_tiles.sh, dev_tiles_all.sh).  SZG_TILES_ALL / SZG_NO_TILES override for A/B runs.
    const bool tiled = (quant_bits <= 8 || getenv("SZG_TILES_ALL") != nullptr) && ix->pitch % 64 == 0 &&
                       getenv("SZG_NO_TILES") == nullptr;
    ix->layout = szg::RowLayout{ix->pitch, tiled ? 1u : 0u, tiled ? ix->pitch / 64u : 0u};
    ix->map = choose_map((int)(ix->pitch / 16), tiled);
    ix->qsw_bytes = szg::query_lds_bytes(quant_bits, ix->map.r16);
    if (quant_bits == 8 || quant_bits == 4) {
        const double M = (double)((1u << quant_bits) - 1u);
        const double slots = (double)ix->map.r16 * (128 / quant_bits);  // elements incl. padding
        ix->norm_bias = slots - (slots - dim) * M * M;  // each padding slot decodes to n = -maxInt
    }
    if (ix->qsw_bytes > 48u * 1024u) {
        delete ix;
        return fail(SZG_E_UNSUPPORTED, "dimension too large for the LDS-resident query");
    }
    std::vector<int> devs;
    if (devices && n_devices > 0) {
        devs.assign(devices, devices + n_devices);
    } else {
        int cur = 0;
        (void)hipGetDevice(&cur);
        devs.push_back(cur);
    }
    for (int d : devs) {
        if (d < 0 || d >= count) {
            szg_index_destroy(ix);
            return fail(SZG_E_INVALID, "device ordinal out of range");
        }
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, d);
        if (e != hipSuccess) {
            szg_index_destroy(ix);
            return fail(SZG_E_NODEVICE, "hipGetDeviceProperties", e);
        }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            szg_index_destroy(ix);
            return fail(SZG_E_NODEVICE, "device is not gfx950 (kernels are built for MI355X only)");
        }
        Shard *sh = new Shard();
        sh->device = d;
        sh->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ix->shards.push_back(sh);
    }
    for (Shard *sh : ix->shards) {
        if (hipSetDevice(sh->device) != hipSuccess ||
            hipStreamCreateWithFlags(&sh->scan_stream, hipStreamNonBlocking) != hipSuccess) {
            szg_index_destroy(ix);
            return fail(SZG_E_DEVICE, "hipStreamCreate(scan stream)");
        }
        if (hipMalloc((void **)&sh->zero16, 64) != hipSuccess || hipMemset(sh->zero16, 0, 64) != hipSuccess) {
            szg_index_destroy(ix);
            return fail(SZG_E_NOMEM, "hipMalloc(zero16)");
        }
        for (int i = 0; i < ix->n_ctx; i++) {
            Ctx *c = nullptr;
            int rc = ctx_alloc(ix, sh, &c);
            if (rc) {
                ctx_free(c);
                szg_index_destroy(ix);
                return rc;
            }
            sh->all_ctx.push_back(c);
            (i < ix->n_ctx_active ? sh->free_ctx : sh->parked_ctx).push_back(c);
        }
    }
    *out = ix;
    return SZG_OK;
    SZG_CATCH
}

void szg_index_destroy(szg_index *ix)
{
    if (!ix) return;
    if (ix->sketch) {
        szg_index_destroy(ix->sketch);
        ix->sketch = nullptr;
    }
    if (g_sites.on) {
        for (int i = 0; i < SiteTimers::N; i++)
            if (g_sites.n[i])
                fprintf(stderr, "[szg sites] %-14s %10.1f us / %8llu calls = %7.2f us\n", g_sites.name[i], g_sites.us[i],
                        (unsigned long long)g_sites.n[i], g_sites.us[i] / (double)g_sites.n[i]);
        g_sites = SiteTimers{};
    }
    for (Shard *sh : ix->shards) {
        (void)hipSetDevice(sh->device);
        (void)hipDeviceSynchronize();
        for (Ctx *c : sh->all_ctx) ctx_free(c);
        if (sh->scan_stream) (void)hipStreamDestroy(sh->scan_stream);
        (void)hipFree(sh->zero16);
        (void)hipFree(sh->stage);
        (void)hipFree(sh->sk_buf);
        (void)hipFree(sh->rows);
        (void)hipFree(sh->live_bits);
        delete sh;
    }
    delete ix;
}

uint64_t szg_index_rows(const szg_index *ix)
{
    uint64_t n = 0;
    if (ix) for (const Shard *s : ix->shards) n += s->n_rows;
    return n;
}

uint64_t szg_index_live_rows(const szg_index *ix)
{
    uint64_t n = 0;
    if (ix) for (const Shard *s : ix->shards) n += s->n_live;
    return n;
}

int szg_index_load(szg_index *ix, const uint8_t *rows, uint64_t n_rows)
{
    SZG_TRY
    if (ix) { ix->gen++; ix->sk_need_full = true; }
    if (!ix || (!rows && n_rows)) return fail(SZG_E_INVALID, "null argument");
    std::vector<uint64_t> counts;
    split_rows(ix, n_rows, &counts);
    int rc = reset_shards(ix, counts);
    if (rc) return rc;
    for (size_t s = 0; s < ix->shards.size(); s++) {
        Shard *sh = ix->shards[s];
        rc = upload_rows(ix, sh, 0, rows + sh->first * ix->row_bytes, counts[s]);
        if (rc) return rc;
        sh->n_rows = counts[s];
        sh->n_live = counts[s];
    }
    return SZG_OK;
    SZG_CATCH
}

int szg_index_synth(szg_index *ix, uint64_t n_rows, uint64_t seed, uint64_t first_row)
{
    SZG_TRY
    if (ix) { ix->gen++; ix->sk_need_full = true; }
    if (!ix) return fail(SZG_E_INVALID, "null argument");
    std::vector<uint64_t> counts;
    split_rows(ix, n_rows, &counts);
    int rc = reset_shards(ix, counts);
    if (rc) return rc;
    for (size_t s = 0; s < ix->shards.size(); s++) {
        Shard *sh = ix->shards[s];
        HIPCHK(hipSetDevice(sh->device));
        HIPCHK(szg::launch_synth(ix->bits, sh->rows, ix->layout, 0, ix->dim, counts[s], seed,
                                 first_row + sh->first, nullptr, nullptr));
        HIPCHK(hipDeviceSynchronize());
        sh->n_rows = counts[s];
        sh->n_live = counts[s];
    }
    return SZG_OK;
    SZG_CATCH
}

// The shard new rows go to.  Ranges stay contiguous in row order, so only the last shard
// that holds rows can grow -- or the next, still empty one can start, which it does only at
// a 64-row boundary (every shard's first row must be a multiple of 64: filter and tombstone
// bitmaps are split between shards by whole words) and once its predecessor holds 4M rows.
Shard *append_target(szg_index *ix)
{
    size_t idx = 0;
    for (size_t s = 0; s < ix->shards.size(); s++)
        if (ix->shards[s]->n_rows) idx = s;
    Shard *t = ix->shards[idx];
    if (idx + 1 < ix->shards.size() && t->n_rows >= (4ull << 20) && (t->first + t->n_rows) % 64 == 0) {
        Shard *nx = ix->shards[idx + 1];
        nx->first = t->first + t->n_rows;
        return nx;
    }
    if (t->n_rows == 0) t->first = 0;
    return t;
}

int szg_index_append(szg_index *ix, const uint8_t *rows, uint64_t n_rows)
{
    SZG_TRY
    if (ix) ix->gen++;
    if (!ix || (!rows && n_rows)) return fail(SZG_E_INVALID, "null argument");
    if (n_rows == 0) return SZG_OK;
    Shard *sh = append_target(ix);
    HIPCHK(hipSetDevice(sh->device));
    HIPCHK(hipDeviceSynchronize());
    int rc = shard_reserve(ix, sh, sh->n_rows + n_rows);
    if (rc) return rc;
    rc = upload_rows(ix, sh, sh->n_rows, rows, n_rows);
    if (rc) return rc;
    rc = shard_set_live(sh, sh->n_rows, sh->n_rows + n_rows);
    if (rc) return rc;
    sh->n_rows += n_rows;
    sh->n_live += n_rows;
    return SZG_OK;
    SZG_CATCH
}

// AddDocument for a block of float64 vectors: quantize + pack on the device
int szg_index_append_f64(szg_index *ix, const double *vectors, uint64_t n_rows)
{
    SZG_TRY
    if (ix) ix->gen++;
    if (!ix || (!vectors && n_rows)) return fail(SZG_E_INVALID, "null argument");
    if (n_rows == 0) return SZG_OK;
    Shard *sh = append_target(ix);
    HIPCHK(hipSetDevice(sh->device));
    HIPCHK(hipDeviceSynchronize());
    int rc = shard_reserve(ix, sh, sh->n_rows + n_rows);
    if (rc) return rc;
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)ix->dim * 8));
    uint8_t *stage8 = nullptr;
    std::unique_lock<std::mutex> stage_lock(sh->stage_mu);
    rc = shard_stage(sh, std::min(chunk, n_rows) * (uint64_t)ix->dim * 8, &stage8);
    if (rc) return rc;
    double *stage = reinterpret_cast<double *>(stage8);
    hipError_t e = hipSuccess;
    for (uint64_t off = 0; off < n_rows && e == hipSuccess; off += chunk) {
        const uint64_t m = std::min(chunk, n_rows - off);
        e = hipMemcpy(stage, vectors + off * (uint64_t)ix->dim, m * (uint64_t)ix->dim * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = szg::launch_synth(ix->bits, sh->rows, ix->layout, sh->n_rows + off, ix->dim, m, 0, 0, stage,
                                  nullptr);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    stage_lock.unlock();
    if (e != hipSuccess) return fail(SZG_E_DEVICE, "append_f64", e);
    rc = shard_set_live(sh, sh->n_rows, sh->n_rows + n_rows);
    if (rc) return rc;
    sh->n_rows += n_rows;
    sh->n_live += n_rows;
    return SZG_OK;
    SZG_CATCH
}

// AddDocument on an existing id from a float64 vector: re-encode one row in place
int szg_index_overwrite_f64(szg_index *ix, uint64_t row, const double *vector)
{
    SZG_TRY
    if (ix) { ix->gen++; ix->sk_dirty_rows.push_back(row); }
    if (!ix || !vector) return fail(SZG_E_INVALID, "null argument");
    uint64_t local;
    Shard *sh = shard_of(ix, row, &local);
    if (!sh) return fail(SZG_E_RANGE, "row out of range");
    HIPCHK(hipSetDevice(sh->device));
    HIPCHK(hipDeviceSynchronize());
    uint8_t *stage8 = nullptr;
    std::lock_guard<std::mutex> stage_lock(sh->stage_mu);
    int rc = shard_stage(sh, (size_t)ix->dim * 8, &stage8);
    if (rc) return rc;
    double *stage = reinterpret_cast<double *>(stage8);
    hipError_t e = hipMemcpy(stage, vector, (size_t)ix->dim * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = szg::launch_synth(ix->bits, sh->rows, ix->layout, local, ix->dim, 1, 0, 0, stage, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(SZG_E_DEVICE, "overwrite_f64", e);
    return SZG_OK;
    SZG_CATCH
}

// The reference's float64 distance from one query to each listed row
int szg_distances(szg_index *ix, const double *query, const uint64_t *rows, uint64_t n, double *out_dist)
{
    SZG_TRY
    if (!ix || !query || (!rows && n) || (!out_dist && n)) return fail(SZG_E_INVALID, "null argument");
    if (n == 0) return SZG_OK;
    const uint64_t total = szg_index_rows(ix);
    for (uint64_t i = 0; i < n; i++)
        if (rows[i] < ix->row_base || rows[i] - ix->row_base >= total) return fail(SZG_E_RANGE, "row out of range");
    for (Shard *sh : ix->shards) {
        if (sh->n_rows == 0) continue;
        std::vector<uint64_t> cands;
        std::vector<uint64_t> where;
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t r = rows[i] - ix->row_base;
            if (r >= sh->first && r < sh->first + sh->n_rows) {
                cands.push_back(r - sh->first);  // key bits 0, row in the low word
                where.push_back(i);
            }
        }
        if (cands.empty()) continue;
        Ctx *c = ctx_acquire(sh);
        CtxGuard guard{sh, c};
        int rc = SZG_OK;
        auto body = [&]() -> int {
            HIPCHK(hipSetDevice(sh->device));
            memcpy(c->h_q64, query, sizeof(double) * ix->dim);
            HIPCHK(hipMemcpyAsync(c->d_q64, c->h_q64, sizeof(double) * ix->dim, hipMemcpyHostToDevice,
                                  c->stream));
            int r2 = ensure_dev(&c->d_collect, &c->collect_cap, cands.size());
            if (r2) return r2;
            r2 = ensure_dev(&c->d_out, &c->d_out_cap, cands.size());
            if (r2) return r2;
            r2 = ensure_host(&c->h_out, &c->h_out_cap, cands.size());
            if (r2) return r2;
            HIPCHK(hipMemcpyAsync(c->d_collect, cands.data(), cands.size() * sizeof(uint64_t),
                                  hipMemcpyHostToDevice, c->stream));
            HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64,
                                      c->d_collect, nullptr, (uint32_t)cands.size(), 1, c->d_out, c->stream));
            HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * cands.size(),
                                  hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (size_t i = 0; i < cands.size(); i++) out_dist[where[i]] = c->h_out[i].dist;
            return SZG_OK;
        };
        rc = body();
        if (rc) return rc;
    }
    return SZG_OK;
    SZG_CATCH
}

// decodeVector + dequantize on the host (collection.go:768-794, quantization.go:25-36);
// integer -> float64 conversions, one correctly rounded division, exact *2 and -1
static void decode_row_host(const uint8_t *data, int dim, int q, double *out)
{
    for (int i = 0; i < dim; i++) {
        uint64_t v = 0;
        switch (q) {
        case 4: v = (i % 2 == 0) ? (uint64_t)(data[i / 2] >> 4) : (uint64_t)(data[i / 2] & 0x0F); break;
        case 8: v = data[i]; break;
        case 16: v = ((uint64_t)data[i * 2] << 8) | data[i * 2 + 1]; break;
        case 32: for (int b = 0; b < 4; b++) v = (v << 8) | data[i * 4 + b]; break;
        default: for (int b = 0; b < 8; b++) v = (v << 8) | data[i * 8 + b]; break;
        }
        if (q == 32) {
            const uint32_t u = (uint32_t)v;
            float f;
            memcpy(&f, &u, 4);
            out[i] = (double)f;
        } else if (q == 64) {
            memcpy(&out[i], &v, 8);
        } else {
            const double maxInt = (double)((1ull << q) - 1);
            const double t = (double)v / maxInt;
            out[i] = t * 2 - 1;
        }
    }
}

int szg_pair_distances(szg_index *ix, const uint64_t *rows_a, const uint64_t *rows_b, uint64_t n_pairs,
                       double *out_dist)
{
    SZG_TRY
    if (!ix || ((!rows_a || !rows_b || !out_dist) && n_pairs)) return fail(SZG_E_INVALID, "null argument");
    if (n_pairs == 0) return SZG_OK;
    const uint64_t total = szg_index_rows(ix);
    for (uint64_t i = 0; i < n_pairs; i++)
        if (rows_a[i] < ix->row_base || rows_a[i] - ix->row_base >= total || rows_b[i] < ix->row_base ||
            rows_b[i] - ix->row_base >= total)
            return fail(SZG_E_RANGE, "row out of range");
    std::vector<uint8_t> done(n_pairs, 0);
    // pairs whose two rows live in one shard: both decoded and compared on the device, one launch per shard
    for (Shard *sh : ix->shards) {
        if (sh->n_rows == 0) continue;
        std::vector<uint32_t> left;
        std::vector<uint64_t> right, where;
        for (uint64_t i = 0; i < n_pairs; i++) {
            const uint64_t a = rows_a[i] - ix->row_base, b = rows_b[i] - ix->row_base;
            if (a >= sh->first && a < sh->first + sh->n_rows && b >= sh->first && b < sh->first + sh->n_rows) {
                left.push_back((uint32_t)(a - sh->first));
                right.push_back(b - sh->first);
                where.push_back(i);
            }
        }
        if (where.empty()) continue;
        Ctx *c = ctx_acquire(sh);
        CtxGuard guard{sh, c};
        HIPCHK(hipSetDevice(sh->device));
        const size_t n = where.size();
        int rc = ensure_dev(&c->d_collect, &c->collect_cap, n + (n + 1) / 2);  // right rows (u64) + left rows (u32)
        if (rc) return rc;
        rc = ensure_dev(&c->d_out, &c->d_out_cap, n);
        if (rc) return rc;
        rc = ensure_host(&c->h_out, &c->h_out_cap, n);
        if (rc) return rc;
        uint32_t *d_left = reinterpret_cast<uint32_t *>(c->d_collect + n);
        HIPCHK(hipMemcpyAsync(c->d_collect, right.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_left, left.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(szg::launch_rerank_pairs(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, d_left, c->d_collect,
                                        (uint32_t)n, c->d_out, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // also keeps `left` / `right` alive until the copies are done
        for (size_t i = 0; i < n; i++) {
            out_dist[where[i]] = c->h_out[i].dist;
            done[where[i]] = 1;
        }
    }
    // pairs that straddle two shards (devices): the left row is read back, decoded as the reference
    // does and sent as the query of a szg_distances call on the right row's shard
    std::vector<uint8_t> bytes((size_t)szg_row_bytes(ix->bits, ix->dim));
    std::vector<double> vec(ix->dim);
    for (uint64_t i = 0; i < n_pairs; i++) {
        if (done[i]) continue;
        int rc = szg_index_read_rows(ix, rows_a[i] - ix->row_base, 1, bytes.data());
        if (rc) return rc;
        decode_row_host(bytes.data(), ix->dim, ix->bits, vec.data());
        rc = szg_distances(ix, vec.data(), &rows_b[i], 1, &out_dist[i]);
        if (rc) return rc;
    }
    return SZG_OK;
    SZG_CATCH
}

int szg_index_overwrite(szg_index *ix, uint64_t row, const uint8_t *row_bytes)
{
    SZG_TRY
    if (ix) { ix->gen++; ix->sk_dirty_rows.push_back(row); }
    if (!ix || !row_bytes) return fail(SZG_E_INVALID, "null argument");
    uint64_t local;
    Shard *sh = shard_of(ix, row, &local);
    if (!sh) return fail(SZG_E_RANGE, "row out of range");
    HIPCHK(hipSetDevice(sh->device));
    HIPCHK(hipDeviceSynchronize());
    return upload_rows(ix, sh, local, row_bytes, 1);
    SZG_CATCH
}

int szg_index_tombstone(szg_index *ix, uint64_t row)
{
    if (!ix) return fail(SZG_E_INVALID, "null argument");
    ix->gen++;
    ix->sk_live_dirty = true;
    uint64_t local;
    Shard *sh = shard_of(ix, row, &local);
    if (!sh) return fail(SZG_E_RANGE, "row out of range");
    HIPCHK(hipSetDevice(sh->device));
    const uint64_t bit = 1ull << (local % 64);
    uint64_t &w = sh->live_host[local / 64];
    if (w & bit) {
        // searches in flight on this device finish first (callers hold the write lock; this
        // also covers a search that failed half-way)
        HIPCHK(hipDeviceSynchronize());
        w &= ~bit;
        HIPCHK(hipMemcpy(sh->live_bits + local / 64, &w, sizeof(w), hipMemcpyHostToDevice));
        sh->n_live--;
        sh->has_dead = true;
    }
    return SZG_OK;
}

int szg_index_read_rows(szg_index *ix, uint64_t first_row, uint64_t n_rows, uint8_t *out)
{
    SZG_TRY
    if (!ix || (!out && n_rows)) return fail(SZG_E_INVALID, "null argument");
    if (first_row + n_rows > szg_index_rows(ix)) return fail(SZG_E_RANGE, "row range out of bounds");
    for (Shard *sh : ix->shards) {
        const uint64_t lo = std::max(first_row, sh->first);
        const uint64_t hi = std::min(first_row + n_rows, sh->first + sh->n_rows);
        if (hi <= lo) continue;
        HIPCHK(hipSetDevice(sh->device));
        const uint64_t m = hi - lo;
        const uint64_t cr = std::min<uint64_t>(m, std::max<uint64_t>(1, (64ull << 20) / ix->row_bytes));
        uint8_t *stage = nullptr;
        std::lock_guard<std::mutex> stage_lock(sh->stage_mu);
        int rc = shard_stage(sh, cr * ix->row_bytes, &stage);
        if (rc) return rc;
        hipError_t e = hipSuccess;
        for (uint64_t off = 0; off < m && e == hipSuccess; off += cr) {
            const uint64_t mm = std::min(cr, m - off);
            e = szg::launch_repack(ix->bits, stage, ix->row_bytes, sh->rows, ix->layout, lo - sh->first + off, mm, 1,
                                   nullptr);
            if (e == hipSuccess)
                e = hipMemcpy(out + (lo - first_row + off) * ix->row_bytes, stage, mm * ix->row_bytes,
                              hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) return fail(SZG_E_DEVICE, "read_rows", e);
    }
    return SZG_OK;
    SZG_CATCH
}

int szg_index_set_row_base(szg_index *ix, uint64_t base)
{
    if (!ix) return fail(SZG_E_INVALID, "null argument");
    ix->row_base = base;
    return SZG_OK;
}

int szg_search_topk(szg_index *ix, const double *queries, int n_queries, int k,
                    const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist,
                    int32_t *out_count)
{
    if (!ix || !queries || !out_rows || !out_dist) return fail(SZG_E_INVALID, "null argument");
    if (n_queries < 0 || k <= 0) return fail(SZG_E_INVALID, "k must be > 0 (K==0 is listing mode, collection.go:633)");
    if (n_queries == 0) return SZG_OK;
    if (szg_index_rows(ix) == 0) {  // empty collection -> no results (collection_test.go:294-309)
        for (size_t i = 0; i < (size_t)n_queries * k; i++) {
            out_rows[i] = UINT64_MAX;
            out_dist[i] = 0.0;
        }
        if (out_count) for (int i = 0; i < n_queries; i++) out_count[i] = 0;
        return SZG_OK;
    }
    if (!(ix->coalesce && n_queries == 1 && ix->multi_query)) {
        SZG_TRY
        return search_topk_any(ix, queries, n_queries, k, allow_bits, out_rows, out_dist, out_count);
        SZG_CATCH
    }

    // Search holds only RLock in the reference (collection.go:570), so many goroutines call in
    // at once, each with ONE query.  Whoever finds no batch in flight becomes the leader: it
    // answers everything that is waiting with the same k as one batch (one shared sweep instead
    // of one sweep per caller), then hands the lead to a waiter if its own answer has arrived.
    // A lone caller is its own batch of one and pays nothing for this.
    PendingSearch me;
    me.query = queries;
    me.allow = allow_bits;
    me.k = k;
    me.out_rows = out_rows;
    me.out_dist = out_dist;
    me.out_count = out_count;
    std::vector<PendingSearch *> batch;
    std::unique_lock<std::mutex> lk(ix->comb_mu);
    try {
        batch.reserve(kMaxBatch);  // nothing below that touches the combiner's state may throw
        ix->comb_waiting.push_back(&me);
    } catch (...) {
        return fail(SZG_E_NOMEM, "out of memory (host)");
    }
    if (ix->comb_leader) {
        me.cv.wait(lk, [&] { return me.done || me.lead; });
        if (me.done) return me.rc;
    }
    ix->comb_leader = true;
    std::vector<double> q;
    std::vector<uint64_t> rows;
    std::vector<double> dist;
    std::vector<int32_t> count;
    std::vector<const uint64_t *> masks;
    while (!me.done) {
        batch.clear();
        const int kk = ix->comb_waiting.front()->k;  // never empty here: `me` is in it until done
        for (auto it = ix->comb_waiting.begin(); it != ix->comb_waiting.end() && batch.size() < (size_t)kMaxBatch;) {
            if ((*it)->k == kk) {
                batch.push_back(*it);  // within the reserved capacity
                it = ix->comb_waiting.erase(it);
            } else {
                ++it;
            }
        }
        lk.unlock();
        const int nq = (int)batch.size();
        int rc;
        try {
            if (nq == 1) {
                PendingSearch *p = batch[0];
                rc = search_topk_any(ix, p->query, 1, kk, p->allow, p->out_rows, p->out_dist, p->out_count);
            } else {
                q.resize((size_t)nq * ix->dim);
                rows.resize((size_t)nq * kk);
                dist.resize((size_t)nq * kk);
                count.resize(nq);
                masks.resize(nq);
                bool any = false;
                for (int i = 0; i < nq; i++) {
                    memcpy(&q[(size_t)i * ix->dim], batch[i]->query, sizeof(double) * ix->dim);
                    masks[i] = batch[i]->allow;  // each caller's own filter, if it has one
                    any |= masks[i] != nullptr;
                }
                rc = search_topk_any(ix, q.data(), nq, kk, nullptr, rows.data(), dist.data(), count.data(),
                                      any ? masks.data() : nullptr);
                for (int i = 0; i < nq && rc == SZG_OK; i++) {
                    memcpy(batch[i]->out_rows, &rows[(size_t)i * kk], sizeof(uint64_t) * kk);
                    memcpy(batch[i]->out_dist, &dist[(size_t)i * kk], sizeof(double) * kk);
                    if (batch[i]->out_count) *batch[i]->out_count = count[i];
                }
            }
        } catch (const std::bad_alloc &) {
            rc = fail(SZG_E_NOMEM, "out of memory (host)");
        } catch (...) {
            rc = fail(SZG_E_DEVICE, "unexpected exception");
        }
        lk.lock();
        for (PendingSearch *p : batch) {
            p->rc = rc;
            p->done = true;
            if (p != &me) p->cv.notify_one();
        }
    }
    if (!ix->comb_waiting.empty()) {
        ix->comb_waiting.front()->lead = true;  // it stays queued and forms the next batch itself
        ix->comb_waiting.front()->cv.notify_one();
    } else {
        ix->comb_leader = false;
    }
    return me.rc;
}

int szg_search_radius(szg_index *ix, const double *query, double radius,
                      const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist,
                      uint64_t capacity, uint64_t *out_total)
{
    SZG_TRY
    if (!ix || !query || !out_total) return fail(SZG_E_INVALID, "null argument");
    if (!(radius > 0)) return fail(SZG_E_INVALID, "radius must be > 0 (collection.go:598)");
    if (capacity && (!out_rows || !out_dist)) return fail(SZG_E_INVALID, "null output buffer");
    *out_total = 0;
    const uint64_t total_rows = szg_index_rows(ix);
    if (total_rows == 0) return SZG_OK;

    // key threshold that surely contains every row with distance <= radius
    QMeta meta;
    std::vector<uint8_t> tmp(ix->qsw_bytes);
    prep_query(ix, query, tmp.data(), &meta);
    const double m1 = meta.m1;
    float thr_f;
    if (ix->metric == SZG_COSINE) {
        if (radius >= 1.0 || m1 == 0) {
            thr_f = 3.0e38f;  // acos(c)/pi <= 1 always; zero query -> all 1.0
        } else {
            const double t = -std::cos(M_PI * radius) + 2.0 * key_eps(ix, 1.0, meta) + 1e-12;
            thr_f = std::nextafter((float)t, INFINITY);
        }
    } else {
        const double scale = ix->bits <= 16 ? (double)((1u << ix->bits) - 1u) : 1.0;
        const double kk = (radius * scale) * (radius * scale);
        const double t = kk * (1.0 + 1e-12) + 2.0 * key_eps(ix, kk, meta);
        // (an infinite radius with a zero query makes t = inf + 0 * inf = NaN: everything, as for any t beyond the floats)
        thr_f = !(t < 3.0e38) ? 3.0e38f : std::nextafter((float)t, INFINITY);
    }

    std::vector<Cand> cands;
    int rc = SZG_OK;
    for (size_t s = 0; s < ix->shards.size() && rc == SZG_OK; s++) {
        Shard *sh = ix->shards[s];
        if (sh->n_rows == 0) continue;
        Ctx *c = ctx_acquire(sh);
        CtxGuard guard{sh, c};
        memcpy(c->h_qsw, tmp.data(), ix->qsw_bytes);
        c->meta[0] = meta;
        rc = enqueue_queries(ix, sh, c, query, 1, allow_bits ? &allow_bits : nullptr);
        if (rc == SZG_OK) rc = run_collect(ix, sh, c, 0, thr_f, allow_bits != nullptr, &cands);
    }
    if (rc) return rc;
    // consider()'s radius branch (collection.go:598-605) in visit order, then the pop loop
    std::sort(cands.begin(), cands.end(), [](const Cand &x, const Cand &y) { return x.row < y.row; });
    GoHeap h;
    for (const Cand &c : cands)
        if (c.dist <= radius) h.push(HeapItem{c.row, c.dist});
    const uint64_t total = h.a.size();
    *out_total = total;
    for (uint64_t i = total; i-- > 0;) {
        const HeapItem it = h.pop();
        if (i < capacity) {
            out_rows[i] = it.row + ix->row_base;
            out_dist[i] = it.priority;
        }
    }
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->stats.queries++;
    }
    if (total > capacity) return fail(SZG_E_TRUNCATED, "radius search: capacity too small");
    return SZG_OK;
    SZG_CATCH
}

/*
 * Cross-shard result assembly for one-process-per-GPU sharding: every rank
 * answers the query on its row range with szg_search_topk (list_len = k+1
 * results, rows already global via szg_index_set_row_base), the per-rank
 * lists are exchanged (RCCL all-gather) and this replays consider()'s top-k
 * branch over their union in visit order.  Pure host code.
 */
int szg_merge_topk(int k, int n_lists, int list_len, int n_queries, const uint64_t *rows,
                   const double *dist, const int32_t *counts, uint64_t *out_rows, double *out_dist,
                   int32_t *out_count, uint8_t *out_history_dependent)
{
    if (k <= 0 || n_lists <= 0 || list_len <= 0 || n_queries < 0 || !rows || !dist || !counts ||
        !out_rows || !out_dist)
        return fail(SZG_E_INVALID, "bad argument");
    try {
        return merge_lists(
            k, n_lists, list_len, n_queries, [&](int l, int q) { return counts[(size_t)l * n_queries + q]; },
            [&](int l, int q, int i, uint64_t *r, double *d) {
                const size_t at = ((size_t)l * n_queries + q) * list_len + i;
                *r = rows[at];
                *d = dist[at];
            },
            out_rows, out_dist, out_count, out_history_dependent);
    } catch (const std::bad_alloc &) {
        return fail(SZG_E_NOMEM, "out of memory");
    }
}

/* The same merge straight from the exchanged records (no repacking on the caller's side):
 * records[n_lists][n_queries][2*list_len + 1] int64 = list_len rows | list_len float64 bit
 * patterns | count -- exactly what each rank contributes to the all-gather. */
int szg_merge_topk_records(int k, int n_lists, int list_len, int n_queries, const int64_t *records,
                           uint64_t *out_rows, double *out_dist, int32_t *out_count,
                           uint8_t *out_history_dependent)
{
    if (k <= 0 || n_lists <= 0 || list_len <= 0 || n_queries < 0 || !records || !out_rows || !out_dist)
        return fail(SZG_E_INVALID, "bad argument");
    const size_t rec = 2 * (size_t)list_len + 1;
    try {
        return merge_lists(
            k, n_lists, list_len, n_queries,
            [&](int l, int q) { return (int)records[((size_t)l * n_queries + q) * rec + 2 * list_len]; },
            [&](int l, int q, int i, uint64_t *r, double *d) {
                const int64_t *p = records + ((size_t)l * n_queries + q) * rec;
                *r = (uint64_t)p[i];
                memcpy(d, &p[list_len + i], sizeof(double));
            },
            out_rows, out_dist, out_count, out_history_dependent);
    } catch (const std::bad_alloc &) {
        return fail(SZG_E_NOMEM, "out of memory");
    }
}

int szg_set_timing(szg_index *ix, int enabled)
{
    if (!ix) return fail(SZG_E_INVALID, "null argument");
    for (Shard *sh : ix->shards) {
        (void)hipSetDevice(sh->device);
        (void)hipDeviceSynchronize();
    }
    ix->timing = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
    if (ix->sketch) ix->sketch->timing = ix->timing;
    return SZG_OK;
}

int szg_get_stats(szg_index *ix, szg_stats *out)
{
    if (!ix || !out) return fail(SZG_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    *out = ix->stats;
    if (ix->sketch) {  // the sweeps of the sketch pre-pass count as this index's
        std::lock_guard<std::mutex> lk2(ix->sketch->stats_mu);
        const szg_stats &k = ix->sketch->stats;
        out->scan_launches += k.scan_launches;
        out->escalations += k.escalations;
        out->scan_bytes += k.scan_bytes;
        out->scan_ms += k.scan_ms;
        out->total_ms += k.total_ms;
        out->timed_launches += k.timed_launches;
        out->full_replays += k.full_replays;
        out->mq_launches += k.mq_launches;
        out->mq_queries += k.mq_queries;
        out->mq_fallbacks += k.mq_fallbacks;
        out->host_prep_us += k.host_prep_us;
        out->host_finish_us += k.host_finish_us;
        out->host_enqueue_us += k.host_enqueue_us;
    }
    return SZG_OK;
}

int szg_reset_stats(szg_index *ix)
{
    if (!ix) return fail(SZG_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    ix->stats = szg_stats{};
    if (ix->sketch) {
        std::lock_guard<std::mutex> lk2(ix->sketch->stats_mu);
        ix->sketch->stats = szg_stats{};
    }
    return SZG_OK;
}

int szg_set_option(szg_index *ix, const char *name, int64_t value)
{
    SZG_TRY
    if (!ix || !name) return fail(SZG_E_INVALID, "null argument");
    const std::string n(name);
    if (n == "sketch") {
        ix->sketch_on = value != 0;
        return SZG_OK;
    }
    if (n == "sketch_min_rows") {
        if (value < 1) return fail(SZG_E_INVALID, "sketch_min_rows out of range");
        ix->sketch_min_rows = (int)std::min<int64_t>(value, 1 << 30);
        return SZG_OK;
    }
    if (n == "sketch_extra") {
        if (value < 0 || value > 900) return fail(SZG_E_INVALID, "sketch_extra out of range");
        ix->sketch_extra = (int)value;
        return SZG_OK;
    }
    ix->opt_log.emplace_back(n, value);  // the sketch index follows the same tunables
    if (ix->sketch) (void)szg_set_option(ix->sketch, name, value);
    if (n == "slack") {
        if (value < 0 || value > 4096) return fail(SZG_E_INVALID, "slack out of range");
        ix->slack_min = (int)value;
    } else if (n == "blocks_per_cu") {
        if (value < 0 || value > 16) return fail(SZG_E_INVALID, "blocks_per_cu out of range");
        ix->blocks_per_cu = (int)value;
    } else if (n == "block_threads") {
        if (value != 64 && value != 128 && value != 256)
            return fail(SZG_E_INVALID, "block_threads must be 64/128/256");
        ix->block_threads = (int)value;
    } else if (n == "shape_kernels") {
        ix->shape_kernels = value != 0;
    } else if (n == "ring") {
        if (value != 0 && value != 8) return fail(SZG_E_INVALID, "ring must be 0 (auto) or 8 (deep)");
        ix->ring = (int)value;
    } else if (n == "queries_per_launch") {
        if (value < 1 || value > szg::kMaxSweepsPerLaunch)
            return fail(SZG_E_INVALID, "queries_per_launch out of range");
        ix->queries_per_launch = (int)value;
    } else if (n == "query_batch") {
        if (value < 1 || value > kMaxBatch) return fail(SZG_E_INVALID, "query_batch out of range");
        ix->query_batch = (int)value;
    } else if (n == "contexts") {
        if (value < 1 || value > ix->n_ctx) return fail(SZG_E_INVALID, "contexts out of range");
        for (Shard *sh : ix->shards) {   // call while no search is in flight
            std::lock_guard<std::mutex> lk(sh->mu);
            while (!sh->parked_ctx.empty()) {
                sh->free_ctx.push_back(sh->parked_ctx.back());
                sh->parked_ctx.pop_back();
            }
            while ((int64_t)sh->free_ctx.size() > value) {
                sh->parked_ctx.push_back(sh->free_ctx.back());
                sh->free_ctx.pop_back();
            }
        }
    } else if (n == "lanes_per_row") {
        // tuning hook: force the lane-group width L (power of two, L*P >= r16)
        if (ix->layout.tiled) return fail(SZG_E_UNSUPPORTED, "lanes_per_row: tiled rows walk 4 lanes per row");
        const int L = (int)value, r16 = ix->map.r16;
        if (L < 1 || L > 64 || (L & (L - 1))) return fail(SZG_E_INVALID, "lanes_per_row must be a power of two <= 64");
        const int P = (r16 + L - 1) / L;
        ix->map = szg::RowMap{r16, L, P, 64 / L, 1, (L * P == r16) ? 1 : 0};
    } else if (n == "multi_query") {
        ix->multi_query = value != 0;
    } else if (n == "mq_blocks") {
        if (value < 1 || value > 6) return fail(SZG_E_INVALID, "mq_blocks must be 1..6");
        ix->mq_blocks_max = (int)value;
    } else if (n == "mask_dense") {
        ix->mask_dense = value != 0;
    } else if (n == "coalesce") {
        ix->coalesce = value != 0;
    } else if (n == "mq_fused") {
        ix->mq_fused = value != 0;
    } else if (n == "mq_i8") {
        ix->mq_i8 = value != 0;
    } else if (n == "mq_i8_groups") {
        if (value < 1 || value > 2) return fail(SZG_E_INVALID, "mq_i8_groups must be 1 or 2");
        ix->mq_i8_groups = (int)value;
    } else if (n == "mq_bf16") {
        ix->mq_bf16 = value != 0;
    } else if (n == "mq_overlap") {
        ix->mq_overlap = value != 0;
    } else if (n == "mq_bf16_slack") {
        if (value < 0 || value > 4000) return fail(SZG_E_INVALID, "mq_bf16_slack out of range");
        ix->mq_bf16_slack = (int)value;
    } else if (n == "mq_tail_overlap") {
        ix->mq_tail_overlap = value != 0;
    } else if (n == "mq_hits") {
        if (value < 64 || value > 65536) return fail(SZG_E_INVALID, "mq_hits out of range");
        ix->mq_hits = (int)value;
    } else if (n == "mq_min") {
        if (value < 1 || value > 32) return fail(SZG_E_INVALID, "mq_min out of range");
        ix->mq_min = (int)value;
    } else if (n == "serialize_scans") {
        ix->serialize_scans = value != 0;
    } else if (n == "tie_mode") {
        if (value != 0 && value != 1) return fail(SZG_E_INVALID, "tie_mode must be 0 or 1");
        ix->tie_mode = (int)value;
    } else if (n == "force_escalate") {
        ix->force_escalate = value != 0;
    } else {
        return fail(SZG_E_INVALID, "unknown option");
    }
    return SZG_OK;
    SZG_CATCH
}

// test hook: device float64 primitives (0 div, 1 sqrt, 2 go acos, 3 round, 4 f32 narrowing)
int szg_debug_f64_probe(int op, const double *a, const double *b, double *out, uint64_t n)
{
    if (!a || !out) return fail(SZG_E_INVALID, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(SZG_E_NODEVICE, "hipGetDeviceCount");
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void **)&da, n * sizeof(double)));
    HIPCHK(hipMalloc((void **)&db, n * sizeof(double)));
    HIPCHK(hipMalloc((void **)&dout, n * sizeof(double)));
    HIPCHK(hipMemcpy(da, a, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b ? b : a, n * sizeof(double), hipMemcpyHostToDevice));
    hipError_t e = szg::launch_f64_probe(op, da, db, dout, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, dout, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    if (e != hipSuccess) return fail(SZG_E_DEVICE, "f64 probe", e);
    return SZG_OK;
}

}  // extern "C"
