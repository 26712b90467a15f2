// scan_comm.cpp -- one process per GPU: the cross-shard exchange inside the library.
//
// The rows of a collection are sharded over the ranks in contiguous ranges (szg_index_set_row_base makes the rows a
// rank returns global); every rank holds every query.  A sharded search is: the rank's own exact top-(k+1) on its
// range (the whole pipeline of scan_topk.cpp) -> ONE ncclAllGather (RCCL, xGMI inside a node) of one int64 record
// per query [k+1 rows | k+1 float64 bit patterns | count] -> consider()'s top-k branch replayed over the union in
// visit order on every rank (szg_merge_topk_records: collection.go:606-619, :694-697).  A host (Go, C++, Python)
// needs nothing but this library for it: rank 0 creates the communicator id, hands the 128 bytes to the other ranks
// by any means it has, every rank attaches.  Long query lists are pipelined: a worker thread sweeps chunk i+1 while
// the calling thread exchanges and merges chunk i.
//
// A second transport -- a host callback that all-gathers bytes -- serves hosts with their own fabric and the tests
// (gloo on CPU, several ranks on one card, where RCCL refuses to form a communicator).
#include "scan_internal.h"

#include <rccl/rccl.h>

#include <thread>

// The communicator: RCCL (device staging, a stream of its own) or the host's transport (plain host memory, no HIP
// call at all -- usable without a GPU, which is how the record path is tested on CPU).
struct szg_comm {
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    szg_allgather_fn host_fn = nullptr;
    void *host_user = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    // staging of one exchange, reused call after call: cap int64 words per rank
    int64_t *h_mine = nullptr, *h_all = nullptr;  // pinned (RCCL) / malloc (host transport)
    int64_t *d_mine = nullptr, *d_all = nullptr;  // RCCL only
    size_t cap = 0;
    std::mutex mu;  // collectives must be issued in the same order on every rank: one sharded call at a time
    szg_comm_stats stats{};
};

namespace szgi {

namespace {

int nccl_fail(const char *what, ncclResult_t r)
{
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, ncclGetErrorString(r));
    return fail(SZG_E_DEVICE, buf);
}
#define NCCLCHK(expr)                                        \
    do {                                                     \
        const ncclResult_t r__ = (expr);                     \
        if (r__ != ncclSuccess) return nccl_fail(#expr, r__); \
    } while (0)

void comm_free_staging(szg_comm *cm)
{
    if (cm->nccl) {
        (void)hipHostFree(cm->h_mine);
        (void)hipHostFree(cm->h_all);
        (void)hipFree(cm->d_mine);
        (void)hipFree(cm->d_all);
    } else {
        free(cm->h_mine);
        free(cm->h_all);
    }
    cm->h_mine = cm->h_all = cm->d_mine = cm->d_all = nullptr;
    cm->cap = 0;
}

// room for `words` int64 per rank (grown geometrically; szg_comm_reserve sizes it ahead of a timed run)
int comm_reserve(szg_comm *cm, size_t words)
{
    if (words <= cm->cap) return SZG_OK;
    const size_t want = std::max<size_t>(std::max(words, cm->cap * 2), 4096);
    if (cm->nccl) {
        HIPCHK(hipSetDevice(cm->device));
        HIPCHK(hipStreamSynchronize(cm->stream));
        comm_free_staging(cm);
        HIPCHK(hipHostMalloc((void **)&cm->h_mine, want * sizeof(int64_t), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&cm->h_all, want * sizeof(int64_t) * cm->world, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&cm->d_mine, want * sizeof(int64_t)));
        HIPCHK(hipMalloc((void **)&cm->d_all, want * sizeof(int64_t) * cm->world));
    } else {
        comm_free_staging(cm);
        cm->h_mine = (int64_t *)malloc(want * sizeof(int64_t));
        cm->h_all = (int64_t *)malloc(want * sizeof(int64_t) * cm->world);
        if (!cm->h_mine || !cm->h_all) return fail(SZG_E_NOMEM, "exchange staging");
    }
    cm->cap = want;
    return SZG_OK;
}

// all-gather of `words` int64 per rank: h_mine -> h_all [world][words]
int comm_exchange(szg_comm *cm, size_t words)
{
    const double t0 = now_us();
    if (cm->nccl) {
        HIPCHK(hipSetDevice(cm->device));
        HIPCHK(hipMemcpyAsync(cm->d_mine, cm->h_mine, words * sizeof(int64_t), hipMemcpyHostToDevice, cm->stream));
        NCCLCHK(ncclAllGather(cm->d_mine, cm->d_all, words, ncclInt64, cm->nccl, cm->stream));
        HIPCHK(hipMemcpyAsync(cm->h_all, cm->d_all, words * sizeof(int64_t) * cm->world, hipMemcpyDeviceToHost,
                              cm->stream));
        HIPCHK(hipStreamSynchronize(cm->stream));
    } else {
        const int rc = cm->host_fn(cm->host_user, cm->h_mine, cm->h_all, (uint64_t)(words * sizeof(int64_t)));
        if (rc != 0) return fail(SZG_E_DEVICE, "exchange callback failed");
    }
    cm->stats.exchanges++;
    cm->stats.exchange_us += now_us() - t0;
    return SZG_OK;
}

// The rank's top-(k+1) lists of n queries -> all-gather -> consider()'s top-k branch over the union (every rank).
// kk = k + 1 entries per query in rows / dist.  local_rc != 0: this rank's own search failed; it still takes part in
// the collective (the others are waiting in it) and says so with a count of -1.  Caller holds cm->mu.
int comm_merge_topk(szg_comm *cm, int k, int n, const uint64_t *rows, const double *dist, const int32_t *counts,
                    int local_rc, uint64_t *out_rows, double *out_dist, int32_t *out_count, uint8_t *out_hist)
{
    const int kk = k + 1;
    const size_t w = 2 * (size_t)kk + 1;
    int rc = comm_reserve(cm, (size_t)n * w);
    if (rc) return rc;
    const double t0 = now_us();
    // one int64 record per query: kk rows | kk distance bit patterns | count
    for (int j = 0; j < n; j++) {
        int64_t *rec = cm->h_mine + (size_t)j * w;
        memcpy(rec, rows + (size_t)j * kk, sizeof(uint64_t) * kk);
        memcpy(rec + kk, dist + (size_t)j * kk, sizeof(double) * kk);
        rec[2 * kk] = local_rc == SZG_OK ? (int64_t)counts[j] : -1;
    }
    const double t1 = now_us();
    rc = comm_exchange(cm, (size_t)n * w);
    if (rc) return rc;  // the transport itself failed
    const double t2 = now_us();
    for (int g = 0; g < cm->world; g++)
        if (cm->h_all[(size_t)g * n * w + 2 * kk] < 0)
            return local_rc != SZG_OK ? local_rc : fail(SZG_E_DEVICE, "a peer rank's search failed");
    rc = szg_merge_topk_records(k, cm->world, kk, n, cm->h_all, out_rows, out_dist, out_count, out_hist);
    cm->stats.host_us += (t1 - t0) + (now_us() - t2);
    return rc;
}

// The rank's radius hits (CSR: offsets[n + 1], rows GLOBAL) -> all-gather of the counts -> one padded all-gather of
// (row, distance bits) records -> consider()'s radius branch over the union in visit order (collection.go:598-603)
// and the pop loop (:694-697): the single-collection answer, ties included.  Caller holds cm->mu.
int comm_merge_radius(szg_comm *cm, int n, const uint64_t *offsets, const uint64_t *rows, const double *dist,
                      int local_rc, uint64_t *out_rows, double *out_dist, uint64_t capacity, uint64_t *out_offsets)
{
    const int G = cm->world;
    int rc = comm_reserve(cm, (size_t)n);
    if (rc) return rc;
    for (int i = 0; i < n; i++) cm->h_mine[i] = local_rc == SZG_OK ? (int64_t)(offsets[i + 1] - offsets[i]) : -1;
    rc = comm_exchange(cm, (size_t)n);
    if (rc) return rc;
    std::vector<int64_t> counts(cm->h_all, cm->h_all + (size_t)G * n);
    size_t most = 0;
    for (int g = 0; g < G; g++) {
        size_t sum = 0;
        for (int i = 0; i < n; i++) {
            if (counts[(size_t)g * n + i] < 0)
                return local_rc != SZG_OK ? local_rc : fail(SZG_E_DEVICE, "a peer rank's search failed");
            sum += (size_t)counts[(size_t)g * n + i];
        }
        most = std::max(most, sum);
    }
    if (most) {  // every rank's queries back to back, padded to the largest rank
        rc = comm_reserve(cm, 2 * most);
        if (rc) return rc;
        const size_t mine = (size_t)(offsets[n] - offsets[0]);
        for (size_t i = 0; i < mine; i++) {
            cm->h_mine[2 * i] = (int64_t)rows[offsets[0] + i];
            memcpy(&cm->h_mine[2 * i + 1], &dist[offsets[0] + i], sizeof(double));
        }
        rc = comm_exchange(cm, 2 * most);
        if (rc) return rc;
    }
    const double t0 = now_us();
    std::vector<size_t> cursor(G, 0);
    std::vector<Cand> cs;
    std::vector<HeapItem> res;
    uint64_t off = 0;
    for (int i = 0; i < n; i++) {
        cs.clear();
        for (int g = 0; g < G; g++) {
            const int64_t *p = cm->h_all + (size_t)g * 2 * most + cursor[g];
            const size_t m = (size_t)counts[(size_t)g * n + i];
            for (size_t j = 0; j < m; j++) {
                Cand c{(uint64_t)p[2 * j], 0.0, 0.0f, 0.0};
                memcpy(&c.dist, &p[2 * j + 1], sizeof(double));
                cs.push_back(c);
            }
            cursor[g] += 2 * m;
        }
        std::sort(cs.begin(), cs.end(), [](const Cand &x, const Cand &y) { return x.row < y.row; });
        GoHeap h;
        for (const Cand &c : cs) h.push(HeapItem{c.row, c.dist});
        h.drain(&res);
        out_offsets[i] = off;
        for (const HeapItem &it : res) {
            if (off < capacity) {
                out_rows[off] = it.row;
                out_dist[off] = it.priority;
            }
            off++;
        }
    }
    out_offsets[n] = off;
    cm->stats.host_us += now_us() - t0;
    if (off > capacity) return fail(SZG_E_TRUNCATED, "radius search: capacity too small");
    return SZG_OK;
}

}  // namespace

}  // namespace szgi

using namespace szgi;

extern "C" {

int szg_comm_unique_id(uint8_t *id)
{
    if (!id) return fail(SZG_E_INVALID, "null argument");
    static_assert(SZG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "communicator id size");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, u.internal, SZG_COMM_ID_BYTES);
    return SZG_OK;
}

void szg_comm_destroy(szg_comm *cm)
{
    if (!cm) return;
    if (cm->nccl) {
        (void)hipSetDevice(cm->device);
        if (cm->stream) (void)hipStreamSynchronize(cm->stream);
        (void)ncclCommDestroy(cm->nccl);
    }
    comm_free_staging(cm);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
}

int szg_comm_create(szg_comm **out, const uint8_t *id, int rank, int world, int device)
{
    SZG_TRY
    if (!out || !id) return fail(SZG_E_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(SZG_E_INVALID, "rank / world out of range");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(SZG_E_NODEVICE, "hipGetDeviceCount", e);
    if (device < 0 || device >= count) return fail(SZG_E_INVALID, "device ordinal out of range");
    szg_comm *cm = new szg_comm();
    cm->rank = rank;
    cm->world = world;
    cm->device = device;
    auto bail = [&](int rc) {
        szg_comm_destroy(cm);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(SZG_E_DEVICE, "hipStreamCreate(exchange stream)"));
    ncclUniqueId u;
    memcpy(u.internal, id, SZG_COMM_ID_BYTES);
    const ncclResult_t r = ncclCommInitRank(&cm->nccl, world, u, rank);  // collective: every rank is in here now
    if (r != ncclSuccess) {
        cm->nccl = nullptr;
        return bail(nccl_fail("ncclCommInitRank", r));
    }
    int n = 0;
    if (ncclCommCount(cm->nccl, &n) == ncclSuccess) cm->stats.rccl_ranks = n;
    // first use of a communicator builds its rings: one small untimed all-gather now, so that no search pays for it
    int rc = comm_reserve(cm, 4096);
    if (rc == SZG_OK) {
        for (int i = 0; i < 8; i++) cm->h_mine[i] = rank;
        rc = comm_exchange(cm, 8);
        for (int g = 0; g < world && rc == SZG_OK; g++)
            if (cm->h_all[(size_t)g * 8] != g) rc = fail(SZG_E_DEVICE, "all-gather self-test returned foreign data");
    }
    if (rc) return bail(rc);
    cm->stats.exchanges = 0;
    cm->stats.exchange_us = 0;
    *out = cm;
    return SZG_OK;
    SZG_CATCH
}

int szg_comm_create_host(szg_comm **out, szg_allgather_fn fn, void *user, int rank, int world)
{
    SZG_TRY
    if (!out || !fn) return fail(SZG_E_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(SZG_E_INVALID, "rank / world out of range");
    szg_comm *cm = new szg_comm();
    cm->rank = rank;
    cm->world = world;
    cm->host_fn = fn;
    cm->host_user = user;
    const int rc = comm_reserve(cm, 4096);
    if (rc) {
        szg_comm_destroy(cm);
        return rc;
    }
    *out = cm;
    return SZG_OK;
    SZG_CATCH
}

int szg_comm_reserve(szg_comm *cm, int n_queries, int k)
{
    SZG_TRY
    if (!cm) return fail(SZG_E_INVALID, "null argument");
    if (n_queries < 0 || k <= 0) return fail(SZG_E_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(cm->mu);
    return comm_reserve(cm, (size_t)n_queries * (2 * ((size_t)k + 1) + 1));
    SZG_CATCH
}

int szg_comm_get_stats(szg_comm *cm, szg_comm_stats *out)
{
    if (!cm || !out) return fail(SZG_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(cm->mu);
    *out = cm->stats;
    return SZG_OK;
}

int szg_comm_reset_stats(szg_comm *cm)
{
    if (!cm) return fail(SZG_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(cm->mu);
    const int n = cm->stats.rccl_ranks;
    cm->stats = szg_comm_stats{};
    cm->stats.rccl_ranks = n;
    return SZG_OK;
}

int szg_comm_merge_topk(szg_comm *cm, int k, int n_queries, const uint64_t *rows, const double *dist,
                        const int32_t *counts, uint64_t *out_rows, double *out_dist, int32_t *out_count,
                        uint8_t *out_history_dependent)
{
    SZG_TRY
    if (!cm || !rows || !dist || !counts || !out_rows || !out_dist) return fail(SZG_E_INVALID, "null argument");
    if (k <= 0 || n_queries < 0) return fail(SZG_E_INVALID, "bad argument");
    if (n_queries == 0) return SZG_OK;
    std::lock_guard<std::mutex> lk(cm->mu);
    return comm_merge_topk(cm, k, n_queries, rows, dist, counts, SZG_OK, out_rows, out_dist, out_count,
                           out_history_dependent);
    SZG_CATCH
}

int szg_comm_merge_radius(szg_comm *cm, int n_queries, const uint64_t *offsets, const uint64_t *rows,
                          const double *dist, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                          uint64_t *out_offsets)
{
    SZG_TRY
    if (!cm || !offsets || !out_offsets) return fail(SZG_E_INVALID, "null argument");
    if (n_queries < 0) return fail(SZG_E_INVALID, "bad argument");
    if (capacity && (!out_rows || !out_dist)) return fail(SZG_E_INVALID, "null output buffer");
    if (offsets[n_queries] > offsets[0] && (!rows || !dist)) return fail(SZG_E_INVALID, "null argument");
    for (int i = 0; i <= n_queries; i++) out_offsets[i] = 0;
    if (n_queries == 0) return SZG_OK;
    std::lock_guard<std::mutex> lk(cm->mu);
    return comm_merge_radius(cm, n_queries, offsets, rows, dist, SZG_OK, out_rows, out_dist, capacity, out_offsets);
    SZG_CATCH
}

int szg_index_attach_comm(szg_index *ix, szg_comm *cm)
{
    if (!ix) return fail(SZG_E_INVALID, "null argument");
    ix->comm = cm;  // borrowed; NULL detaches
    return SZG_OK;
}

int szg_search_topk_sharded(szg_index *ix, const double *queries, int n_queries, int k, const uint64_t *allow_bits,
                            uint64_t *out_rows, double *out_dist, int32_t *out_count, uint8_t *out_history_dependent)
{
    SZG_TRY
    if (!ix || !queries || !out_rows || !out_dist) return fail(SZG_E_INVALID, "null argument");
    if (!ix->comm) return fail(SZG_E_INVALID, "no communicator attached (szg_index_attach_comm)");
    if (n_queries < 0 || k <= 0) return fail(SZG_E_INVALID, "k must be > 0");
    if (n_queries == 0) return SZG_OK;
    szg_comm *cm = ix->comm;
    std::lock_guard<std::mutex> comm_lock(cm->mu);
    const int kk = k + 1;  // one extra per shard so that equal distances at the k boundary stay visible
    // runs of <= 128 queries: one local call and one exchange (one pipeline fill / drain); longer ones in chunks of
    // 256 so that the exchange and merge of a chunk hide behind the next chunk's sweeps
    const int chunk = n_queries <= 128 ? n_queries : 256;
    const int n_chunks = (n_queries + chunk - 1) / chunk;
    const size_t words = (size_t)((szg_index_rows(ix) + 63) / 64);  // of this rank's own mask per query

    std::vector<uint64_t> lrows((size_t)n_queries * kk);
    std::vector<double> ldist((size_t)n_queries * kk);
    std::vector<int32_t> lcount(n_queries);
    std::vector<int> lrc(n_chunks, SZG_OK);
    auto local = [&](int c) {
        const int c0 = c * chunk, n = std::min(chunk, n_queries - c0);
        lrc[c] = szg_search_topk(ix, queries + (size_t)c0 * ix->dim, n, kk,
                                 allow_bits ? allow_bits + (size_t)c0 * words : nullptr, lrows.data() + (size_t)c0 * kk,
                                 ldist.data() + (size_t)c0 * kk, lcount.data() + c0);
    };
    std::mutex mu;
    std::condition_variable cv;
    int ready = 0;
    std::thread producer;
    struct Joiner {  // (an exception below must not leave the worker running on this frame's buffers)
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{producer};
    if (n_chunks > 1) {
        producer = std::thread([&] {
            for (int c = 0; c < n_chunks; c++) {
                local(c);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready = c + 1;
                }
                cv.notify_one();
            }
        });
    }
    int first_err = SZG_OK;
    for (int c = 0; c < n_chunks; c++) {
        if (n_chunks > 1) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready > c; });
        } else {
            local(c);
        }
        const int c0 = c * chunk, n = std::min(chunk, n_queries - c0);
        const int rc = comm_merge_topk(cm, k, n, lrows.data() + (size_t)c0 * kk, ldist.data() + (size_t)c0 * kk,
                                       lcount.data() + c0, lrc[c], out_rows + (size_t)c0 * k, out_dist + (size_t)c0 * k,
                                       out_count ? out_count + c0 : nullptr,
                                       out_history_dependent ? out_history_dependent + c0 : nullptr);
        if (rc && first_err == SZG_OK) first_err = rc;  // (later chunks still take part in their collectives)
    }
    return first_err;
    SZG_CATCH
}

int szg_search_radius_sharded(szg_index *ix, const double *queries, int n_queries, const double *radii,
                              const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                              uint64_t *out_offsets)
{
    SZG_TRY
    if (!ix || !queries || !radii || !out_offsets) return fail(SZG_E_INVALID, "null argument");
    if (!ix->comm) return fail(SZG_E_INVALID, "no communicator attached (szg_index_attach_comm)");
    if (n_queries < 0) return fail(SZG_E_INVALID, "n_queries < 0");
    if (capacity && (!out_rows || !out_dist)) return fail(SZG_E_INVALID, "null output buffer");
    for (int i = 0; i < n_queries; i++)
        if (!(radii[i] > 0)) return fail(SZG_E_INVALID, "radius must be > 0 (collection.go:598)");
    for (int i = 0; i <= n_queries; i++) out_offsets[i] = 0;
    if (n_queries == 0) return SZG_OK;
    szg_comm *cm = ix->comm;
    std::lock_guard<std::mutex> comm_lock(cm->mu);
    // the rank's own hits (an empty shard has none), as CSR with global rows
    std::vector<std::vector<HeapItem>> hits(n_queries);
    int lrc = SZG_OK;
    if (szg_index_rows(ix) > 0) {
        const size_t words = (size_t)((szg_index_rows(ix) + 63) / 64);
        std::vector<const uint64_t *> masks;
        if (allow_bits) {
            masks.resize(n_queries);
            for (int i = 0; i < n_queries; i++) masks[i] = allow_bits + (size_t)i * words;
        }
        lrc = search_radius_impl(ix, queries, n_queries, radii, allow_bits ? masks.data() : nullptr, &hits);
    }
    std::vector<uint64_t> off(n_queries + 1, 0), rows;
    std::vector<double> dist;
    for (int i = 0; i < n_queries; i++) {
        for (const HeapItem &h : hits[i]) {
            rows.push_back(h.row + ix->row_base);
            dist.push_back(h.priority);
        }
        off[i + 1] = rows.size();
    }
    return comm_merge_radius(cm, n_queries, off.data(), rows.data(), dist.data(), lrc, out_rows, out_dist, capacity,
                             out_offsets);
    SZG_CATCH
}

}  // extern "C"
