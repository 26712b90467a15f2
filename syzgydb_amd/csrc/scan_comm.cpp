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
//
// Nobody hangs because a peer could not take part (round 4).  A collective is only entered with staging every rank
// is known to hold: the ranks AGREE on the staging size in a one-word status round over the minimal staging that
// exists from creation on (comm_agree; only when a call needs more than the last agreed size, so a steady state has
// no extra collective), and a rank whose allocation failed says so there -- every rank then returns an error
// without entering the data exchange.  A rank whose own SEARCH failed still takes part and marks its records with
// a count of -1.  What is left is a transport failure part-way through an exchange (a HIP / RCCL error between the
// enqueue and the wait): the communicator is then unusable, the failing rank returns SZG_E_DEVICE, and its peers
// need the host's own timeout (documented in include/syzgy_scan.h).
//
// RCCL is resolved at run time (dlopen of librccl.so.1 in szg_comm_unique_id / szg_comm_create): a host that never
// shards -- and the host-transport path, which needs no device at all -- carries no RCCL dependency.
#include "scan_internal.h"

#include <rccl/rccl.h>  // types and constants only: the entry points are resolved with dlsym (rccl_api)

#include <dlfcn.h>

#include <thread>

namespace szgi {
namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

// librccl.so.1, once per process; nullptr (and SZG_E_NODEVICE at the call site) when it is not installed
const RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        auto sym = [&](const char *n) { return dlsym(api.lib, n); };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommCount && api.AllGather && api.CommDestroy &&
                 api.GetErrorString;
    });
    return api.ok ? &api : nullptr;
}

}  // namespace
}  // namespace szgi

// The communicator: RCCL (a stream of its own; pinned host staging the collective reads and writes directly, or
// device staging) or the host's transport (plain host memory, no HIP call at all -- usable without a GPU, which is
// how the record path is tested on CPU).
struct szg_comm {
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    szg_allgather_fn host_fn = nullptr;
    void *host_user = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    // staging of one exchange, reused call after call: cap int64 words per rank
    int64_t *h_mine = nullptr, *h_all = nullptr;  // pinned (RCCL) / malloc (host transport)
    int64_t *d_mine = nullptr, *d_all = nullptr;  // RCCL, staged form only
    size_t cap = 0;
    size_t agreed = 0;       // words per rank EVERY rank is known to hold (comm_agree)
    bool zero_copy = false;  // RCCL: the all-gather reads h_mine and writes h_all itself (no H2D / D2H copies)
    int inject_fail = 0;     // test hook (szg_comm_debug_inject): the next staging growth fails
    std::mutex mu;  // collectives must be issued in the same order on every rank: one sharded call at a time
    szg_comm_stats stats{};
    // the merged answer of the last radius call (szg_comm_last_radius: a too-small caller buffer is a LOCAL retry)
    std::vector<uint64_t> last_rows, last_off;
    std::vector<double> last_dist;
};

namespace szgi {

namespace {

int nccl_fail(const char *what, ncclResult_t r)
{
    char buf[256];
    const RcclApi *api = rccl_api();
    snprintf(buf, sizeof(buf), "%s: %s", what, api ? api->GetErrorString(r) : "rccl error");
    return fail(SZG_E_DEVICE, buf);
}
#define NCCLCHK(expr)                                        \
    do {                                                     \
        const ncclResult_t r__ = (expr);                     \
        if (r__ != ncclSuccess) return nccl_fail(#expr, r__); \
    } while (0)

struct Staging {
    int64_t *h_mine = nullptr, *h_all = nullptr, *d_mine = nullptr, *d_all = nullptr;
};
void staging_free(const szg_comm *cm, Staging &s)
{
    if (cm->nccl) {
        (void)hipHostFree(s.h_mine);
        (void)hipHostFree(s.h_all);
        (void)hipFree(s.d_mine);
        (void)hipFree(s.d_all);
    } else {
        free(s.h_mine);
        free(s.h_all);
    }
    s = Staging{};
}

void comm_free_staging(szg_comm *cm)
{
    Staging s{cm->h_mine, cm->h_all, cm->d_mine, cm->d_all};
    staging_free(cm, s);
    cm->h_mine = cm->h_all = cm->d_mine = cm->d_all = nullptr;
    cm->cap = 0;
}

// room for `words` int64 per rank (grown geometrically; szg_comm_reserve sizes it ahead of a timed run).  LOCAL: the
// new buffers are allocated before the old ones go, so a failure leaves the staging the ranks last agreed on intact.
int comm_reserve(szg_comm *cm, size_t words)
{
    if (words <= cm->cap) return SZG_OK;
    if (cm->inject_fail > 0) {
        cm->inject_fail--;
        return fail(SZG_E_NOMEM, "exchange staging (injected failure)");
    }
    const size_t want = std::max<size_t>(std::max(words, cm->cap * 2), 4096);
    Staging n;
    if (cm->nccl) {
        HIPCHK(hipSetDevice(cm->device));
        HIPCHK(hipStreamSynchronize(cm->stream));
        hipError_t e = hipHostMalloc((void **)&n.h_mine, want * sizeof(int64_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&n.h_all, want * sizeof(int64_t) * cm->world, hipHostMallocDefault);
        if (e == hipSuccess && !cm->zero_copy) e = hipMalloc((void **)&n.d_mine, want * sizeof(int64_t));
        if (e == hipSuccess && !cm->zero_copy) e = hipMalloc((void **)&n.d_all, want * sizeof(int64_t) * cm->world);
        if (e != hipSuccess) {
            staging_free(cm, n);
            return fail(SZG_E_NOMEM, "exchange staging", e);
        }
    } else {
        n.h_mine = (int64_t *)malloc(want * sizeof(int64_t));
        n.h_all = (int64_t *)malloc(want * sizeof(int64_t) * cm->world);
        if (!n.h_mine || !n.h_all) {
            staging_free(cm, n);
            return fail(SZG_E_NOMEM, "exchange staging");
        }
    }
    comm_free_staging(cm);
    cm->h_mine = n.h_mine;
    cm->h_all = n.h_all;
    cm->d_mine = n.d_mine;
    cm->d_all = n.d_all;
    cm->cap = want;
    return SZG_OK;
}

// all-gather of `words` int64 per rank: h_mine -> h_all [world][words].  kind: 0 = a data exchange of a search (what
// szg_comm_stats.exchanges counts), 1 = a status round, 2 = a round of the heap chain
int comm_exchange(szg_comm *cm, size_t words, int kind = 0)
{
    const double t0 = now_us();
    if (cm->nccl) {
        const RcclApi *api = rccl_api();
        HIPCHK(hipSetDevice(cm->device));
        if (cm->zero_copy) {
            // the staging is pinned host memory mapped into the device's address space: the collective's kernel
            // reads the rank's records and writes everybody's straight there -- one enqueue and one wait, no copies
            NCCLCHK(api->AllGather(cm->h_mine, cm->h_all, words, ncclInt64, cm->nccl, cm->stream));
        } else {
            HIPCHK(hipMemcpyAsync(cm->d_mine, cm->h_mine, words * sizeof(int64_t), hipMemcpyHostToDevice, cm->stream));
            NCCLCHK(api->AllGather(cm->d_mine, cm->d_all, words, ncclInt64, cm->nccl, cm->stream));
            HIPCHK(hipMemcpyAsync(cm->h_all, cm->d_all, words * sizeof(int64_t) * cm->world, hipMemcpyDeviceToHost,
                                  cm->stream));
        }
        HIPCHK(hipStreamSynchronize(cm->stream));
    } else {
        const int rc = cm->host_fn(cm->host_user, cm->h_mine, cm->h_all, (uint64_t)(words * sizeof(int64_t)));
        if (rc != 0) return fail(SZG_E_DEVICE, "exchange callback failed");
    }
    if (kind == 0) cm->stats.exchanges++;
    else if (kind == 1) cm->stats.status_rounds++;
    else cm->stats.chain_rounds++;
    cm->stats.exchange_us += now_us() - t0;
    return SZG_OK;
}

// Every rank holds staging for `words` int64 per rank -- or every rank returns an error, before any data exchange.
// COLLECTIVE only when `words` exceeds what the ranks last agreed on (sharded calls are collective with the same
// arguments everywhere, so every rank takes the same branch): the rank grows its staging locally, then one status
// word per rank -- its capacity, or -1 -- travels over the staging that already exists.
int comm_agree(szg_comm *cm, size_t words)
{
    if (words <= cm->agreed) return SZG_OK;
    const int lrc = comm_reserve(cm, words);
    const std::string lerr = lrc ? g_last_error : std::string();
    if (!cm->h_mine || cm->cap < 1) return lrc ? lrc : fail(SZG_E_NOMEM, "exchange staging");  // (never had any: creation failed)
    cm->h_mine[0] = lrc == SZG_OK ? (int64_t)cm->cap : -1;
    const int xrc = comm_exchange(cm, 1, 1);
    if (xrc) return xrc;
    int64_t least = INT64_MAX;
    for (int g = 0; g < cm->world; g++) least = std::min(least, cm->h_all[g]);
    if (least < 0) {
        if (lrc) return fail(lrc, lerr.c_str());
        return fail(SZG_E_NOMEM, "a peer rank could not allocate its exchange staging");
    }
    cm->agreed = (size_t)least;
    return SZG_OK;
}

// The rank's top-(k+1) lists of n queries -> all-gather -> consider()'s top-k branch over the union (every rank).
// kk = k + 1 entries per query in rows / dist.  local_rc != 0: this rank's own search failed; it still takes part in
// the collective (the others are waiting in it) and says so with a count of -1.  Caller holds cm->mu and has agreed
// on staging for n records (comm_agree).
int comm_merge_topk(szg_comm *cm, int k, int n, const uint64_t *rows, const double *dist, const int32_t *counts,
                    int local_rc, uint64_t *out_rows, double *out_dist, int32_t *out_count, uint8_t *out_hist)
{
    const int kk = k + 1;
    const size_t w = 2 * (size_t)kk + 1;
    if ((size_t)n * w > cm->cap) return fail(SZG_E_INVALID, "exchange staging not agreed (internal)");
    const double t0 = now_us();
    // one int64 record per query: kk rows | kk distance bit patterns | count
    for (int j = 0; j < n; j++) {
        int64_t *rec = cm->h_mine + (size_t)j * w;
        if (local_rc == SZG_OK) {
            memcpy(rec, rows + (size_t)j * kk, sizeof(uint64_t) * kk);
            memcpy(rec + kk, dist + (size_t)j * kk, sizeof(double) * kk);
            rec[2 * kk] = (int64_t)counts[j];
        } else {
            memset(rec, 0, sizeof(int64_t) * 2 * kk);
            rec[2 * kk] = -1;
        }
    }
    const double t1 = now_us();
    int rc = comm_exchange(cm, (size_t)n * w);
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
// and the pop loop (:694-697): the single-collection answer, ties included, kept in cm->last_* (the caller's buffer
// may be too small on SOME ranks only: fetching it again is local, szg_comm_last_radius).  Caller holds cm->mu.
int comm_merge_radius(szg_comm *cm, int n, const uint64_t *offsets, const uint64_t *rows, const double *dist,
                      int local_rc)
{
    const int G = cm->world;
    cm->last_rows.clear();
    cm->last_dist.clear();
    cm->last_off.assign((size_t)n + 1, 0);
    int rc = comm_agree(cm, (size_t)n);
    if (rc) return rc;
    for (int i = 0; i < n; i++) cm->h_mine[i] = local_rc == SZG_OK ? (int64_t)(offsets[i + 1] - offsets[i]) : -1;
    rc = comm_exchange(cm, (size_t)n);
    if (rc) return rc;
    std::vector<int64_t> counts(cm->h_all, cm->h_all + (size_t)G * n);
    size_t most = 0;
    bool peer_failed = false;
    for (int g = 0; g < G; g++) {
        size_t sum = 0;
        for (int i = 0; i < n; i++) {
            if (counts[(size_t)g * n + i] < 0) peer_failed = true;
            else sum += (size_t)counts[(size_t)g * n + i];
        }
        most = std::max(most, sum);
    }
    if (peer_failed)  // (every rank sees the same counts: every rank leaves here, nobody enters the second exchange)
        return local_rc != SZG_OK ? local_rc : fail(SZG_E_DEVICE, "a peer rank's search failed");
    if (most) {  // every rank's queries back to back, padded to the largest rank
        rc = comm_agree(cm, 2 * most);  // (`most` is the same number on every rank)
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
        for (const HeapItem &it : res) {
            cm->last_rows.push_back(it.row);
            cm->last_dist.push_back(it.priority);
        }
        cm->last_off[i + 1] = cm->last_rows.size();
    }
    cm->stats.host_us += now_us() - t0;
    return SZG_OK;
}

// the kept answer into the caller's buffers: offsets always complete, the first `capacity` entries written
int comm_copy_radius(const szg_comm *cm, int n, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                     uint64_t *out_offsets)
{
    if ((size_t)n + 1 != cm->last_off.size()) return fail(SZG_E_INVALID, "no merged radius answer for this many queries");
    for (int i = 0; i <= n; i++) out_offsets[i] = cm->last_off[i];
    const size_t total = cm->last_rows.size(), m = std::min<size_t>(total, capacity);
    if (m) {
        memcpy(out_rows, cm->last_rows.data(), m * sizeof(uint64_t));
        memcpy(out_dist, cm->last_dist.data(), m * sizeof(double));
    }
    if (total > capacity) return fail(SZG_E_TRUNCATED, "radius search: capacity too small (szg_comm_last_radius fetches it again)");
    return SZG_OK;
}

// ---- equal distances across shards: the reference's order (collection.go:606-619) -------------------------------
//
// When two of the best k+1 merged distances are equal (or one is NaN) the reference's answer depends on its whole
// heap history, which no single rank holds.  The history is sequential in the visit order, and the visit order is
// rank 0's rows, then rank 1's, ...: so the heap travels.  Round g of G: rank g takes the heap as rank g-1 left it
// (container/heap's array, element for element), replays consider() over its own rows in order, and the all-gather
// of round g hands its array on; after round G-1 every rank holds the final heap and pops it.  Costs G small
// all-gathers and one exact pass over every rank's rows, for the flagged queries of a call together -- the price of
// the single-handle path's exact replay (scan_topk.cpp: run_full_replay), paid as rarely.
int comm_chain_topk(szg_comm *cm, int k, int n_flagged, szg_replay_fn replay, void *user, uint64_t *out_rows,
                    double *out_dist, int32_t *out_count)
{
    const size_t w = 2 * (size_t)k + 1;  // per query: k rows | k distance bit patterns | entries in the heap
    int rc = comm_agree(cm, (size_t)n_flagged * w);
    if (rc) return rc;
    std::vector<uint64_t> hrows((size_t)k);
    std::vector<double> hdist((size_t)k);
    int failed = SZG_OK;
    for (int g = 0; g < cm->world; g++) {
        if (g == cm->rank) {
            for (int j = 0; j < n_flagged; j++) {
                int64_t *rec = cm->h_mine + (size_t)j * w;
                int32_t hn = 0;
                bool poisoned = false;  // a rank before this one could not replay: passed on, reported by every rank
                if (g > 0) {  // the heap as the previous rank left it
                    const int64_t *prev = cm->h_all + ((size_t)(g - 1) * n_flagged + j) * w;
                    hn = (int32_t)prev[2 * k];
                    if (hn < 0 || hn > k) {
                        poisoned = true;
                    } else {
                        memcpy(hrows.data(), prev, sizeof(uint64_t) * (size_t)hn);
                        memcpy(hdist.data(), prev + k, sizeof(double) * (size_t)hn);
                    }
                }
                if (!poisoned) {
                    const int r = replay(user, j, k, hrows.data(), hdist.data(), &hn);
                    if (r != SZG_OK || hn < 0 || hn > k) {
                        if (!failed) failed = r != SZG_OK ? r : fail(SZG_E_DEVICE, "replay callback returned a bad heap");
                        poisoned = true;
                    }
                }
                if (poisoned) hn = -1;
                memset(rec, 0, sizeof(int64_t) * w);
                if (hn > 0) {
                    memcpy(rec, hrows.data(), sizeof(uint64_t) * (size_t)hn);
                    memcpy(rec + k, hdist.data(), sizeof(double) * (size_t)hn);
                }
                rec[2 * k] = hn;
            }
        } else {
            memset(cm->h_mine, 0, sizeof(int64_t) * (size_t)n_flagged * w);
        }
        rc = comm_exchange(cm, (size_t)n_flagged * w, 2);
        if (rc) return rc;
    }
    GoHeap h;
    std::vector<HeapItem> res;
    for (int j = 0; j < n_flagged; j++) {
        const int64_t *fin = cm->h_all + ((size_t)(cm->world - 1) * n_flagged + j) * w;
        const int64_t hn = fin[2 * k];
        if (hn < 0) return failed ? failed : fail(SZG_E_DEVICE, "a peer rank's replay failed");
        h.a.clear();
        for (int64_t i = 0; i < hn; i++) {
            HeapItem it{(uint64_t)fin[i], 0.0};
            memcpy(&it.priority, &fin[k + i], sizeof(double));
            h.a.push_back(it);  // (array order: the heap itself, not a re-push)
        }
        h.drain(&res);
        for (int i = 0; i < k; i++) {
            const bool have = i < (int)res.size();
            out_rows[(size_t)j * k + i] = have ? res[i].row : UINT64_MAX;
            out_dist[(size_t)j * k + i] = have ? res[i].priority : 0.0;
        }
        if (out_count) out_count[j] = (int32_t)res.size();
    }
    return SZG_OK;
}

// the replay callback of a handle: consider() over the handle's own rows, continuing the heap it is given
struct IndexReplay {
    szg_index *ix;
    const double *queries;          // of the whole call
    const uint64_t *allow_bits;     // n_queries x words, or nullptr
    size_t words;
    const int *flagged;             // call-level query index of flagged query j
};
int index_replay_cb(void *user, int j, int k, uint64_t *heap_rows, double *heap_dist, int32_t *heap_n)
{
    const IndexReplay *u = static_cast<const IndexReplay *>(user);
    const int qi = u->flagged[j];
    GoHeap h;
    for (int32_t i = 0; i < *heap_n; i++) h.a.push_back(HeapItem{heap_rows[i], heap_dist[i]});
    if (szg_index_rows(u->ix) > 0) {
        const int rc = replay_rows_into_heap(u->ix, u->queries + (size_t)qi * u->ix->dim,
                                             u->allow_bits ? u->allow_bits + (size_t)qi * u->words : nullptr, k, &h);
        if (rc) return rc;
    }
    *heap_n = (int32_t)h.a.size();
    for (size_t i = 0; i < h.a.size(); i++) {
        heap_rows[i] = h.a[i].row;
        heap_dist[i] = h.a[i].priority;
    }
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
    const RcclApi *api = rccl_api();
    if (!api) return fail(SZG_E_NODEVICE, "librccl.so.1 is not loadable (the host transport needs no RCCL: szg_comm_create_host)");
    ncclUniqueId u;
    NCCLCHK(api->GetUniqueId(&u));
    memcpy(id, u.internal, SZG_COMM_ID_BYTES);
    return SZG_OK;
}

void szg_comm_destroy(szg_comm *cm)
{
    if (!cm) return;
    if (cm->nccl) {
        (void)hipSetDevice(cm->device);
        if (cm->stream) (void)hipStreamSynchronize(cm->stream);
        if (const RcclApi *api = rccl_api()) (void)api->CommDestroy(cm->nccl);
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
    const RcclApi *api = rccl_api();
    if (!api) return fail(SZG_E_NODEVICE, "librccl.so.1 is not loadable");
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
    const ncclResult_t r = api->CommInitRank(&cm->nccl, world, u, rank);  // collective: every rank is in here now
    if (r != ncclSuccess) {
        cm->nccl = nullptr;
        return bail(nccl_fail("ncclCommInitRank", r));
    }
    int n = 0;
    if (api->CommCount(cm->nccl, &n) == ncclSuccess) cm->stats.rccl_ranks = n;
    // First use of a communicator builds its rings: small untimed all-gathers now, so that no search pays for it.
    // The first goes through device staging (always valid); the second asks the collective to read and write the
    // pinned host staging directly -- if that returns the right data on this rank, this rank's later exchanges are
    // one enqueue and one wait (the form is a local matter: every rank still issues the same collective).
    int rc = comm_reserve(cm, 4096);
    auto selftest = [&]() -> int {
        for (int i = 0; i < 8; i++) cm->h_mine[i] = rank * 8 + i;
        for (size_t i = 0; i < (size_t)8 * world; i++) cm->h_all[i] = -7;
        int x = comm_exchange(cm, 8, 1);
        for (int g = 0; g < world && x == SZG_OK; g++)
            for (int i = 0; i < 8; i++)
                if (cm->h_all[(size_t)g * 8 + i] != g * 8 + i) x = fail(SZG_E_DEVICE, "all-gather self-test returned foreign data");
        return x;
    };
    if (rc == SZG_OK) rc = selftest();
    // (opt-in, SZG_COMM_ZERO_COPY=1: with one rank it measured the same 18-22 us per exchange as the staged form, and
    // no node with two cards was ever available to measure the collective proper writing over PCIe)
    if (rc == SZG_OK && getenv("SZG_COMM_ZERO_COPY") != nullptr) {
        cm->zero_copy = true;
        if (selftest() != SZG_OK) {  // (every rank has issued the same two collectives either way)
            cm->zero_copy = false;
            g_last_error.clear();
        }
    }
    if (rc) return bail(rc);
    cm->agreed = cm->cap;  // every rank allocates this minimal staging or fails its creation
    cm->stats.exchanges = 0;
    cm->stats.status_rounds = 0;
    cm->stats.exchange_us = 0;
    cm->stats.zero_copy = cm->zero_copy ? 1 : 0;
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
    cm->agreed = cm->cap;
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
    return comm_agree(cm, (size_t)n_queries * (2 * ((size_t)k + 1) + 1));
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
    const int n = cm->stats.rccl_ranks, z = cm->stats.zero_copy;
    cm->stats = szg_comm_stats{};
    cm->stats.rccl_ranks = n;
    cm->stats.zero_copy = z;
    return SZG_OK;
}

int szg_comm_debug_inject(szg_comm *cm, int what, int value)
{
    if (!cm) return fail(SZG_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(cm->mu);
    if (what == 1) cm->inject_fail = value;  // the next `value` staging growths of this rank fail
    else return fail(SZG_E_INVALID, "unknown injection");
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
    const int rc = comm_agree(cm, (size_t)n_queries * (2 * ((size_t)k + 1) + 1));
    if (rc) return rc;
    return comm_merge_topk(cm, k, n_queries, rows, dist, counts, SZG_OK, out_rows, out_dist, out_count,
                           out_history_dependent);
    SZG_CATCH
}

int szg_comm_chain_topk(szg_comm *cm, int k, int n_flagged, szg_replay_fn replay, void *user, uint64_t *out_rows,
                        double *out_dist, int32_t *out_count)
{
    SZG_TRY
    if (!cm || !replay || !out_rows || !out_dist) return fail(SZG_E_INVALID, "null argument");
    if (k <= 0 || n_flagged < 0) return fail(SZG_E_INVALID, "bad argument");
    if (n_flagged == 0) return SZG_OK;
    std::lock_guard<std::mutex> lk(cm->mu);
    return comm_chain_topk(cm, k, n_flagged, replay, user, out_rows, out_dist, out_count);
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
    const int rc = comm_merge_radius(cm, n_queries, offsets, rows, dist, SZG_OK);
    if (rc) return rc;
    return comm_copy_radius(cm, n_queries, out_rows, out_dist, capacity, out_offsets);
    SZG_CATCH
}

int szg_comm_last_radius(szg_comm *cm, int n_queries, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                         uint64_t *out_offsets)
{
    SZG_TRY
    if (!cm || !out_offsets) return fail(SZG_E_INVALID, "null argument");
    if (capacity && (!out_rows || !out_dist)) return fail(SZG_E_INVALID, "null output buffer");
    std::lock_guard<std::mutex> lk(cm->mu);
    return comm_copy_radius(cm, n_queries, out_rows, out_dist, capacity, out_offsets);
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
    // runs of <= 128 queries: one local call and one exchange (one pipeline fill / drain; the exchange is one
    // latency, which splitting the run would pay twice); longer ones in chunks of 256 so that the exchange and
    // merge of a chunk hide behind the next chunk's sweeps
    const int chunk = n_queries <= 128 ? n_queries : 256;
    const int n_chunks = (n_queries + chunk - 1) / chunk;
    const size_t words = (size_t)((szg_index_rows(ix) + 63) / 64);  // of this rank's own mask per query

    // staging every rank holds, before anything can go wrong on one of them alone (collective only on growth)
    int rc = comm_agree(cm, (size_t)chunk * (2 * (size_t)kk + 1));
    if (rc) return rc;

    // From here on this rank enters every exchange of the call, whatever happens to it locally: a failed allocation
    // or search turns into records with a count of -1.
    std::vector<uint64_t> lrows;
    std::vector<double> ldist;
    std::vector<int32_t> lcount;
    std::vector<uint8_t> hist_own;
    int alloc_rc = SZG_OK;
    try {
        lrows.resize((size_t)n_queries * kk);
        ldist.resize((size_t)n_queries * kk);
        lcount.resize(n_queries);
        if (!out_history_dependent) hist_own.resize(n_queries);
    } catch (const std::bad_alloc &) {
        alloc_rc = fail(SZG_E_NOMEM, "out of memory (host)");
    }
    uint8_t *hist = out_history_dependent ? out_history_dependent : hist_own.data();
    std::vector<int> lrc(n_chunks, alloc_rc);
    auto local = [&](int c) {
        if (alloc_rc) return;
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
    bool threaded = false;
    if (n_chunks > 1) {
        try {
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
            threaded = true;
        } catch (...) {
            threaded = false;  // no thread to be had: this one sweeps and exchanges in turn
        }
    }
    int first_err = SZG_OK;
    for (int c = 0; c < n_chunks; c++) {
        if (threaded) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready > c; });
        } else {
            local(c);
        }
        const int c0 = c * chunk, n = std::min(chunk, n_queries - c0);
        rc = comm_merge_topk(cm, k, n, lrows.data() + (size_t)c0 * kk, ldist.data() + (size_t)c0 * kk,
                             lcount.data() + c0, lrc[c], out_rows + (size_t)c0 * k, out_dist + (size_t)c0 * k,
                             out_count ? out_count + c0 : nullptr, alloc_rc ? nullptr : hist + c0);
        if (rc && first_err == SZG_OK) first_err = rc;  // (later chunks still take part in their collectives)
    }
    if (first_err) return first_err;  // (a failure anywhere was seen by every rank: nobody goes on to the chain)

    // equal distances (or a NaN) among the best k+1 of the union: the reference's order depends on its heap history.
    // Every rank sees the same merged lists, so every rank flags the same queries and enters the chain below.
    if (ix->tie_mode == 0) {
        std::vector<int> flagged;
        for (int i = 0; i < n_queries; i++)
            if (hist[i]) flagged.push_back(i);
        if (!flagged.empty()) {
            const int nf = (int)flagged.size();
            std::vector<uint64_t> crows((size_t)nf * k);
            std::vector<double> cdist((size_t)nf * k);
            std::vector<int32_t> ccount(nf);
            IndexReplay u{ix, queries, allow_bits, words, flagged.data()};
            rc = comm_chain_topk(cm, k, nf, index_replay_cb, &u, crows.data(), cdist.data(), ccount.data());
            if (rc) return rc;
            for (int j = 0; j < nf; j++) {
                const int qi = flagged[j];
                memcpy(out_rows + (size_t)qi * k, crows.data() + (size_t)j * k, sizeof(uint64_t) * k);
                memcpy(out_dist + (size_t)qi * k, cdist.data() + (size_t)j * k, sizeof(double) * k);
                if (out_count) out_count[qi] = ccount[j];
            }
            cm->stats.chained_replays += (uint64_t)nf;
        }
    }
    return SZG_OK;
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
    // the rank's own hits (an empty shard has none), as CSR with global rows; a local failure of any kind -- the
    // search, an allocation -- still takes part in the exchange, with counts of -1
    std::vector<uint64_t> off, rows;
    std::vector<double> dist;
    int lrc = SZG_OK;
    try {
        std::vector<std::vector<HeapItem>> hits(n_queries);
        if (szg_index_rows(ix) > 0) {
            const size_t words = (size_t)((szg_index_rows(ix) + 63) / 64);
            std::vector<const uint64_t *> masks;
            if (allow_bits) {
                masks.resize(n_queries);
                for (int i = 0; i < n_queries; i++) masks[i] = allow_bits + (size_t)i * words;
            }
            lrc = search_radius_impl(ix, queries, n_queries, radii, allow_bits ? masks.data() : nullptr, &hits);
        }
        off.assign((size_t)n_queries + 1, 0);
        for (int i = 0; i < n_queries && lrc == SZG_OK; i++) {
            for (const HeapItem &h : hits[i]) {
                rows.push_back(h.row + ix->row_base);
                dist.push_back(h.priority);
            }
            off[i + 1] = rows.size();
        }
    } catch (const std::bad_alloc &) {
        lrc = fail(SZG_E_NOMEM, "out of memory (host)");
    }
    static const uint64_t zero_off[2] = {0, 0};
    const int rc = comm_merge_radius(cm, n_queries, lrc == SZG_OK ? off.data() : zero_off, rows.data(), dist.data(), lrc);
    if (rc) return rc;
    return comm_copy_radius(cm, n_queries, out_rows, out_dist, capacity, out_offsets);
    SZG_CATCH
}

}  // extern "C"
