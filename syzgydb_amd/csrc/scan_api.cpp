// scan_api.cpp -- the search entry points of include/syzgy_scan.h.
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
#include "scan_internal.h"

using namespace szgi;

extern "C" {

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

}  // extern "C"

namespace {

// min(total, capacity) closest hits of a radius search into the caller's buffers (global rows)
int radius_output(const szg_index *ix, const std::vector<HeapItem> &hits, uint64_t *out_rows, double *out_dist,
                  uint64_t capacity, uint64_t *out_total)
{
    const uint64_t total = hits.size();
    *out_total = total;
    for (uint64_t i = 0; i < total && i < capacity; i++) {
        out_rows[i] = hits[i].row + ix->row_base;
        out_dist[i] = hits[i].priority;
    }
    if (total > capacity) return fail(SZG_E_TRUNCATED, "radius search: capacity too small");
    return SZG_OK;
}

// One batch of coalesced single-query callers of the same kind (top-k with the same k, or radius searches -- each
// with its own radius): answered through one call of the batch path, results handed to every caller's own buffers.
struct BatchScratch {
    std::vector<double> q, dist, radii;
    std::vector<uint64_t> rows;
    std::vector<int32_t> count;
    std::vector<const uint64_t *> masks;
    std::vector<std::vector<HeapItem>> hits;
};

void serve_batch(szg_index *ix, const std::vector<PendingSearch *> &batch, BatchScratch &w)
{
    const int nq = (int)batch.size();
    const bool radius = batch[0]->radius > 0;
    int rc = SZG_OK;
    try {
        w.q.resize((size_t)nq * ix->dim);
        w.masks.resize(nq);
        bool any = false;
        for (int i = 0; i < nq; i++) {
            memcpy(&w.q[(size_t)i * ix->dim], batch[i]->query, sizeof(double) * ix->dim);
            w.masks[i] = batch[i]->allow;  // each caller's own filter, if it has one
            any |= w.masks[i] != nullptr;
        }
        if (radius) {
            w.radii.resize(nq);
            for (int i = 0; i < nq; i++) w.radii[i] = batch[i]->radius;
            rc = search_radius_impl(ix, w.q.data(), nq, w.radii.data(), any ? w.masks.data() : nullptr, &w.hits);
            for (int i = 0; i < nq; i++) {
                PendingSearch *p = batch[i];
                p->rc = rc ? rc : radius_output(ix, w.hits[i], p->out_rows, p->out_dist, p->capacity, p->out_total);
            }
            return;
        }
        const int kk = batch[0]->k;
        if (nq == 1) {
            PendingSearch *p = batch[0];
            rc = search_topk_any(ix, p->query, 1, kk, p->allow, p->out_rows, p->out_dist, p->out_count);
        } else {
            w.rows.resize((size_t)nq * kk);
            w.dist.resize((size_t)nq * kk);
            w.count.resize(nq);
            rc = search_topk_any(ix, w.q.data(), nq, kk, nullptr, w.rows.data(), w.dist.data(), w.count.data(),
                                 any ? w.masks.data() : nullptr);
            for (int i = 0; i < nq && rc == SZG_OK; i++) {
                memcpy(batch[i]->out_rows, &w.rows[(size_t)i * kk], sizeof(uint64_t) * kk);
                memcpy(batch[i]->out_dist, &w.dist[(size_t)i * kk], sizeof(double) * kk);
                if (batch[i]->out_count) *batch[i]->out_count = w.count[i];
            }
        }
    } catch (const std::bad_alloc &) {
        rc = fail(SZG_E_NOMEM, "out of memory (host)");
    } catch (...) {
        rc = fail(SZG_E_DEVICE, "unexpected exception");
    }
    for (PendingSearch *p : batch) p->rc = rc;
}

// Search holds only RLock in the reference (collection.go:570), so many goroutines call in at once, each with ONE
// query.  Whoever finds no batch in flight becomes the leader: it answers everything that is waiting and is of one
// kind as one batch (one shared sweep, or one query-major collect launch, instead of one launch per caller), then
// hands the lead to a waiter once its own answer has arrived.  A lone caller is its own batch of one and pays
// nothing for this.
int combine(szg_index *ix, PendingSearch &me)
{
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
        if (me.done) {
            if (me.rc) (void)fail(me.rc, szg_strerror(me.rc));  // (the error text lives on the leader's thread)
            return me.rc;
        }
    }
    ix->comb_leader = true;
    BatchScratch scratch;
    while (!me.done) {
        batch.clear();
        const PendingSearch *head = ix->comb_waiting.front();  // never empty here: `me` is in it until done
        const bool radius = head->radius > 0;
        for (auto it = ix->comb_waiting.begin(); it != ix->comb_waiting.end() && batch.size() < (size_t)kMaxBatch;) {
            const bool same = radius ? (*it)->radius > 0 : ((*it)->radius == 0 && (*it)->k == head->k);
            if (same) {
                batch.push_back(*it);  // within the reserved capacity
                it = ix->comb_waiting.erase(it);
            } else {
                ++it;
            }
        }
        lk.unlock();
        serve_batch(ix, batch, scratch);
        lk.lock();
        for (PendingSearch *p : batch) {
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

}  // namespace

extern "C" {

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
    PendingSearch me;
    me.query = queries;
    me.allow = allow_bits;
    me.k = k;
    me.out_rows = out_rows;
    me.out_dist = out_dist;
    me.out_count = out_count;
    return combine(ix, me);
}

int szg_search_radius(szg_index *ix, const double *query, double radius,
                      const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist,
                      uint64_t capacity, uint64_t *out_total)
{
    if (!ix || !query || !out_total) return fail(SZG_E_INVALID, "null argument");
    if (!(radius > 0)) return fail(SZG_E_INVALID, "radius must be > 0 (collection.go:598)");
    if (capacity && (!out_rows || !out_dist)) return fail(SZG_E_INVALID, "null output buffer");
    *out_total = 0;
    if (szg_index_rows(ix) == 0) return SZG_OK;
    if (!ix->coalesce) {
        SZG_TRY
        std::vector<std::vector<HeapItem>> hits;
        const int rc = search_radius_impl(ix, query, 1, &radius, allow_bits ? &allow_bits : nullptr, &hits);
        if (rc) return rc;
        return radius_output(ix, hits[0], out_rows, out_dist, capacity, out_total);
        SZG_CATCH
    }
    // concurrent callers (the reference's Searches under RLock) share query-major collect launches
    PendingSearch me;
    me.query = query;
    me.allow = allow_bits;
    me.k = 0;
    me.radius = radius;
    me.out_rows = out_rows;
    me.out_dist = out_dist;
    me.capacity = capacity;
    me.out_total = out_total;
    return combine(ix, me);
}

int szg_search_radius_batch(szg_index *ix, const double *queries, int n_queries, const double *radii,
                            const uint64_t *allow_bits, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                            uint64_t *out_offsets)
{
    SZG_TRY
    if (!ix || !queries || !radii || !out_offsets) return fail(SZG_E_INVALID, "null argument");
    if (n_queries < 0) return fail(SZG_E_INVALID, "n_queries < 0");
    if (capacity && (!out_rows || !out_dist)) return fail(SZG_E_INVALID, "null output buffer");
    for (int i = 0; i < n_queries; i++)
        if (!(radii[i] > 0)) return fail(SZG_E_INVALID, "radius must be > 0 (collection.go:598)");
    for (int i = 0; i <= n_queries; i++) out_offsets[i] = 0;
    const uint64_t total_rows = szg_index_rows(ix);
    if (n_queries == 0 || total_rows == 0) return SZG_OK;
    const size_t words = (size_t)((total_rows + 63) / 64);
    std::vector<const uint64_t *> masks;
    if (allow_bits) {
        masks.resize(n_queries);
        for (int i = 0; i < n_queries; i++) masks[i] = allow_bits + (size_t)i * words;
    }
    std::vector<std::vector<HeapItem>> hits;
    const int rc = search_radius_impl(ix, queries, n_queries, radii, allow_bits ? masks.data() : nullptr, &hits);
    if (rc) return rc;
    uint64_t off = 0;
    for (int i = 0; i < n_queries; i++) {
        out_offsets[i] = off;
        for (const HeapItem &h : hits[i]) {
            if (off < capacity) {
                out_rows[off] = h.row + ix->row_base;
                out_dist[off] = h.priority;
            }
            off++;
        }
    }
    out_offsets[n_queries] = off;
    if (off > capacity) return fail(SZG_E_TRUNCATED, "radius search: capacity too small");
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
