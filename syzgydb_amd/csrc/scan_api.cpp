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
