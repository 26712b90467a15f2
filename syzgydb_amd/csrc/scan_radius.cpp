// scan_radius.cpp -- radius search, Search{Precision:"exact", Radius > 0} (K is ignored: collection.go:598-605).
//
// A radius search is a COLLECT sweep: every row whose scan key is at or below a threshold derived from the radius
// (plus the key's error bound) is appended to the query's buffer, the hits are re-ranked in float64 and the
// reference's exact predicate `distance <= Radius` (:598) decides.  The sweeps of a batch are walked query-major by
// ONE launch (per-sweep threshold, buffer and counter in ScanArgs), the re-rank reads the hit counters on the
// device, and up to three batches are in flight per shard -- no launch gap and no host round trip between the sweeps
// of a batch, which is what a single query per launch per caller used to cost (5.7 of 6.9 TB/s on cfg5's shard).
#include "scan_internal.h"

namespace szgi {

// key threshold that surely contains every row with distance <= radius
float radius_key_threshold(const szg_index *ix, double radius, const QMeta &meta)
{
    if (ix->metric == SZG_COSINE) {
        if (radius >= 1.0 || meta.m1 == 0) return 3.0e38f;  // acos(c)/pi <= 1 always; zero query -> all 1.0
        const double t = -std::cos(M_PI * radius) + 2.0 * key_eps(ix, 1.0, meta) + 1e-12;
        return std::nextafter((float)t, INFINITY);
    }
    const double scale = ix->bits <= 16 ? (double)((1u << ix->bits) - 1u) : 1.0;
    const double kk = (radius * scale) * (radius * scale);
    const double t = kk * (1.0 + 1e-12) + 2.0 * key_eps(ix, kk, meta);
    // (an infinite radius with a zero query makes t = inf + 0 * inf = NaN: everything, as for any t beyond the floats)
    return !(t < 3.0e38) ? 3.0e38f : std::nextafter((float)t, INFINITY);
}

namespace {

constexpr size_t kRadiusCapMin = 1024;      // entries per sweep a batch's buffers start with
constexpr size_t kRadiusCapMax = 1u << 20;  // beyond this a query goes through run_collect on its own
constexpr size_t kRadiusBlockCopyBytes = 2u << 20;  // a batch's re-ranked hits travel as one block up to this size

// consider()'s radius branch (collection.go:598-605) over the candidates in visit order -- every record with
// distance <= Radius is pushed onto the max-heap -- then the pop loop (:694-697).  Popping a heap of pairwise
// distinct priorities yields them in descending order whatever the push order was, so the result is simply the hits
// sorted by distance; only when two hits share a distance (or one is NaN, which `<=` never admits anyway) does the
// order inside the tie depend on the heap's history, and only then is the heap replayed (three times the work).
void radius_assemble(std::vector<HeapItem> &cs, bool presorted, std::vector<HeapItem> *out)
{   // cs: the hits (distance <= radius already applied); presorted: already ascending by distance (the device's sort)
    const size_t n = cs.size();
    if (!presorted)
        std::sort(cs.begin(), cs.end(), [](const HeapItem &x, const HeapItem &y) { return x.priority < y.priority; });
    bool tie = false;
    for (size_t i = 1; i < n && !tie; i++) tie = cs[i].priority == cs[i - 1].priority;
    if (!tie) {
        out->swap(cs);
        return;
    }
    std::sort(cs.begin(), cs.end(), [](const HeapItem &x, const HeapItem &y) { return x.row < y.row; });
    GoHeap h;
    h.a.reserve(n);
    for (const HeapItem &c : cs) h.push(c);
    h.drain(out);
}

struct RadiusCall;
struct RadiusTicket {
    int first = 0, nq = 0;
    std::vector<Ctx *> ctx;       // one per shard
    std::vector<float> thr;       // key threshold per query (of the sweep that runs: shared or one per query)
    std::vector<float> thr_single; // ... of the single-query collect sweep (a query that overflows is swept again on its own)
    int nb = 0;                   // > 0: the batch shares ONE sweep (query blocks of 16)
    std::vector<size_t> cap;      // per shard: entries per sweep of this batch's buffers
    std::vector<uint8_t> copied;  // per shard: the block of re-ranked hits was copied back at enqueue time
    std::vector<uint8_t> sorted;  // per shard: lists of up to kSortHitsMax hits arrive sorted by distance
    bool any_mask = false;
    bool failed = false;
    RadiusCall *owner = nullptr;
    RadiusTicket() = default;
    RadiusTicket(RadiusTicket &&) = default;
    RadiusTicket(const RadiusTicket &) = delete;
    RadiusTicket &operator=(const RadiusTicket &) = delete;
    ~RadiusTicket();  // a ticket dropped with contexts attached (error return, exception) drains and returns them
};

struct RadiusCall {
    szg_index *ix;
    const double *queries;
    int n_queries;
    const double *radii;
    const uint64_t *const *masks;  // nullable; null entries unfiltered
    std::vector<std::vector<HeapItem>> *results;
    size_t n_sh = 0;
    bool single_batch = false;  // the whole call is one batch (a short call)

    void release(RadiusTicket &t)
    {
        for (size_t s = 0; s < n_sh; s++) {
            if (!t.ctx[s]) continue;
            if (t.failed) {
                (void)hipSetDevice(ix->shards[s]->device);
                (void)hipStreamSynchronize(t.ctx[s]->work);
            }
            ctx_release(ix->shards[s], t.ctx[s]);
        }
        t.ctx.assign(n_sh, nullptr);
    }
    bool acquire(RadiusTicket &t, bool may_block)
    {
        for (size_t s = 0; s < n_sh; s++) {
            if (ix->shards[s]->n_rows == 0) continue;
            Ctx *c = may_block ? ctx_acquire(ix->shards[s]) : ctx_try_acquire(ix->shards[s]);
            if (!c) {
                release(t);
                return false;
            }
            // (a call that is one batch runs on the scan stream from upload to copy-back: scan_topk.cpp, acquire)
            if (single_batch && ix->serialize_scans) c->work = ix->shards[s]->scan_stream;
            t.ctx[s] = c;
        }
        return true;
    }
    int stage(RadiusTicket &t);
    int enqueue_shard(RadiusTicket &t, size_t s);
    int finish(RadiusTicket &t);
    int run();
};

RadiusTicket::~RadiusTicket()
{
    if (!owner) return;
    bool any = false;
    for (Ctx *c : ctx) any |= c != nullptr;
    if (!any) return;
    failed = true;  // (release() then waits for the streams first)
    owner->release(*this);
}

int RadiusCall::enqueue_shard(RadiusTicket &t, size_t s)
{
    Shard *sh = ix->shards[s];
    Ctx *c = t.ctx[s];
    HIPCHK(hipSetDevice(sh->device));
    // buffers: t.nq sweeps x cap entries; cap follows the largest hit count this context has seen
    size_t cap = std::max(kRadiusCapMin, c->radius_cap);
    // (a shared sweep's batch is up to 96 queries: keep its buffers within 8M entries; a query with more hits than
    // its share is swept again on its own)
    if (t.nb > 0) cap = std::min(cap, std::max(kRadiusCapMin, ((size_t)8 << 20) / (size_t)t.nq));
    t.cap[s] = cap;
    int rc = ensure_dev(&c->d_collect, &c->collect_cap, cap * (size_t)t.nq);
    if (rc) return rc;
    rc = ensure_dev(&c->d_out, &c->d_out_cap, cap * (size_t)t.nq);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(c->d_count, 0, sizeof(uint32_t) * (size_t)t.nq * szg::kCandCountStride, c->work));
    HIPCHK(hipEventRecord(c->ev_up, c->work));  // the sweeps must see the queries, the masks and the zeroed counters
    if (t.nb > 0) {  // one shared sweep for the whole batch
        rc = enqueue_collect_mq(ix, sh, c, t.nq, t.nb, t.any_mask, t.thr.data(), cap);
        if (rc) return rc;
    }
    const int qpl = std::max(1, std::min(ix->queries_per_launch, szg::kMaxSweepsPerLaunch));
    std::vector<szg::ScanArgs> a(t.nb > 0 ? 0 : (t.nq + qpl - 1) / qpl);
    for (int j0 = 0; j0 < t.nq && t.nb == 0; j0 += qpl) {  // launches of <= 16 sweeps, back to back
        szg::ScanArgs &x = a[j0 / qpl];
        const int m = std::min(qpl, t.nq - j0);
        fill_scan_args(ix, sh, c, t.any_mask, j0, m, &x);
        x.collect = 1;
        for (int j = 0; j < m; j++) x.thr_ukeys[j] = szg::ordered_key(t.thr[j0 + j]);
        x.collect_buf = c->d_collect + (size_t)j0 * cap;
        x.collect_cap = (uint32_t)cap;
        x.collect_count = c->d_count + (size_t)j0 * szg::kCandCountStride;
        if ((t.any_mask || sh->has_dead) && ix->mask_dense) {  // most rows pass: read every row, masks at the row finish
            double lowest = 1.0;
            for (int j = 0; j < m; j++) lowest = std::min(lowest, mask_pass_rate(sh, c, t.any_mask, j0 + j));
            x.mask_dense = lowest >= 0.5 ? 1 : 0;
        }
    }
    if (t.nb == 0) {
        rc = launch_scans_chained(ix, sh, c, a, scan_geometry(ix, sh, 0));
        if (rc) return rc;
    }
    // float64 distances of the hits: the counts stay on the device
    HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64, c->d_collect, c->d_count,
                              (uint32_t)cap, t.nq, c->d_out, c->work, szg::kCandCountStride));
    // ... and each query's hits sorted by distance there too (lists of up to kSortHitsMax: the host only filters)
    if (ix->radius_sort) HIPCHK(szg::launch_sort_hits(c->d_out, c->d_count, szg::kCandCountStride, (uint32_t)cap, t.nq, c->work));
    t.sorted[s] = ix->radius_sort != 0;
    HIPCHK(hipMemcpyAsync(c->h_count, c->d_count, sizeof(uint32_t) * (size_t)t.nq * szg::kCandCountStride, hipMemcpyDeviceToHost,
                          c->work));
    // the re-ranked hits: while the batch's buffers are small (the usual hundreds of hits per query) the whole block
    // follows in ONE copy right here -- finish() then needs a single wait; larger ones are copied hit list by hit
    // list once the counts are known
    // (a shared sweep's batch of up to 96 queries: a copy call per hit list costs more than 64 KiB of transfer each)
    t.copied[s] = cap * (size_t)t.nq * sizeof(szg::RerankOut) <= std::max(kRadiusBlockCopyBytes, (size_t)t.nq * (64u << 10));
    if (t.copied[s]) {
        rc = ensure_host(&c->h_out, &c->h_out_cap, cap * (size_t)t.nq);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, cap * (size_t)t.nq * sizeof(szg::RerankOut), hipMemcpyDeviceToHost,
                              c->work));
    }
    return SZG_OK;
}

int RadiusCall::stage(RadiusTicket &t)
{
    int rc = SZG_OK;
    const double *q = queries + (size_t)t.first * ix->dim;
    std::vector<const uint64_t *> m(t.nq, nullptr);
    for (int j = 0; j < t.nq; j++) {
        m[j] = masks ? masks[t.first + j] : nullptr;
        t.any_mask |= m[j] != nullptr;
    }
    const double t0 = now_us();
    Ctx *c0 = nullptr;
    const bool int_planes = t.nb > 0 && mq_uses_i8(ix, true);
    for (size_t s = 0; s < n_sh; s++) {
        Ctx *c = t.ctx[s];
        if (!c) continue;
        if (int_planes && !c->h_mqQ) {
            c->h_mqQ = (int32_t *)malloc(sizeof(int32_t) * (size_t)kMaxBatch * ix->dim);
            if (!c->h_mqQ) return fail(SZG_E_NOMEM, "host scratch");
        }
        if (!c0) {
            c0 = c;
            for (int j = 0; j < t.nq; j++) {
                // (a shared sweep stages its own image: the single-query form -- digit planes / swizzled floats, its
                // quantization step and the threshold that goes with it -- is built in finish() for a query whose
                // hits overflow the batch's buffers and which is then swept again on its own.  Building it here for
                // every query was 1.4 us of the 2.5 us of host preparation per query of a cfg5 batch.)
                if (t.nb > 0) {
                    prep_query_meta(ix, q + (size_t)j * ix->dim, &c->meta[j]);
                } else {
                    prep_query(ix, q + (size_t)j * ix->dim, c->h_qsw + (size_t)j * ix->qsw_bytes, &c->meta[j]);
                    t.thr_single[j] = t.thr[j] = radius_key_threshold(ix, radii[t.first + j], c->meta[j]);
                }
                if (t.nb > 0) {  // the shared sweep's arithmetic has its own error bound
                    if (int_planes) prep_mq_int(ix, q + (size_t)j * ix->dim, &c->meta[j], c->h_mqQ + (size_t)j * ix->dim);
                    QMeta m2 = c->meta[j];
                    m2.mq = !int_planes;
                    m2.mq_bf16 = mq_uses_bf16(ix, true);
                    t.thr[j] = radius_key_threshold(ix, radii[t.first + j], m2);
                }
            }
        } else {
            if (t.nb == 0) memcpy(c->h_qsw, c0->h_qsw, ix->qsw_bytes * (size_t)t.nq);
            if (int_planes) memcpy(c->h_mqQ, c0->h_mqQ, sizeof(int32_t) * (size_t)t.nq * ix->dim);
            for (int j = 0; j < t.nq; j++) c->meta[j] = c0->meta[j];
        }
    }
    const double t1 = now_us();
    for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
        if (!t.ctx[s]) continue;
        rc = enqueue_queries(ix, ix->shards[s], t.ctx[s], q, t.nq, t.any_mask ? m.data() : nullptr, t.nb == 0);
        if (rc == SZG_OK) rc = enqueue_shard(t, s);
    }
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    ix->stats.host_prep_us += t1 - t0;
    ix->stats.host_enqueue_us += now_us() - t1;
    return rc;
}

int RadiusCall::finish(RadiusTicket &t)
{
    if (t.failed) {
        release(t);
        return SZG_OK;
    }
    int rc = SZG_OK;
    std::vector<std::vector<HeapItem>> cands(t.nq);  // the hits proper: `distance <= Radius` (collection.go:598) applied here
    std::vector<uint8_t> redo(t.nq, 0);  // more hits than the batch's buffers hold: the query is swept again on its own
    std::vector<uint8_t> presorted(t.nq, 1);  // the query's hits come from ONE shard's device-sorted list
    std::vector<uint8_t> lists(t.nq, 0);
    double t_wait = 0;
    const double t0 = now_us();
    for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
        Ctx *c = t.ctx[s];
        if (!c) continue;
        Shard *sh = ix->shards[s];
        const double tw = now_us();
        hipError_t e = hipSetDevice(sh->device);
        if (e == hipSuccess) e = hipStreamSynchronize(c->work);
        if (e != hipSuccess) return fail(SZG_E_DEVICE, "hipStreamSynchronize", e);
        rc = finish_timing(ix, c);
        if (rc) break;
        const size_t cap = t.cap[s];
        size_t most = 0;
        std::vector<size_t> cnt(t.nq, 0), off(t.nq + 1, 0);
        for (int j = 0; j < t.nq; j++) {
            const size_t n = c->h_count[(size_t)j * szg::kCandCountStride];
            most = std::max(most, n);
            if (n > cap) redo[j] = 1;
            cnt[j] = n > cap ? 0 : n;
            off[j + 1] = off[j] + cnt[j];
        }
        // the next batch on this context starts with room for what this one saw (and shrinks again slowly)
        size_t want = kRadiusCapMin;
        while (want < most + most / 4 && want < kRadiusCapMax) want <<= 1;
        c->radius_cap = std::max(want, c->radius_cap - c->radius_cap / 8);
        if (t.copied[s]) {  // hits of sweep j at h_out + j * cap
            for (int j = 0; j < t.nq; j++) off[j] = (size_t)j * cap;
        } else if (off[t.nq]) {
            // exact-size copies of each sweep's re-ranked hits, back to back in the pinned buffer
            rc = ensure_host(&c->h_out, &c->h_out_cap, off[t.nq]);
            if (rc) break;
            for (int j = 0; j < t.nq; j++) {
                if (!cnt[j]) continue;
                e = hipMemcpyAsync(c->h_out + off[j], c->d_out + (size_t)j * cap, cnt[j] * sizeof(szg::RerankOut),
                                   hipMemcpyDeviceToHost, c->work);
                if (e != hipSuccess) return fail(SZG_E_DEVICE, "hipMemcpyAsync(radius hits)", e);
            }
            e = hipStreamSynchronize(c->work);
            if (e != hipSuccess) return fail(SZG_E_DEVICE, "hipStreamSynchronize", e);
        }
        t_wait += now_us() - tw;
        for (int j = 0; j < t.nq; j++) {
            if (redo[j]) continue;
            cands[j].reserve(cands[j].size() + cnt[j]);
            if (cnt[j]) {
                if (!t.sorted[s] || cnt[j] > (size_t)szg::kSortHitsMax || ++lists[j] > 1) presorted[j] = 0;
            }
            const double rad = radii[t.first + j];
            for (size_t i = off[j]; i < off[j] + cnt[j]; i++) {
                const szg::RerankOut &r = c->h_out[i];
                if (r.dist <= rad) cands[j].push_back(HeapItem{sh->first + r.row, r.dist});
            }
        }
    }
    for (int j = 0; j < t.nq && rc == SZG_OK; j++) {
        if (redo[j]) {
            cands[j].clear();
            std::vector<Cand> all;
            const double tw = now_us();
            if (t.nb > 0) {  // the single-query form of this query, now that it is needed (see stage())
                const double *qj = queries + (size_t)(t.first + j) * ix->dim;
                for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
                    Ctx *c = t.ctx[s];
                    if (!c) continue;
                    QMeta m;
                    prep_query(ix, qj, c->h_qsw + (size_t)j * ix->qsw_bytes, &m);
                    c->meta[j] = m;
                    t.thr_single[j] = radius_key_threshold(ix, radii[t.first + j], m);
                    hipError_t e = hipSetDevice(ix->shards[s]->device);
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(c->d_qsw + (size_t)j * ix->qsw_bytes, c->h_qsw + (size_t)j * ix->qsw_bytes,
                                           ix->qsw_bytes, hipMemcpyHostToDevice, c->work);
                    if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipMemcpyAsync(single-query form)", e);
                }
            }
            for (size_t s = 0; s < n_sh && rc == SZG_OK; s++)
                if (t.ctx[s]) rc = run_collect(ix, ix->shards[s], t.ctx[s], j, t.thr_single[j], t.any_mask, &all);
            t_wait += now_us() - tw;
            if (rc) break;
            const double rad = radii[t.first + j];
            for (const Cand &c : all)
                if (c.dist <= rad) cands[j].push_back(HeapItem{c.row, c.dist});
            presorted[j] = 0;
        }
        radius_assemble(cands[j], presorted[j] != 0, &(*results)[t.first + j]);
    }
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->stats.host_finish_us += now_us() - t0 - t_wait;
        if (rc == SZG_OK) ix->stats.queries += t.nq;
    }
    release(t);
    return rc;
}

int RadiusCall::run()
{
    n_sh = ix->shards.size();
    results->assign(n_queries, {});
    std::deque<RadiusTicket> inflight;
    int rc = SZG_OK;
    const int qpl = std::max(1, std::min(ix->queries_per_launch, szg::kMaxSweepsPerLaunch));
    for (int q0 = 0; q0 < n_queries && rc == SZG_OK;) {
        RadiusTicket t;
        t.owner = this;
        t.first = q0;
        // a small first batch (the card starts sweeping after a few queries' preparation) and a small last one
        // (what is left to do once the last sweep has ended is that batch's result assembly)
        const int left = n_queries - q0;
        const int edge = std::max(1, std::min(qpl, ix->first_batch > 0 ? ix->first_batch : qpl));
        t.nq = std::min(qpl, left);
        // two or more queries left: they share one sweep of the corpus (up to 96 per pass), as top-k batches do
        t.nb = ix->radius_mq && left >= 2 ? mq_blocks(ix, left, true) : 0;
        if (t.nb > 0) {
            const int groups = t.nb == 3 && mq_uses_i8(ix, true) && ix->mq_i8_groups > 1 && left > 48 &&
                                       szg::mq_i8_lds_bytes(ix->bits, ix->map.r16, 3, 2) <= 160u * 1024u
                                   ? 2 : 1;
            t.nq = std::min(left, 16 * t.nb * groups);
        }
        // (only the smallest calls are ONE batch here: a radius query's result assembly -- hundreds of hits to sort --
        // takes the host ~15 us, which a longer call hides behind the next batch's sweeps)
        single_batch = t.nb == 0 && q0 == 0 && ix->short_call > 0 && n_queries <= std::min(edge, ix->short_call);
        if (single_batch) t.nq = n_queries;
        else if (t.nb > 0) ;  // (a shared sweep takes what fits its image)
        else if (q0 == 0 && left > edge) t.nq = edge;
        else if (left > edge && left <= qpl + edge) t.nq = left - edge;
        t.ctx.assign(n_sh, nullptr);
        t.cap.assign(n_sh, 0);
        t.copied.assign(n_sh, 0);
        t.sorted.assign(n_sh, 0);
        t.thr.assign(t.nq, 0.0f);
        t.thr_single.assign(t.nq, 0.0f);
        if (!acquire(t, inflight.empty())) {
            rc = finish(inflight.front());
            inflight.pop_front();
            continue;
        }
        rc = stage(t);
        t.failed = rc != SZG_OK;
        inflight.push_back(std::move(t));
        q0 += inflight.back().nq;
    }
    while (!inflight.empty()) {
        const int r2 = finish(inflight.front());
        if (rc == SZG_OK) rc = r2;
        inflight.pop_front();
    }
    return rc;
}

}  // namespace

int search_radius_impl(szg_index *ix, const double *queries, int n_queries, const double *radii,
                       const uint64_t *const *masks, std::vector<std::vector<HeapItem>> *results)
{
    RadiusCall call{ix, queries, n_queries, radii, masks, results};
    return call.run();
}

}  // namespace szgi
