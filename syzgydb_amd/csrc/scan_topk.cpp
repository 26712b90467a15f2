// scan_topk.cpp -- the one-sweep-per-query pipeline: enqueue, certification, escalation, exact replay.
#include "scan_internal.h"

namespace szgi {

LaunchGeom scan_geometry(const szg_index *ix, const Shard *sh, int kp, bool plain_topk)
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
int enqueue_queries(szg_index *ix, Shard *sh, Ctx *c, const double *q, int nq, const uint64_t *const *masks,
                    bool with_single_form)
{
    HIPCHK(hipSetDevice(sh->device));
    memcpy(c->h_q64, q, sizeof(double) * ix->dim * nq);
    if (ix->timing >= 2) {
        SiteScope t_(10);
        HIPCHK(hipEventRecord(c->ev_all0, c->work));
    }
    {
        SiteScope t_(0);
        if (with_single_form)
            HIPCHK(hipMemcpyAsync(c->d_qsw, c->h_qsw, ix->qsw_bytes * nq, hipMemcpyHostToDevice, c->work));
        HIPCHK(hipMemcpyAsync(c->d_q64, c->h_q64, sizeof(double) * ix->dim * nq, hipMemcpyHostToDevice,
                              c->work));
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
                              hipMemcpyHostToDevice, c->work));
    }
    // what the sweeps wait for ends here: work enqueued on this stream afterwards (the first-k
    // rows' distances) runs beside the sweeps
    HIPCHK(hipEventRecord(c->ev_up, c->work));
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
                         const LaunchGeom &g, hipStream_t after, int part)
{
    if (!after) after = c->work;
    const int n = (int)a.size();
    {
        std::lock_guard<std::mutex> lk(sh->chain_mu);
        hipStream_t st = ix->serialize_scans ? sh->scan_stream : c->work;
        if (st != c->work) {
            SiteScope t_(1);
            HIPCHK(hipStreamWaitEvent(st, c->ev_up, 0));  // recorded by enqueue_queries
        }
        if (ix->timing) {
            SiteScope t_(2);
            HIPCHK(hipEventRecord(part ? c->ev_p0 : c->ev_scan0, st));
        }
        {
            SiteScope t_(3);
            for (int j = 0; j < n; j++)
                HIPCHK(szg::launch_scan(ix->bits, ix->metric, a[j], g.grid, g.block, st));
        }
        if (ix->timing) {
            SiteScope t_(4);
            HIPCHK(hipEventRecord(part ? c->ev_p1 : c->ev_scan1, st));
            if (part) {
                c->timed_part = true;
                c->timed_part_n = n;
            } else {
                c->timed_scan = true;
                c->timed_n = n;
            }
        }
        if (st != after) {
            SiteScope t_(5);
            HIPCHK(hipEventRecord(c->ev_scan_done, st));
            HIPCHK(hipStreamWaitEvent(after, c->ev_scan_done, 0));
        }
    }
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    uint64_t sweeps = 0;
    for (const szg::ScanArgs &x : a) sweeps += (uint64_t)x.n_queries;
    ix->stats.scan_launches += n;
    ix->stats.scan_bytes += sweeps * sh->n_rows * (uint64_t)ix->row_bytes;
    return SZG_OK;
}

// Pass rates are estimated from a sample of each mask's words.
double mask_pass_rate(const Shard *sh, const Ctx *c, bool has_allow, int slot)
{
    const double live = sh->n_rows ? (double)sh->n_live / (double)sh->n_rows : 1.0;
    if (!has_allow) return live;
    const size_t words = shard_words(sh);
    const uint64_t *m = c->h_allow + (size_t)slot * words;
    const size_t step = std::max<size_t>(1, words / 256);
    uint64_t ones = 0, seen = 0;
    for (size_t w = 0; w < words; w += step) {
        ones += (uint64_t)__builtin_popcountll(m[w]);
        seen += 64;
    }
    return live * (seen ? (double)ones / (double)seen : 1.0);
}

// top-k pass for the nq staged queries of one shard: scan -> merges -> rerank -> D2H (async)
int enqueue_topk(szg_index *ix, Shard *sh, Ctx *c, int kp, int nq, bool has_allow)
{
    c->kp_used = kp;
    c->out_stride = kp;
    c->sent_in_out = false;
    c->mq_stage2 = false;
    c->mq_band_used = false;
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
    // keeps the form that tests a row before issuing its loads.
    auto pass_rate = [&](int j) { return mask_pass_rate(sh, c, has_allow, j); };
    const bool masked = has_allow || sh->has_dead;
    const int qpl = std::max(1, ix->queries_per_launch);
    // One part -- or, for a short call (c->early_n), two: the sweeps of queries [0, early_n), whose merges, re-rank and
    // copy-back leave the scan stream for the context's own (one event), and the last few queries, whose tail is all
    // that is left to do after the call's final sweep.
    const int early = c->early_n > 0 && c->early_n < nq ? c->early_n : 0;
    for (int part = early ? 1 : 0; part >= 0; part--) {
        const int q0 = part ? 0 : early, q1 = part ? early : nq, nqp = q1 - q0;
        hipStream_t tl = part ? c->stream : c->work;
        std::vector<szg::ScanArgs> args((nqp + qpl - 1) / qpl);
        for (int j = q0; j < q1; j += qpl) {  // one sweep per query, results side by side
            szg::ScanArgs &a = args[(j - q0) / qpl];
            fill_scan_args(ix, sh, c, has_allow, j, std::min(qpl, q1 - j), &a);
            if (masked && ix->mask_dense) {
                double lowest = 1.0;
                for (int i = j; i < std::min(q1, j + qpl); i++) lowest = std::min(lowest, pass_rate(i));
                a.mask_dense = lowest >= 0.5 ? 1 : 0;
            }
            a.kp = kp;
            a.block_lists = c->d_lists_a + (size_t)j * g.grid * kp;
        }
        rc = launch_scans_chained(ix, sh, c, args, g, tl, part);
        if (rc) return rc;

        int n_lists = g.grid;
        uint64_t *src = c->d_lists_a + (size_t)q0 * g.grid * kp, *dst = c->d_lists_b + (size_t)q0 * g.grid * kp;
        const int fan = szg::merge_fan(kp);
        {
            SiteScope t_(6);
            while (n_lists > 1) {
                HIPCHK(szg::launch_merge(src, n_lists, kp, nqp, dst, tl));
                n_lists = (n_lists + fan - 1) / fan;
                std::swap(src, dst);
            }
        }
        {
            SiteScope t_(7);
            HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64 + (size_t)q0 * ix->dim,
                                      src, nullptr, (uint32_t)kp, nqp, c->d_out + (size_t)q0 * kp, tl));
        }
        {
            SiteScope t_(8);
            HIPCHK(hipMemcpyAsync(c->h_out + (size_t)q0 * kp, c->d_out + (size_t)q0 * kp, sizeof(szg::RerankOut) * kp * nqp,
                                  hipMemcpyDeviceToHost, tl));
        }
    }
    if (ix->timing >= 2) {
        SiteScope t_(10);
        HIPCHK(hipEventRecord(c->ev_all1, c->work));
    }
    return SZG_OK;
}

int finish_timing(szg_index *ix, Ctx *c)
{
    if (!ix->timing) return SZG_OK;
    float ms_scan = 0, ms_all = 0, ms_part = 0;
    if (c->timed_scan) HIPCHK(hipEventElapsedTime(&ms_scan, c->ev_scan0, c->ev_scan1));
    if (c->timed_part) {  // the early part of a short call
        HIPCHK(hipEventElapsedTime(&ms_part, c->ev_p0, c->ev_p1));
        ms_scan += ms_part;
        c->timed_n = (c->timed_scan ? c->timed_n : 0) + c->timed_part_n;
        c->timed_scan = true;
        c->timed_part = false;
    }
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
        const szg::RerankOut &r = c->h_out[(size_t)slot * c->out_stride + i];
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
        QMeta bm = m;
        bm.mq_bf16 = true;
        const float thr = c->h_thr[slot];
        if (thr < 3.0e38f) *lb = std::min(*lb, (double)thr - key_eps(ix, thr, bm));
        // collected, but outside the band that was scored again in float32: bfloat16 key above the band's edge
        if (c->mq_band_used) {
            const float edge = c->h_thr[128 + slot];
            if (edge < 3.0e38f) *lb = std::min(*lb, (double)edge - key_eps(ix, edge, bm));
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
        if (ix->timing >= 2) HIPCHK(hipEventRecord(c->ev_all0, c->work));
        HIPCHK(hipMemsetAsync(c->d_count, 0, sizeof(uint32_t), c->work));
        HIPCHK(hipEventRecord(c->ev_up, c->work));  // the sweep must see the zeroed counter
        std::vector<szg::ScanArgs> a(1);
        fill_scan_args(ix, sh, c, has_allow, slot, 1, &a[0]);
        a[0].collect = 1;
        a[0].thr_ukeys[0] = szg::ordered_key(thr_key);
        a[0].collect_buf = c->d_collect;
        a[0].collect_cap = (uint32_t)std::min<size_t>(c->collect_cap, 0xFFFFFFFFu);
        a[0].collect_count = c->d_count;
        const LaunchGeom g = scan_geometry(ix, sh, 0);
        rc = launch_scans_chained(ix, sh, c, a, g);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->h_count, c->d_count, sizeof(uint32_t), hipMemcpyDeviceToHost,
                              c->work));
        if (ix->timing >= 2) HIPCHK(hipEventRecord(c->ev_all1, c->work));
        HIPCHK(hipStreamSynchronize(c->work));
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
                                  c->d_out, c->work));
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * count,
                              hipMemcpyDeviceToHost, c->work));
        HIPCHK(hipStreamSynchronize(c->work));
        cands->reserve(cands->size() + count);
        for (uint32_t i = 0; i < count; i++) {
            const szg::RerankOut &r = c->h_out[i];
            cands->push_back(Cand{sh->first + r.row, r.dist, szg::key_from_ordered(r.ukey), 0.0});
        }
        return SZG_OK;
    }
}

// Exact replay of the reference loop over EVERY row (collection.go:672-684 with
// consider(), :583-629): float64 distances for all rows on the device, then the
// heap on the host in visit order.  Bit-faithful in every case, used only when
// history_dependent() says the fast answer could differ.
// (h: the heap to continue -- empty for a search of this handle alone; row_add: what makes a row of this handle global)
static int replay_all_rows(szg_index *ix, std::vector<Ctx *> &ctx, int slot, const uint64_t *allow, int k, GoHeap &h,
                           uint64_t row_add)
{
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
                                  c->d_out, c->work));
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * n, hipMemcpyDeviceToHost,
                              c->work));
        std::vector<uint64_t> live((n + 63) / 64, ~0ull);
        if (sh->has_dead)
            HIPCHK(hipMemcpyAsync(live.data(), sh->live_bits, live.size() * sizeof(uint64_t),
                                  hipMemcpyDeviceToHost, c->work));
        HIPCHK(hipStreamSynchronize(c->work));
        const uint64_t *aw = allow ? allow + sh->first / 64 : nullptr;
        for (size_t r = 0; r < n; r++) {
            if (!((live[r >> 6] >> (r & 63)) & 1)) continue;       // removed record
            if (aw && !((aw[r >> 6] >> (r & 63)) & 1)) continue;   // collection.go:592-594
            h.consider_topk(sh->first + r + row_add, c->h_out[r].dist, k);
        }
    }
    return SZG_OK;
}

int run_full_replay(szg_index *ix, std::vector<Ctx *> &ctx, int slot, const uint64_t *allow, int k,
                    std::vector<HeapItem> *res)
{
    GoHeap h;
    const int rc = replay_all_rows(ix, ctx, slot, allow, k, h, 0);
    if (rc) return rc;
    h.drain(res);
    return SZG_OK;
}

// consider()'s top-k branch over every row of this handle in visit order, CONTINUING the heap `h` (rows global:
// + row_base) -- one link of the rank-to-rank chain that settles equal distances across shards (scan_comm.cpp)
int replay_rows_into_heap(szg_index *ix, const double *query, const uint64_t *allow, int k, GoHeap *h)
{
    std::vector<Ctx *> ctx(ix->shards.size(), nullptr);
    struct Release {
        szg_index *ix;
        std::vector<Ctx *> &ctx;
        ~Release()
        {
            for (size_t s = 0; s < ctx.size(); s++)
                if (ctx[s]) ctx_release(ix->shards[s], ctx[s]);
        }
    } release{ix, ctx};
    for (size_t s = 0; s < ix->shards.size(); s++) {
        Shard *sh = ix->shards[s];
        if (sh->n_rows == 0) continue;
        Ctx *c = ctx[s] = ctx_acquire(sh);
        HIPCHK(hipSetDevice(sh->device));
        memcpy(c->h_q64, query, sizeof(double) * ix->dim);
        HIPCHK(hipMemcpyAsync(c->d_q64, c->h_q64, sizeof(double) * ix->dim, hipMemcpyHostToDevice, c->work));
    }
    const int rc = replay_all_rows(ix, ctx, 0, allow, k, *h, ix->row_base);
    if (rc == SZG_OK) {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->stats.full_replays++;
    }
    return rc;
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
int launch_sentinel_rerank(szg_index *ix, Shard *sh, Ctx *c, int nq, hipStream_t stream)
{
    const size_t total = (size_t)c->sent_n * (size_t)nq;
    if (!total) return SZG_OK;
    int rc = ensure_host(&c->h_sent_out, &c->h_sent_out_cap, total);
    if (rc) return rc;
    rc = ensure_dev(&c->d_sent_out, &c->d_sent_out_cap, total);
    if (rc) return rc;
    HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64, c->d_sent, nullptr,
                              (uint32_t)c->sent_n, nq, c->d_sent_out, stream));
    HIPCHK(hipMemcpyAsync(c->h_sent_out, c->d_sent_out, total * sizeof(szg::RerankOut), hipMemcpyDeviceToHost, stream));
    c->sent_deferred = false;
    return SZG_OK;
}

// defer: only stage the rows (the batch's tail computes their distances in its one rerank launch).
// The sentinels always ride on the context's OWN stream: they have the whole batch's sweeps to finish in, so when
// the batch itself runs on the scan stream (a short call) they wait for its uploads through an event and stay off
// the critical path.
int enqueue_sentinels(szg_index *ix, Shard *sh, Ctx *c, const std::vector<std::vector<uint64_t>> &lists, int nq, bool defer)
{
    c->sent_n = 0;
    c->sent_deferred = false;
    c->sent_own_stream = false;
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
    for (int j = 0; j < nq; j++) {
        size_t n = 0;
        for (uint64_t r : lists[j])
            if (r >= sh->first && r < sh->first + sh->n_rows) c->h_sent[(size_t)j * most + n++] = r - sh->first;
        for (; n < most; n++) c->h_sent[(size_t)j * most + n] = szg::kInvalidCand;
    }
    hipStream_t st = c->work;
    if (!defer && c->work != c->stream) {
        st = c->stream;
        c->sent_own_stream = true;
        HIPCHK(hipStreamWaitEvent(st, c->ev_up, 0));  // the queries are up (recorded by enqueue_queries on c->work)
    }
    HIPCHK(hipMemcpyAsync(c->d_sent, c->h_sent, total * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    c->sent_n = (int)most;
    c->sent_deferred = true;
    if (defer) return SZG_OK;
    return launch_sentinel_rerank(ix, sh, c, nq, st);
}

// ---- one szg_search_topk call -------------------------------------------------------------------------------------
//
// The queries travel in batches ("tickets": up to 16 with one sweep each, or up to 96 sharing one sweep), as many in
// flight as the shards have free contexts.  stage() prepares and enqueues a batch on every shard; finish() waits
// for it and runs the reference's result assembly per query (settle()).
constexpr int kShortCallLast = 4;  // queries of a short call whose tail is left for after the final sweep

struct TopkCall {
    szg_index *ix;
    const double *queries;
    int n_queries, k;
    const uint64_t *allow_bits;          // n_queries masks back to back, or nullptr
    const uint64_t *const *allow_ptrs;   // used instead when given: one mask pointer per query, null = unfiltered
    uint64_t *out_rows;
    double *out_dist;
    int32_t *out_count;

    size_t n_sh = 0, allow_stride = 0;
    int kp = 0;
    bool replay_all = false;  // K beyond the fused selection: every query takes the exact replay
    bool single_batch = false;  // the whole call is one batch of sweeps (a short call)
    int early_n = 0;            // ... whose first early_n queries' merges / re-rank / copy-back / assembly run beside its
                                // last sweeps (Ctx::early_n)

    const uint64_t *mask_of(int qi) const
    {
        if (allow_ptrs) return allow_ptrs[qi];
        return allow_bits ? allow_bits + (size_t)qi * allow_stride : nullptr;
    }
    void release(Ticket &t)
    {
        for (size_t s = 0; s < n_sh; s++)
            if (t.ctx[s]) ctx_release(ix->shards[s], t.ctx[s]);
        t.ctx.assign(n_sh, nullptr);
    }

    int run();
    bool acquire(Ticket &t, bool may_block);
    int stage(Ticket &t, int nb, bool bf16_sweep);
    int wait_shards(Ticket &t);
    int stage_single_form(Ticket &t, int j);
    void gather(Ticket &t, std::vector<std::vector<Cand>> *all, std::vector<double> *thr_min,
                std::vector<uint8_t> *nan_first, int j0, int j1);
    int settle(Ticket &t, int j, std::vector<Cand> &cands, double thr_min, bool nan_first, double *t_dev,
               std::vector<HeapItem> *res, bool *defer = nullptr);
    int finish(Ticket &t);
};

// one context per shard; never block while holding in-flight work
bool TopkCall::acquire(Ticket &t, bool may_block)
{
    for (size_t s = 0; s < n_sh; s++) {
        if (ix->shards[s]->n_rows == 0) continue;
        Ctx *c = may_block ? ctx_acquire(ix->shards[s]) : ctx_try_acquire(ix->shards[s]);
        if (!c) {
            release(t);
            return false;
        }
        // A call that is ONE batch has nothing to overlap with: uploads, sweeps, merges, re-rank and copy-back go
        // onto the shard's scan stream in order.  On the context's own stream every hand-over to and from the scan
        // stream is a cross-queue event wait, and those cost 20-100 us each on this platform (rocprofv3 timeline of
        // 20-query calls on a 125 K-row shard: the sweeps of a call's batches sat 24-105 us apart).
        if (single_batch && ix->serialize_scans) {
            c->work = ix->shards[s]->scan_stream;
            c->early_n = early_n;
        }
        t.ctx[s] = c;
    }
    return true;
}

// prepare the ticket's queries ONCE (swizzled / digit-plane forms, constants; the other shards get copies) and
// enqueue uploads, the first-k rows' distances and the sweeps on every shard
int TopkCall::stage(Ticket &t, int nb, bool bf16_sweep)
{
    int rc = SZG_OK;
    const double *q = queries + (size_t)t.first * ix->dim;
    std::vector<const uint64_t *> masks(t.nq);
    for (int j = 0; j < t.nq; j++) {
        masks[j] = mask_of(t.first + j);
        t.any_mask |= masks[j] != nullptr;
    }
    const uint64_t *const *mptr = t.any_mask ? masks.data() : nullptr;
    const double t_prep0 = now_us();
    Ctx *c0 = nullptr;
    const bool int_planes = nb > 0 && mq_uses_i8(ix, false, t.nq);
    t.lazy_single = nb > 0 && !replay_all;
    for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
        if (ix->shards[s]->n_rows == 0) continue;
        Ctx *cx = t.ctx[s];
        if (int_planes && !cx->h_mqQ) {
            cx->h_mqQ = (int32_t *)malloc(sizeof(int32_t) * (size_t)kMaxBatch * ix->dim);
            if (!cx->h_mqQ) {
                rc = fail(SZG_E_NOMEM, "host scratch");  // the ticket is still finished by the caller
                break;
            }
        }
        if (!c0) {
            c0 = cx;
            for (int j = 0; j < t.nq; j++) {
                // (a shared sweep stages its own image; the single-query form is built if a query escalates)
                if (t.lazy_single) prep_query_meta(ix, q + (size_t)j * ix->dim, &t.meta[j]);
                else prep_query(ix, q + (size_t)j * ix->dim, cx->h_qsw + (size_t)j * ix->qsw_bytes, &t.meta[j]);
                t.meta[j].mq = nb > 0 && !mq_uses_i8(ix, false, t.nq);  // the integer sweeps keep the integer bound
                t.meta[j].mq_bf16 = bf16_sweep;
                if (int_planes) prep_mq_int(ix, q + (size_t)j * ix->dim, &t.meta[j], cx->h_mqQ + (size_t)j * ix->dim);
                cx->meta[j] = t.meta[j];
            }
        } else {
            if (!t.lazy_single) memcpy(cx->h_qsw, c0->h_qsw, ix->qsw_bytes * (size_t)t.nq);
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
        t.ctx[s]->sent_deferred = false;
        rc = enqueue_queries(ix, sh, t.ctx[s], q, t.nq, mptr, !t.lazy_single);
        // (before the sweeps: on the context's stream this runs while the scan stream sweeps; a shared sweep whose
        // tail is the refine launch takes the rows along in its one rerank instead)
        if (rc == SZG_OK && !sent.empty())
            rc = enqueue_sentinels(ix, sh, t.ctx[s], sent, t.nq,
                                   nb > 0 && mq_tail_takes_sentinels(ix, sh, t.kp, t.kp_wide, t.nq, nb));
        if (rc == SZG_OK && !replay_all)
            rc = nb ? enqueue_topk_mq(ix, sh, t.ctx[s], t.kp, t.kp_wide, t.nq, nb, t.any_mask)
                    : enqueue_topk(ix, sh, t.ctx[s], kp, t.nq, t.any_mask);
        if (rc == SZG_OK && replay_all && ix->timing >= 2) {
            const hipError_t e = hipEventRecord(t.ctx[s]->ev_all1, t.ctx[s]->work);
            if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipEventRecord", e);
        }
    }
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    const double t_end = now_us();
    ix->stats.host_prep_us += t_enq0 - t_prep0;
    ix->stats.host_enqueue_us += t_end - t_enq0;
    return rc;
}

// query j of a shared-sweep batch in the single-query kernels' form (swizzled floats / digit planes + their
// constants), on every shard: built only when the query escalates
int TopkCall::stage_single_form(Ticket &t, int j)
{
    const double *q = queries + (size_t)(t.first + j) * ix->dim;
    for (size_t s = 0; s < n_sh; s++) {
        Ctx *c = t.ctx[s];
        if (!c) continue;
        QMeta m;
        prep_query(ix, q, c->h_qsw + (size_t)j * ix->qsw_bytes, &m);
        t.meta[j].qscale = c->meta[j].qscale = m.qscale;
        t.meta[j].qconst = c->meta[j].qconst = m.qconst;
        HIPCHK(hipSetDevice(ix->shards[s]->device));
        HIPCHK(hipMemcpyAsync(c->d_qsw + (size_t)j * ix->qsw_bytes, c->h_qsw + (size_t)j * ix->qsw_bytes, ix->qsw_bytes,
                              hipMemcpyHostToDevice, c->work));
    }
    return SZG_OK;
}

// wait for the ticket's device work; a shared sweep whose candidate buffer overflowed (threshold from the prefix too
// loose: duplicates, sorted corpora) is redone through the score matrix
int TopkCall::wait_shards(Ticket &t)
{
    int rc = SZG_OK;
    for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
        Shard *sh = ix->shards[s];
        if (sh->n_rows == 0) continue;
        hipError_t e = hipSetDevice(sh->device);
        if (e == hipSuccess) e = hipStreamSynchronize(t.ctx[s]->work);
        if (e == hipSuccess && (t.ctx[s]->sent_own_stream || t.ctx[s]->early_n > 0)) e = hipStreamSynchronize(t.ctx[s]->stream);
        if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipStreamSynchronize", e);
        if (rc == SZG_OK) rc = finish_timing(ix, t.ctx[s]);
        Ctx *c = t.ctx[s];
        if (rc != SZG_OK || !c->mq_fused_used) continue;
        bool overflow = false;
        for (int j = 0; j < t.nq; j++) overflow |= c->h_cand_count[j * szg::kCandCountStride] > c->mq_cand_cap;
        c->mq_fused_used = false;
        if (!overflow) continue;
        {
            std::lock_guard<std::mutex> lk(ix->stats_mu);
            ix->stats.mq_launches -= (uint64_t)((t.nq + 16 * c->mq_nb - 1) / (16 * c->mq_nb));  // counted again by the rerun
            ix->stats.mq_queries -= (uint64_t)t.nq;
            ix->stats.mq_bf16_sweeps -= (c->mq_stage2 || c->mq_bf16_used) ? 1 : 0;
            ix->stats.mq_fallbacks += 1;
        }
        rc = enqueue_topk_mq(ix, sh, c, t.kp, t.kp_wide, t.nq, c->mq_nb, c->mq_has_allow, true);
        if (rc == SZG_OK) {
            e = hipStreamSynchronize(c->work);
            if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipStreamSynchronize", e);
        }
        if (rc == SZG_OK) rc = finish_timing(ix, c);
    }
    return rc;
}

// every query's candidates, the lists' lower bound and the first-k NaN flag -- taken before anything else, since the
// escalation and replay paths reuse the contexts' output buffers
void TopkCall::gather(Ticket &t, std::vector<std::vector<Cand>> *all, std::vector<double> *thr_min,
                      std::vector<uint8_t> *nan_first, int j0, int j1)
{
    for (int j = j0; j < j1; j++) {
        for (size_t s = 0; s < n_sh; s++) {
            Shard *sh = ix->shards[s];
            if (sh->n_rows == 0) continue;
            double lb;
            gather_topk(ix, sh, t.ctx[s], t.meta[j], j, &(*all)[j], &lb);
            (*thr_min)[j] = std::min((*thr_min)[j], lb);
            const Ctx *c = t.ctx[s];
            for (int i = 0; i < c->sent_n; i++) {
                const szg::RerankOut &r = c->sent_in_out ? c->h_out[(size_t)j * c->out_stride + c->kp_used + i]
                                                         : c->h_sent_out[(size_t)j * c->sent_n + i];
                if (r.row != 0xFFFFFFFFu && std::isnan(r.dist)) (*nan_first)[j] = 1;
            }
        }
    }
}

// One query of a finished batch: consider() replayed over its candidates, certification against the rows the lists
// left out, escalation (a collect sweep) when that fails, the exact replay when the reference's answer depends on its
// heap history.  *t_dev accumulates the time spent waiting on device passes.
// defer (non-null): no device pass may be started now (the call's last sweeps are still running and own the
// contexts' buffers) -- a query that needs one is reported back and settled again once everything has been gathered
int TopkCall::settle(Ticket &t, int j, std::vector<Cand> &cands, double thr_min, bool nan_first, double *t_dev,
                     std::vector<HeapItem> *res, bool *defer)
{
    const uint64_t *allow = mask_of(t.first + j);
    int rc = SZG_OK;
    // A NaN distance outside the query's first k eligible rows never enters the reference's heap
    // (`distance < worst` is false, collection.go:608-619); rows with an Inf / NaN element are forced
    // into the lists by the kernels (their float32 norm is not finite) and leave here.  A NaN among
    // the first k rows is the sentinels' business (nan_first: exact replay).
    auto drop_nan = [&](std::vector<Cand> &v) {
        if (nan_first) return;
        v.erase(std::remove_if(v.begin(), v.end(), [](const Cand &c) { return std::isnan(c.dist); }), v.end());
    };
    drop_nan(cands);
    replay_topk(cands, k, res);
    // certification: every row outside the lists has a real-number key >= thr_min (the lists' own lower bound),
    // so the result is final once the upper bound of its worst key stays below that
    bool certified = true;
    double kmax = -INFINITY;  // upper bound of the real-number key of the worst result
    const bool zero_query = ix->metric == SZG_COSINE && t.meta[j].m1 == 0;  // all distances 1.0
    if (thr_min < INFINITY && !zero_query) {
        std::vector<std::pair<uint64_t, double>> by_row;  // cands are sorted by row now
        by_row.reserve(cands.size());
        for (const Cand &c : cands) by_row.emplace_back(c.row, c.ub);
        for (const HeapItem &h : *res) {
            auto it = std::lower_bound(by_row.begin(), by_row.end(), std::make_pair(h.row, (double)-INFINITY));
            kmax = std::max(kmax, it->second);
        }
        certified = (int)res->size() == k && kmax < thr_min;
    }
    if (ix->force_escalate && thr_min < INFINITY) certified = false;
    if (nan_first && ix->tie_mode == 0) certified = true;  // answered by the replay below
    if (!certified) {
        if (defer) {
            *defer = true;
            return SZG_OK;
        }
        {
            std::lock_guard<std::mutex> lk(ix->stats_mu);
            ix->stats.escalations++;
        }
        // kmax bounds the worst result's real-number key; the collect sweep (always the single-query kernel) adds
        // its own error on the rows it tests
        double thr = INFINITY;
        cands.clear();
        const double td = now_us();
        if (t.lazy_single) {  // the escalation sweep is the single-query kernel: it wants the query in ITS form
            rc = stage_single_form(t, j);
            if (rc) return rc;
        }
        if ((int)res->size() == k && std::isfinite(kmax) && !zero_query) {
            QMeta single = t.meta[j];  // (now with the single-query path's quantization step)
            single.mq = false;
            single.mq_int = false;
            single.mq_bf16 = false;
            const double e2 = key_eps(ix, kmax, single);
            thr = kmax + 1.05 * e2 + 0.05 * std::fabs(kmax) * 0x1p-20;
        }
        const float thr_f = !(thr < 3.0e38) ? 3.0e38f : std::nextafter((float)thr, INFINITY);
        for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
            Shard *sh = ix->shards[s];
            if (sh->n_rows == 0) continue;
            rc = run_collect(ix, sh, t.ctx[s], j, thr_f, t.any_mask, &cands);
        }
        *t_dev += now_us() - td;
        if (rc) return rc;
        drop_nan(cands);
        replay_topk(cands, k, res);
    }
    if (ix->tie_mode == 0) {
        std::vector<double> d(cands.size());
        for (size_t i = 0; i < cands.size(); i++) d[i] = cands[i].dist;
        // (a zero cosine query is at distance exactly 1.0 from every row, collection.go:828-830: one big tie, whether or
        // not the candidate list is long enough to show two of its members)
        if (nan_first || zero_query || history_dependent(d.data(), d.size(), k)) {
            if (defer) {
                *defer = true;
                return SZG_OK;
            }
            {
                std::lock_guard<std::mutex> lk(ix->stats_mu);
                ix->stats.full_replays++;
            }
            const double td = now_us();
            rc = run_full_replay(ix, t.ctx, j, allow, k, res);
            *t_dev += now_us() - td;
        }
    }
    return rc;
}

// result assembly for one finished batch
int TopkCall::finish(Ticket &t)
{
    if (t.failed) {  // enqueueing failed part-way: drain and release; the enqueue error is already the call's return code
        for (size_t s = 0; s < n_sh; s++) {
            if (!t.ctx[s]) continue;
            (void)hipSetDevice(ix->shards[s]->device);
            (void)hipStreamSynchronize(t.ctx[s]->work);
            (void)hipStreamSynchronize(t.ctx[s]->stream);
            t.ctx[s]->mq_fused_used = false;
        }
        release(t);
        return SZG_OK;
    }
    double t_dev = 0;  // time spent waiting on escalation / replay passes (device work)
    double t_early = 0;  // host time of the early phase
    std::vector<std::vector<Cand>> all(t.nq);
    std::vector<double> thr_min(t.nq, INFINITY);
    std::vector<uint8_t> nan_first(t.nq, 0);  // a NaN distance among the query's first k eligible rows
    std::vector<uint8_t> done(t.nq, 0);
    std::vector<HeapItem> res;
    auto emit = [&](int j) {
        const int qi = t.first + j;
        for (int i = 0; i < k; i++) {
            const bool have = i < (int)res.size();
            out_rows[(size_t)qi * k + i] = have ? res[i].row + ix->row_base : UINT64_MAX;
            out_dist[(size_t)qi * k + i] = have ? res[i].priority : 0.0;
        }
        if (out_count) out_count[qi] = (int32_t)res.size();
    };
    int rc = SZG_OK;
    // A short call's early part (Ctx::early_n): those queries' lists and the sentinel rows' distances are complete once
    // the contexts' own streams are; they are assembled here while the scan streams run the call's last sweeps.  A
    // query that needs a device pass of its own (escalation, exact replay) waits for the second phase.
    int early = 0;
    for (size_t s = 0; s < n_sh; s++)
        if (t.ctx[s] && t.ctx[s]->early_n > 0) early = t.ctx[s]->early_n;
    if (early > 0 && early < t.nq && !replay_all) {
        for (size_t s = 0; s < n_sh && rc == SZG_OK; s++) {
            if (!t.ctx[s]) continue;
            hipError_t e = hipSetDevice(ix->shards[s]->device);
            if (e == hipSuccess) e = hipStreamSynchronize(t.ctx[s]->stream);
            if (e != hipSuccess) rc = fail(SZG_E_DEVICE, "hipStreamSynchronize", e);
        }
        const double te0 = now_us();
        if (rc == SZG_OK) gather(t, &all, &thr_min, &nan_first, 0, early);
        for (int j = 0; j < early && rc == SZG_OK; j++) {
            bool deferred = false;
            rc = settle(t, j, all[j], thr_min[j], nan_first[j] != 0, &t_dev, &res, &deferred);
            if (rc == SZG_OK && !deferred) {
                emit(j);
                done[j] = 1;
            }
        }
        t_early = now_us() - te0;
    } else {
        early = 0;
    }
    if (rc == SZG_OK) rc = wait_shards(t);
    else (void)wait_shards(t);
    const double t_fin0 = now_us() - t_early;
    if (rc == SZG_OK && !replay_all) gather(t, &all, &thr_min, &nan_first, early, t.nq);
    for (int j = 0; j < t.nq && rc == SZG_OK; j++) {
        if (done[j]) continue;
        const int qi = t.first + j;
        if (replay_all) {
            const double td = now_us();
            rc = run_full_replay(ix, t.ctx, j, mask_of(qi), k, &res);
            t_dev += now_us() - td;
            if (rc == SZG_OK) {
                std::lock_guard<std::mutex> lk(ix->stats_mu);
                ix->stats.full_replays++;
            }
        } else {
            rc = settle(t, j, all[j], thr_min[j], nan_first[j] != 0, &t_dev, &res);
        }
        if (rc) break;
        emit(j);
    }
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->stats.host_finish_us += now_us() - t_fin0 - t_dev;
        if (rc == SZG_OK) ix->stats.queries += t.nq;
    }
    release(t);
    return rc;
}

int TopkCall::run()
{
    n_sh = ix->shards.size();
    uint64_t total_rows = 0;
    for (Shard *s : ix->shards) total_rows += s->n_rows;
    allow_stride = (total_rows + 63) / 64;
    kp = k + std::max(ix->slack_min, k / 2);
    // The reference bounds K by nothing (collection.go:606-619).  The fused selection keeps kp candidates per wave in
    // LDS; beyond that (kp > 4096 or 64 KiB of lists) every query of the call takes the exact replay: float64
    // distances of all rows on the device, consider() over them on the host.
    for (Shard *s : ix->shards) {
        if (s->n_rows == 0 || replay_all) continue;
        const LaunchGeom g = scan_geometry(ix, s, kp);
        if (szg::scan_lds_bytes(ix->bits, ix->map, kp, g.block) > 64u * 1024u || kp > 4096) replay_all = true;
    }
    if (replay_all) kp = 1;  // the batches only stage their queries

    std::deque<Ticket> inflight;
    int rc = SZG_OK;
    const int B1 = std::max(1, std::min(ix->query_batch, kMaxBatch));

    // A call of several shared-sweep batches is bound by its host work (3-4 us per query against 2.7-3.7 us of GPU
    // time for the int8 and 16-bit sweeps): a second thread takes the finished batches -- waits, candidate
    // assembly, certification, output -- while this one prepares and enqueues.  The hand-over is the ticket queue;
    // contexts are the flow control (acquire blocks until the finisher has released one).
    struct Finisher {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Ticket> q;
        bool done = false;
        int rc = SZG_OK;
        std::string err;
        std::atomic<bool> failed{false};
        std::thread th;
        void stop()
        {
            if (!th.joinable()) return;
            {
                std::lock_guard<std::mutex> lk(mu);
                done = true;
            }
            cv.notify_one();
            th.join();
        }
        ~Finisher() { stop(); }  // (an exception unwinding the call: the queued tickets are still finished first)
    } fin;
    bool threaded = false;
    {
        const int nb0 = replay_all ? 0 : mq_blocks(ix, n_queries);
        threaded = ix->finish_thread && nb0 > 0 && n_queries > 2 * 16 * nb0 * (mq_uses_i8(ix, false, n_queries) ? 2 : 1);
    }
    if (threaded) {
        try {
            fin.th = std::thread([this, &fin] {
                for (;;) {
                    std::unique_lock<std::mutex> lk(fin.mu);
                    fin.cv.wait(lk, [&] { return !fin.q.empty() || fin.done; });
                    if (fin.q.empty()) return;
                    Ticket t(std::move(fin.q.front()));
                    fin.q.pop_front();
                    lk.unlock();
                    int r;
                    try {
                        r = finish(t);
                    } catch (const std::bad_alloc &) {
                        r = fail(SZG_E_NOMEM, "out of host memory");
                    } catch (...) {
                        r = fail(SZG_E_DEVICE, "unexpected exception");
                    }
                    if (r != SZG_OK && !fin.failed.load()) {  // (every ticket is still finished: its contexts go back)
                        fin.rc = r;
                        fin.err = szg_last_error();
                        fin.failed.store(true);
                    }
                }
            });
        } catch (...) {
            threaded = false;  // no thread to be had: this one does both
        }
    }
    auto hand_over = [&](Ticket &&t) {
        {
            std::lock_guard<std::mutex> lk(fin.mu);
            fin.q.push_back(std::move(t));
        }
        fin.cv.notify_one();
    };
    for (int q0 = 0; q0 < n_queries && rc == SZG_OK && !fin.failed.load();) {
        Ticket t;
        t.owner = ix;
        t.first = q0;
        const int left = n_queries - q0;
        const int nb = replay_all ? 0 : mq_blocks(ix, left);  // > 0: the batch shares one sweep
        // (int8 sweeps: two groups of 48 per launch when that many queries are waiting and both images fit LDS)
        const int groups = nb == 3 && mq_uses_i8(ix, false, left) && ix->mq_i8_groups > 1 && left > 48 &&
                                   szg::mq_i8_lds_bytes(ix->bits, ix->map.r16, 3, 2) <= 160u * 1024u
                               ? 2 : 1;
        t.nq = nb ? std::min(left, 16 * nb * groups) : std::min(B1, left);
        // A short call with one sweep per query is ONE batch on the scan stream (launches of <= 16 sweeps back to
        // back, uploads ahead of them, no event on the critical path).  From 8 queries on, the merges, re-rank,
        // copy-back and host assembly of all but its last few queries run on the context's own stream and the
        // calling thread WHILE the last sweeps run; only those last queries' tail is left after the final sweep.
        single_batch = !nb && q0 == 0 && ix->short_call > 0 && n_queries <= std::min(ix->short_call, kMaxBatch);
        early_n = 0;
        if (single_batch) {
            t.nq = n_queries;
            static const bool no_early = getenv("SZG_NO_EARLY_TAIL") != nullptr;  // (A/B hook of scripts/dev_short.py)
            if (n_queries >= 8 && ix->serialize_scans && !no_early) early_n = n_queries - std::min(kShortCallLast, n_queries / 2);
        }
        // one sweep per query: the call's FIRST batch is small, so that the card starts sweeping after a few
        // microseconds of preparation instead of a whole batch's (the next batch is prepared while it sweeps)
        // ... and its LAST one too: what is left to do once the last sweep has ended is that batch's merges,
        // re-rank, copy-back and result assembly
        if (!nb && !single_batch && ix->first_batch > 0 && left > ix->first_batch) {
            if (q0 == 0) t.nq = std::min(t.nq, ix->first_batch);
            else if (left <= B1 + ix->first_batch) t.nq = left - ix->first_batch;
        }
        const bool bf16_sweep = nb > 0 && mq_uses_bf16(ix, false, t.nq);
        t.kp = kp;
        // lists of bfloat16-sweep keys (matrix form): the error band holds more rows than the float32 one's, keep
        // enough candidates for the k-th result to clear it
        t.kp_wide = bf16_sweep ? std::min(4096, std::max(kp, k + std::max(ix->mq_bf16_slack, k / 2))) : kp;
        t.ctx.assign(n_sh, nullptr);
        t.meta.assign(t.nq, QMeta{});
        if (threaded) {
            (void)acquire(t, true);  // blocks until the finisher (or another caller) gives a context back
            rc = stage(t, nb, bf16_sweep);
            t.failed = rc != SZG_OK;
            q0 += t.nq;
            hand_over(std::move(t));
            continue;
        }
        if (!acquire(t, inflight.empty())) {  // no free context: finish the oldest batch first
            rc = finish(inflight.front());
            inflight.pop_front();
            continue;
        }
        rc = stage(t, nb, bf16_sweep);
        t.failed = rc != SZG_OK;  // nothing to gather: finish() only drains and releases
        inflight.push_back(std::move(t));
        q0 += inflight.back().nq;
    }
    if (threaded) {
        fin.stop();
        if (rc == SZG_OK && fin.failed.load()) rc = fail(fin.rc, fin.err.c_str());  // (this thread's last-error slot)
        return rc;
    }
    while (!inflight.empty()) {
        const int r2 = finish(inflight.front());
        if (rc == SZG_OK) rc = r2;
        inflight.pop_front();
    }
    return rc;
}

int search_topk_impl(szg_index *ix, const double *queries, int n_queries, int k, const uint64_t *allow_bits,
                     uint64_t *out_rows, double *out_dist, int32_t *out_count, const uint64_t *const *allow_ptrs)
{
    TopkCall call{ix, queries, n_queries, k, allow_bits, allow_ptrs, out_rows, out_dist, out_count};
    return call.run();
}

}  // namespace szgi
