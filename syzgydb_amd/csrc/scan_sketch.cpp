// scan_sketch.cpp -- 8-bit sketch pre-pass for float32 collections (option "sketch", off by default).
#include "scan_internal.h"

namespace szgi {

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
        struct Scratch {  // freed on every exit path
            unsigned long long *&a;
            uint32_t *&b, *&c;
            ~Scratch()
            {
                (void)hipFree(a);
                (void)hipFree(b);
                (void)hipFree(c);
            }
        } scratch{d_ang, d_exc, d_list};
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
        if (e != hipSuccess) return fail(SZG_E_DEVICE, "sketch build", e);
        double ang;
        memcpy(&ang, &ang_bits, sizeof(ang));
        ix->sk_max_ang = std::max(ix->sk_max_ang, ang);
        if (exc[0] > exc_cap || ix->sk_exc.size() + exc[0] > exc_cap) {
            // a collection of zero / non-finite rows: nothing to gain -- give the second index (+25 % memory) back;
            // the next load / synth gets a fresh try
            ix->sk_disabled = true;
            szg_index_destroy(ix->sketch);
            ix->sketch = nullptr;
            ix->sk_need_full = true;
            ix->sk_dirty_rows.clear();
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
                    uint64_t *out_rows, double *out_dist, int32_t *out_count, const uint64_t *const *allow_ptrs)
{
    // (a batch that shares one sweep on the matrix cores is cheaper per query than any pre-pass)
    const bool shared = ix->multi_query && n_queries >= ix->mq_min;
    if (!shared && sketch_applies(ix, k))
        return search_topk_sketch(ix, queries, n_queries, k, allow_bits, out_rows, out_dist, out_count, allow_ptrs);
    return search_topk_impl(ix, queries, n_queries, k, allow_bits, out_rows, out_dist, out_count, allow_ptrs);
}

}  // namespace szgi
