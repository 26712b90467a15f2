// scan_mq.cpp -- shared sweeps: one pass of the corpus for a batch of queries, dot products on the matrix cores.
#include "scan_internal.h"

namespace szgi {

// ---- the batch as the sweeps stage it in LDS (one image per sweep kind), built in the context's pinned buffer ------

// bfloat16 sweep: [32-element step][query block][lane = k-group*16 + query][8 bf16 = elements 8*k-group + 0..7 of the
// step]; cosine: q/|q|.
void build_image_bf16(const szg_index *ix, Ctx *c, int nq, int nb)
{
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
}

// int8 sweep: per group of 16 * nb queries, [64-byte step][digit plane h..l][T halves][query block][lane = chunk*16 +
// query][16 bytes] of the int8 digits of Q = round(v / mq_qscale), then the table [qscale | qconst | qnorm2][48].
// 8-bit rows (T = 1): byte i of a lane's word belongs to element 16*piece + i; the row operand is v' = v - 128 and
// n = 2v' + 1, so sum Q n = 2 sum Q v' + sum Q.  4-bit rows (T = 2: even | odd elements): byte bi belongs to element
// 32*piece + 2*bi (+1 for the odd half); the operand is the nibble x and n = 2x - 15.
void build_image_i8(const szg_index *ix, Ctx *c, int nq, int nb, size_t group_stride)
{
    const int r16 = ix->map.r16;
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
        float *tab = reinterpret_cast<float *>(im8 + szg::mq_i8_image_bytes(ix->bits, r16, nb));
        tab[ql] = (float)c->meta[q].mq_qscale;
        tab[48 + ql] = (float)((ix->bits == 4 ? -15.0 : 1.0) * c->meta[q].mq_qconst);
        tab[96 + ql] = (float)c->meta[q].qnorm2;
    }
}

// float32 sweep: [piece j][query block][group of 4 elements][query 16][4 floats].  Cosine: the normalised queries
// (q / |q|, so the key is -cos; quantized rows decode to n = maxInt * d and the common factor cancels).  Euclid:
// maxInt * q for quantized rows (key = |n - maxInt q|^2 = maxInt^2 |d - q|^2, the single-query path's unit).
void build_image_f32(const szg_index *ix, Ctx *c, int nq, int nb)
{
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

// top-k pass for the nq staged queries through ONE shared sweep:
// score matrix -> per-query selection -> merges -> rerank -> D2H (async)
// kp_wide: the list length when the lists hold bfloat16-sweep keys themselves (matrix form: small
// shards, overflow reruns), whose error band needs more candidates than kp
int enqueue_topk_mq(szg_index *ix, Shard *sh, Ctx *c, int kp, int kp_wide, int nq, int nb, bool has_allow,
                    bool force_matrix)
{
    HIPCHK(hipSetDevice(sh->device));
    const int r16 = ix->map.r16;
    const bool i8 = mq_uses_i8(ix);
    const bool bf16 = mq_uses_bf16(ix);
    // int8 sweeps: up to two groups of 16 * nb queries per launch (the kernel walks their passes back to back)
    const int groups = i8 ? (nq + 16 * nb - 1) / (16 * nb) : 1;
    // (the kernel indexes thresholds, keys and candidates of group g by 48 g + q: a second group needs full groups)
    if (groups > 2 || (groups == 2 && nb != 3)) return fail(SZG_E_INVALID, "int8 shared sweep: two groups need 48 queries each");
    const size_t group_stride =
        i8 ? ((szg::mq_i8_image_bytes(ix->bits, r16, nb) + 3 * 48 * sizeof(float) + 255) & ~(size_t)255) : 0;
    const size_t img = bf16 ? szg::mq_bf16_image_bytes(r16, nb)
                            : i8 ? group_stride * groups : szg::mq_lds_bytes(ix->bits, r16, nb);
    int rc = ensure_host(&c->h_mq, &c->h_mq_cap, img);
    if (rc) return rc;
    rc = ensure_dev(&c->d_mq, &c->d_mq_cap, img);
    if (rc) return rc;
    memset(c->h_mq, 0, img);
    if (bf16) build_image_bf16(ix, c, nq, nb);
    else if (i8) build_image_i8(ix, c, nq, nb, group_stride);
    else build_image_f32(ix, c, nq, nb);
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

}  // namespace szgi
