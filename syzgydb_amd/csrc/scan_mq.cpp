// scan_mq.cpp -- shared sweeps: one pass of the corpus for a batch of queries, dot products on the matrix cores.
#include "scan_internal.h"

namespace szgi {

// ---- the batch as the sweeps stage it in LDS (one image per sweep kind), built in the context's pinned buffer ------

// bfloat16 sweep: [32-element step][query block][lane = k-group*16 + query][8 bf16 = elements 8*k-group + 0..7 of the
// step]; cosine: q/|q|.
void build_image_bf16(const szg_index *ix, Ctx *c, int nq, int nb)
{
    uint16_t *im = reinterpret_cast<uint16_t *>(c->h_mq);
    const int dim = ix->dim;
    for (int q = 0; q < nq; q++) {
        const double *src = c->h_q64 + (size_t)q * dim;
        const double m1 = c->meta[q].m1;
        double scale = m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0;
        if (ix->metric != SZG_COSINE) scale = ix->bits == 16 ? 65535.0 : (ix->bits == 8 ? 255.0 : 1.0);  // maxInt * q against rows decoded to n
        const int b = q / 16, qi = q % 16;
        // (8-bit rows: n = 2v' + 1, v' = v - 128 -- the sweep multiplies v' and adds sum g, over the ROUNDED image values)
        const bool want_sum = ix->bits == 8;
        double gs[4] = {0.0, 0.0, 0.0, 0.0};
        // runs of 8 consecutive elements are contiguous in the image (one lane's 16 bytes)
        for (int e0 = 0; e0 < dim; e0 += 8) {
            const int S = e0 >> 5, kg = (e0 & 31) >> 3;
            uint16_t *dst = im + ((((size_t)S * nb + b) * 64) + kg * 16 + qi) * 8;
            const int cnt = std::min(8, dim - e0);
            for (int i = 0; i < cnt; i++) dst[i] = bf16_rne((float)(src[e0 + i] * scale));
            if (want_sum) {
                for (int i = 0; i < cnt; i++) {
                    const uint32_t u = (uint32_t)dst[i] << 16;
                    float f;
                    memcpy(&f, &u, 4);
                    gs[i & 3] += (double)f;
                }
            }
        }
        const double gsum = (gs[0] + gs[1]) + (gs[2] + gs[3]);
        c->mq_qsum[q] = (float)gsum;
    }
}

// int8 sweep: per group of 16 * nb queries, [64-byte step][digit plane h..l][T halves][query block][lane = chunk*16 +
// query][16 bytes] of the int8 digits of Q = round(v / mq_qscale), then the table [qscale | qconst | qnorm2][48].
// 8-bit rows (T = 1): byte i of a lane's word belongs to element 16*piece + i; the row operand is v' = v - 128 and
// n = 2v' + 1, so sum Q n = 2 sum Q v' + sum Q.  4-bit rows (T = 2: even | odd elements): byte bi belongs to element
// 32*piece + 2*bi (+1 for the odd half); the operand is the nibble x and n = 2x - 15.
void build_image_i8(const szg_index *ix, Ctx *c, int nq, int nb, size_t group_stride)
{
    const int r16 = ix->map.r16, dim = ix->dim;
    constexpr int NP = szg::kMqPlanes;
    const int T = ix->bits == 4 ? 2 : 1;
    const int epp = ix->bits == 4 ? 32 : 16;  // elements per 16-byte piece
    const size_t plane = (size_t)T * nb * 64 * 16;  // bytes between digit planes of a step
    const size_t half = (size_t)nb * 64 * 16;       // 4-bit rows: bytes from the even elements' operand to the odd elements'
    for (int q = 0; q < nq; q++) {
        const int32_t *Qv = c->h_mqQ + (size_t)q * dim;
        const int ql = q % (16 * nb);  // position inside its group
        uint8_t *im8 = c->h_mq + (size_t)(q / (16 * nb)) * group_stride;
        const int b = ql / 16, qi = ql % 16;
        // one 16-byte piece of the row = one lane's word per (plane, half): its bytes are contiguous in the image
        for (int j = 0, e0 = 0; e0 < dim; j++, e0 += epp) {
            const int cnt = std::min(epp, dim - e0);
            int8_t dig[NP][32];
            for (int i = 0; i < cnt; i++) {   // low digits first, balanced in [-64, 63]; plane 0 = the top digit
                int Q = Qv[e0 + i];
                for (int p = NP - 1; p > 0; p--) {
                    const int d = ((Q + 64) & 127) - 64;
                    Q = (Q - d) >> 7;
                    dig[p][i] = (int8_t)d;
                }
                dig[0][i] = (int8_t)Q;
            }
            for (int p = 0; p < NP; p++)
                for (int i = cnt; i < epp; i++) dig[p][i] = 0;
            const int st = j >> 2, ch = j & 3;
            uint8_t *dst = im8 + ((((size_t)st * NP * T) * nb + b) * 64 + ch * 16 + qi) * 16;
            for (int p = 0; p < NP; p++) {
                uint8_t *d = dst + (size_t)p * plane;
                if (T == 1) {
                    memcpy(d, dig[p], 16);
                } else {
                    for (int k = 0; k < 16; k++) {
                        d[k] = (uint8_t)dig[p][2 * k];
                        d[half + k] = (uint8_t)dig[p][2 * k + 1];
                    }
                }
            }
        }
        float *tab = reinterpret_cast<float *>(im8 + szg::mq_i8_image_bytes(ix->bits, r16, nb));
        tab[ql] = (float)c->meta[q].mq_qscale;
        tab[48 + ql] = (float)((ix->bits == 4 ? -15.0 : 1.0) * c->meta[q].mq_qconst);
        tab[96 + ql] = (float)c->meta[q].qnorm2;
    }
}

// Resident row norms.  The 16-bit and the int8 shared sweeps are bound by their vector-instruction issue, and a good
// part of it summed the row's squares -- 8 of the 30 vector instructions of a 16-bit K-step, 12 of the 24 that decode
// a 64-byte step of 4-bit rows -- a number that only changes when the row does.  It lives beside the rows (4 bytes per
// row: 0.26 % more to read for 16-bit rows of 768 dims, 1 % for 4-bit), computed on the device for the rows added since
// the last shared sweep; an overwritten row's is refreshed at once (scan_handle.cpp).
static bool row_norms_apply(const szg_index *ix)
{
    static const bool off = getenv("SZG_NO_ROW_NORMS") != nullptr;  // (A/B hook)
    if (off) return false;
    return (ix->bits == 16 && !ix->layout.tiled) || ix->bits == 8 || ix->bits == 4;
}

int ensure_row_norms(szg_index *ix, Shard *sh)
{
    if (!row_norms_apply(ix) || sh->n_rows == 0) return SZG_OK;
    std::lock_guard<std::mutex> lk(sh->norm_mu);
    if (sh->norm_valid >= sh->n_rows && sh->row_norm) return SZG_OK;
    HIPCHK(hipSetDevice(sh->device));
    if (sh->norm_cap < sh->n_rows) {
        const uint64_t cap = std::max<uint64_t>(sh->cap_rows, sh->n_rows);
        float *nn = nullptr;
        hipError_t e = hipMalloc((void **)&nn, cap * sizeof(float));
        if (e != hipSuccess) return fail(SZG_E_NOMEM, "hipMalloc(row norms)", e);
        HIPCHK(hipStreamSynchronize(sh->scan_stream));  // (sweeps in flight read the old array)
        if (sh->row_norm && sh->norm_valid)
            HIPCHK(hipMemcpy(nn, sh->row_norm, sh->norm_valid * sizeof(float), hipMemcpyDeviceToDevice));
        (void)hipFree(sh->row_norm);
        sh->row_norm = nn;
        sh->norm_cap = cap;
    }
    HIPCHK(szg::launch_row_norms(ix->bits, sh->rows, ix->layout, ix->dim, (float)ix->norm_bias, sh->norm_valid,
                                 sh->n_rows - sh->norm_valid, sh->row_norm, sh->scan_stream));
    HIPCHK(hipStreamSynchronize(sh->scan_stream));  // (the batch's prefix pass may run on another stream)
    sh->norm_valid = sh->n_rows;
    return SZG_OK;
}

// ---- one batch through ONE shared sweep --------------------------------------------------------------------------

namespace {

// Fused selection: sweep a prefix of the rows into a small score matrix, take each query's kp-th best key there as
// its threshold, then sweep everything and collect the (query, row) pairs at or below their threshold -- about
// `hits` per query -- instead of writing and re-reading n_rows x batch keys.  Every row outside a query's buffer
// has a key above the threshold, which is >= the kp-th kept key: certification is unchanged.
struct MqPlan {
    bool i8 = false, bf16 = false;
    int groups = 1;            // int8 sweeps: query groups of 48 one launch walks
    size_t group_stride = 0;   // bytes between the groups' images
    size_t img = 0;            // bytes of the LDS image(s)
    uint64_t prefix = 0;       // rows of the threshold pass
    bool fused = false;        // threshold-collect selection (else: score matrix of every row)
    bool stage2 = false;       // bfloat16 sweep whose collected candidates get float32 keys before the selection
    bool refine = false;       // the batch's tail is one cand_refine launch + one rerank (sentinels included)
    uint32_t cand_cap = 0;     // entries per query's candidate buffer
    size_t key_stride = 0;     // floats per query in the score matrix
    int kp = 0;                // list length (kp_wide when the lists hold bfloat16 keys themselves)
};

MqPlan mq_plan(const szg_index *ix, const Shard *sh, int kp, int kp_wide, int nq, int nb, bool force_matrix)
{
    MqPlan p;
    const int r16 = ix->map.r16;
    p.i8 = mq_uses_i8(ix, false, nq);
    p.bf16 = mq_uses_bf16(ix, false, nq);
    p.groups = p.i8 ? (nq + 16 * nb - 1) / (16 * nb) : 1;
    p.group_stride = p.i8 ? ((szg::mq_i8_image_bytes(ix->bits, r16, nb) + 3 * 48 * sizeof(float) + 255) & ~(size_t)255) : 0;
    p.img = p.bf16 ? szg::mq_bf16_image_bytes(ix->bits, r16, nb) : p.group_stride * p.groups;
    const uint64_t hits = std::max<uint64_t>((uint64_t)ix->mq_hits, 16ull * kp);
    p.prefix = ((sh->n_rows * (uint64_t)kp + hits - 1) / hits + 15) & ~15ull;
    p.prefix = std::max<uint64_t>(p.prefix, 16ull * kp);
    p.fused = !ix->force_matrix && !force_matrix && p.prefix * 4 <= sh->n_rows;
    p.stage2 = p.bf16 && p.fused;
    p.kp = p.bf16 && !p.fused ? std::max(kp, kp_wide) : kp;
    p.cand_cap = (uint32_t)(4 * hits);
    p.key_stride = p.fused ? (size_t)p.prefix : (((size_t)sh->n_rows + 3) & ~(size_t)3);
    p.refine = p.fused && !ix->force_no_refine && szg::cand_refine_applies(p.kp, p.cand_cap, ix->dim, p.stage2);
    return p;
}

int mq_buffers(szg_index *ix, Ctx *c, const MqPlan &p, int nq, int n_out)
{
    const int sb = 16;  // select blocks per query (score-matrix form)
    const size_t need = (size_t)nq * std::max<size_t>((size_t)sb * p.kp, (size_t)n_out);
    if (c->lists_cap < need) {
        if (c->d_lists_a) HIPCHK(hipFree(c->d_lists_a));
        if (c->d_lists_b) HIPCHK(hipFree(c->d_lists_b));
        c->d_lists_a = c->d_lists_b = nullptr;
        c->lists_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_lists_a, need * sizeof(uint64_t)));
        HIPCHK(hipMalloc((void **)&c->d_lists_b, need * sizeof(uint64_t)));
        c->lists_cap = need;
    }
    int rc = ensure_dev(&c->d_out, &c->d_out_cap, (size_t)nq * n_out);
    if (rc) return rc;
    rc = ensure_host(&c->h_out, &c->h_out_cap, (size_t)nq * n_out);
    if (rc) return rc;
    rc = ensure_dev(&c->d_keys, &c->keys_cap, p.key_stride * nq);
    if (rc) return rc;
    if (p.fused) {
        if (!c->d_thr) HIPCHK(hipMalloc((void **)&c->d_thr, 256 * sizeof(float)));  // thresholds | band edges
        if (!c->h_thr) HIPCHK(hipHostMalloc((void **)&c->h_thr, 256 * sizeof(float), hipHostMallocDefault));
        if (!c->d_cand_count) HIPCHK(hipMalloc((void **)&c->d_cand_count, 128 * szg::kCandCountStride * sizeof(uint32_t)));
        if (!c->h_cand_count)
            HIPCHK(hipHostMalloc((void **)&c->h_cand_count, 128 * szg::kCandCountStride * sizeof(uint32_t), hipHostMallocDefault));
        rc = ensure_dev(&c->d_cand, &c->cand_cap_total, (size_t)p.cand_cap * nq);
        if (rc) return rc;
    }
    if (p.stage2) {  // float32 re-score: 1/|q| (cosine) or 1 per query, then |g|^2 (the euclid band's width)
        if (!c->h_qscale) HIPCHK(hipHostMalloc((void **)&c->h_qscale, 256 * sizeof(double), hipHostMallocDefault));
        if (!c->d_qscale) HIPCHK(hipMalloc((void **)&c->d_qscale, 256 * sizeof(double)));
        for (int q = 0; q < nq; q++) {
            const double m1 = c->meta[q].m1;
            c->h_qscale[q] = ix->metric == SZG_COSINE ? (m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0)
                                                      : (ix->bits == 16 ? 65535.0 : (ix->bits == 8 ? 255.0 : 1.0));  // euclid: the prepared query, maxInt * q
            c->h_qscale[128 + q] = c->meta[q].qnorm2;
        }
        HIPCHK(hipMemcpyAsync(c->d_qscale, c->h_qscale, sizeof(double) * 256, hipMemcpyHostToDevice, c->stream));
    }
    return SZG_OK;
}

}  // namespace

bool mq_tail_takes_sentinels(const szg_index *ix, const Shard *sh, int kp, int kp_wide, int nq, int nb)
{
    return mq_plan(ix, sh, kp, kp_wide, nq, nb, false).refine;
}

// the batch's LDS image: built in the context's pinned buffer, uploaded on `st`
static int upload_mq_image(szg_index *ix, Ctx *c, int nq, int nb, bool bf16, bool i8, size_t img, size_t group_stride,
                           hipStream_t st)
{
    int rc = ensure_host(&c->h_mq, &c->h_mq_cap, img);
    if (rc) return rc;
    rc = ensure_dev(&c->d_mq, &c->d_mq_cap, img);
    if (rc) return rc;
    memset(c->h_mq, 0, img);
    if (bf16) build_image_bf16(ix, c, nq, nb);
    else if (i8) build_image_i8(ix, c, nq, nb, group_stride);
    else return fail(SZG_E_INVALID, "no shared sweep for this row width");
    HIPCHK(hipMemcpyAsync(c->d_mq, c->h_mq, img, hipMemcpyHostToDevice, st));
    return SZG_OK;
}

// the arguments every shared sweep of a batch has in common (image in c->d_mq, constants from the staged queries)
static szg::MqArgs mq_args_base(const szg_index *ix, const Shard *sh, const Ctx *c, int nq, int groups, size_t group_stride)
{
    szg::MqArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = sh->rows;
    a.n_rows = (uint32_t)sh->n_rows;
    a.pitch = ix->pitch;
    a.tiled = ix->layout.tiled;
    a.steps = ix->layout.steps;
    a.r16 = ix->map.r16;
    a.dim = ix->dim;
    a.queries = c->d_mq;
    a.n_queries = nq;
    a.n_groups = groups;
    a.shape_kernels = ix->shape_kernels;
    a.group_stride = (uint32_t)group_stride;
    a.metric = ix->metric;
    for (int q = 0; q < nq && q < szg::kMqMaxQueries; q++) {
        a.qnorm2[q] = (float)c->meta[q].qnorm2;
        a.qsum[q] = c->mq_qsum[q];
    }
    a.zero16 = sh->zero16;
    a.norm_bias = (float)ix->norm_bias;
    a.row_norm = row_norms_apply(ix) && sh->row_norm && sh->norm_valid >= sh->n_rows ? sh->row_norm : nullptr;
    return a;
}

// top-k pass for the nq staged queries through ONE shared sweep:
//   threshold pass (prefix) -> full sweep, collecting -> cand_refine (selection, float32 re-score of the bfloat16
//   band, sentinel rows appended) -> ONE rerank -> D2H          [fused selection, the default]
//   full sweep into a score matrix -> per-query selection -> merges -> rerank -> D2H   [small shards, overflow reruns]
// kp_wide: the list length when the lists hold bfloat16-sweep keys themselves (matrix form), whose error band needs
// more candidates than kp.
int enqueue_topk_mq(szg_index *ix, Shard *sh, Ctx *c, int kp, int kp_wide, int nq, int nb, bool has_allow,
                    bool force_matrix)
{
    HIPCHK(hipSetDevice(sh->device));
    int rc_norm = ensure_row_norms(ix, sh);
    if (rc_norm) return rc_norm;
    const MqPlan p = mq_plan(ix, sh, kp, kp_wide, nq, nb, force_matrix);
    // (the kernel indexes thresholds, keys and candidates of group g by 48 g + q: a second group needs full groups)
    if (p.groups > 2 || (p.groups == 2 && nb != 3)) return fail(SZG_E_INVALID, "int8 shared sweep: two groups need 48 queries each");
    kp = p.kp;
    int rc = upload_mq_image(ix, c, nq, nb, p.bf16, p.i8, p.img, p.group_stride, c->stream);
    if (rc) return rc;

    // sentinel rows staged by the caller ride in the batch's one rerank when the tail is the refine launch;
    // otherwise (score-matrix form, an overflow rerun) they get their own
    const int n_sent = c->sent_deferred ? c->sent_n : 0;
    const bool merge_sent = p.refine && n_sent > 0;
    const int n_out = kp + (merge_sent ? n_sent : 0);
    rc = mq_buffers(ix, c, p, nq, n_out);
    if (rc) return rc;
    c->mq_fused_used = p.fused;
    c->mq_cand_cap = p.cand_cap;
    c->mq_nb = nb;
    c->mq_has_allow = has_allow;
    c->kp_used = kp;
    c->out_stride = n_out;
    c->sent_in_out = merge_sent;
    c->mq_stage2 = p.stage2;
    c->mq_band_used = p.stage2 && p.refine;
    c->mq_bf16_used = p.bf16 && !p.stage2;

    szg::MqArgs a = mq_args_base(ix, sh, c, nq, p.groups, p.group_stride);
    a.keys = c->d_keys;
    a.key_stride = p.key_stride;
    const uint64_t *live = sh->has_dead ? sh->live_bits : nullptr;
    const uint64_t *allow = has_allow ? c->d_allow : nullptr;
    const uint32_t words = (uint32_t)shard_words(sh);
    auto launch_score = [&](const szg::MqArgs &x, hipStream_t s2) -> hipError_t {
        if (p.bf16) return szg::launch_mq_score_bf16(ix->bits, x, nb, sh->cu_count, s2);
        return szg::launch_mq_score_i8(ix->bits, x, nb, sh->cu_count, s2);
    };
    // The sweep wants every CU to itself (one big block and up to 150 KiB of LDS per CU), so the batch's kernels go
    // onto the shard's scan stream one after the other; uploads -- and for the HBM-bound sweeps (mq_overlap) the
    // threshold pass and the tail -- ride on the context's stream beside the neighbouring batches' sweeps.
    {
        std::lock_guard<std::mutex> lk(sh->chain_mu);
        hipStream_t st = ix->serialize_scans ? sh->scan_stream : c->stream;
        const bool overlap = (p.bf16 || p.i8) && ix->mq_overlap && st != c->stream;
        hipStream_t head = overlap ? c->stream : st;
        if (st != c->stream && !overlap) {
            HIPCHK(hipEventRecord(c->ev_up, c->stream));
            HIPCHK(hipStreamWaitEvent(st, c->ev_up, 0));
        }
        if (p.fused) {
            szg::MqArgs pa = a;  // the prefix, into the (small) score matrix
            pa.n_rows = (uint32_t)p.prefix;
            HIPCHK(launch_score(pa, head));
            // one block per query selects over the prefix's keys, publishes the query's threshold and zeroes its
            // hit counter
            HIPCHK(szg::launch_mq_select(c->d_keys, p.key_stride, (uint32_t)p.prefix, live, allow, words, kp, nq, 1,
                                         c->d_lists_a, head, c->d_thr, c->d_cand_count));
            a.collect = 1;
            a.thr = c->d_thr;
            a.cand_buf = c->d_cand;
            a.cand_count = c->d_cand_count;
            a.cand_cap = p.cand_cap;
            a.live_bits = live;
            a.allow_bits = allow;
            a.allow_stride = words;
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
            c->timed_n = p.groups;  // (passes: an int8 launch may walk two)
        }
        hipStream_t tail = st;
        if ((ix->mq_tail_overlap || overlap) && st != c->stream) {
            HIPCHK(hipEventRecord(c->ev_scan_done, st));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_scan_done, 0));
            tail = c->stream;
        }
        uint64_t *src = c->d_lists_a;
        if (p.refine) {
            const int mode = !p.stage2 ? 0 : (ix->metric == SZG_COSINE ? 1 : 2);
            HIPCHK(szg::launch_cand_refine(mode, sh->rows, ix->layout, ix->dim, c->d_q64, c->d_qscale,
                                           c->d_qscale ? c->d_qscale + 128 : nullptr, c->d_cand, c->d_cand_count,
                                           p.cand_cap, kp, nq, merge_sent ? c->d_sent : nullptr, merge_sent ? n_sent : 0,
                                           c->d_lists_a, c->d_thr + 128, ix->bits, tail));
        } else if (p.fused) {
            if (p.stage2)
                HIPCHK(szg::launch_cand_rescore(ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64, c->d_qscale, c->d_cand,
                                                c->d_cand_count, p.cand_cap, nq, ix->bits, tail));
            HIPCHK(szg::launch_cand_select(c->d_cand, c->d_cand_count, p.cand_cap, kp, nq, c->d_lists_a, tail));
        } else {
            // score matrix of every row -> per-query sorted lists of kp -> merged
            HIPCHK(szg::launch_mq_select(c->d_keys, p.key_stride, (uint32_t)sh->n_rows, live, allow, words, kp, nq, 16,
                                         c->d_lists_a, tail));
            int n_lists = 16;
            uint64_t *dst = c->d_lists_b;
            const int fan = szg::merge_fan(kp);
            while (n_lists > 1) {
                HIPCHK(szg::launch_merge(src, n_lists, kp, nq, dst, tail));
                n_lists = (n_lists + fan - 1) / fan;
                std::swap(src, dst);
            }
        }
        if (p.fused) {
            HIPCHK(hipMemcpyAsync(c->h_thr, c->d_thr, 256 * sizeof(float), hipMemcpyDeviceToHost, tail));
            HIPCHK(hipMemcpyAsync(c->h_cand_count, c->d_cand_count, 128 * szg::kCandCountStride * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, tail));
        }
        HIPCHK(szg::launch_rerank(ix->bits, ix->metric, sh->rows, ix->layout, ix->dim, c->d_q64, src, nullptr,
                                  (uint32_t)n_out, nq, c->d_out, tail));
        HIPCHK(hipMemcpyAsync(c->h_out, c->d_out, sizeof(szg::RerankOut) * n_out * nq, hipMemcpyDeviceToHost, tail));
        if (n_sent > 0 && !merge_sent) {  // deferred sentinels the tail could not take along
            rc = launch_sentinel_rerank(ix, sh, c, nq, tail);
            if (rc) return rc;
        }
        if (tail != c->stream) {
            HIPCHK(hipEventRecord(c->ev_scan_done, st));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_scan_done, 0));
        }
    }
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->stats.scan_launches += (uint64_t)p.groups;
        ix->stats.scan_bytes += (uint64_t)p.groups * sh->n_rows * (uint64_t)ix->row_bytes;  // ONE pass per group of the batch
        ix->stats.mq_launches += (uint64_t)p.groups;
        ix->stats.mq_queries += (uint64_t)nq;
        ix->stats.mq_bf16_sweeps += p.bf16 ? 1 : 0;
    }
    if (ix->timing >= 2) HIPCHK(hipEventRecord(c->ev_all1, c->stream));
    return SZG_OK;
}

// ---- a radius batch through ONE shared sweep ----------------------------------------------------------------------
//
// A radius search needs no threshold pass: the radius IS the threshold (radius_key_threshold, with the sweep's own
// error bound).  The collect form of the shared sweeps appends every (query, row) pair at or below its query's key
// threshold to the query's buffer; the caller re-ranks all of them in float64 and applies the reference's predicate.
// One pass of the corpus then serves up to 96 queries instead of one.
int enqueue_collect_mq(szg_index *ix, Shard *sh, Ctx *c, int nq, int nb, bool has_allow, const float *thr, size_t cap)
{
    HIPCHK(hipSetDevice(sh->device));
    int rc_norm = ensure_row_norms(ix, sh);
    if (rc_norm) return rc_norm;
    const int r16 = ix->map.r16;
    const bool i8 = mq_uses_i8(ix, true), bf16 = mq_uses_bf16(ix, true);
    const int groups = i8 ? (nq + 16 * nb - 1) / (16 * nb) : 1;
    if (nq > szg::kMqMaxQueries || groups > 2 || (groups == 2 && nb != 3))
        return fail(SZG_E_INVALID, "shared radius sweep: batch too large for the image");
    const size_t group_stride = i8 ? ((szg::mq_i8_image_bytes(ix->bits, r16, nb) + 3 * 48 * sizeof(float) + 255) & ~(size_t)255) : 0;
    if (!bf16 && !i8) return fail(SZG_E_INVALID, "no shared sweep for this row width");
    const size_t img = bf16 ? szg::mq_bf16_image_bytes(ix->bits, r16, nb) : group_stride * groups;
    int rc = upload_mq_image(ix, c, nq, nb, bf16, i8, img, group_stride, c->work);
    if (rc) return rc;
    if (!c->d_thr) HIPCHK(hipMalloc((void **)&c->d_thr, 256 * sizeof(float)));
    if (!c->h_thr) HIPCHK(hipHostMalloc((void **)&c->h_thr, 256 * sizeof(float), hipHostMallocDefault));
    for (int q = 0; q < nq; q++) c->h_thr[q] = thr[q];
    HIPCHK(hipMemcpyAsync(c->d_thr, c->h_thr, sizeof(float) * nq, hipMemcpyHostToDevice, c->work));

    szg::MqArgs a = mq_args_base(ix, sh, c, nq, groups, group_stride);
    a.collect = 1;
    a.thr = c->d_thr;
    a.cand_buf = c->d_collect;
    a.cand_count = c->d_count;
    a.cand_cap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu);
    a.live_bits = sh->has_dead ? sh->live_bits : nullptr;
    a.allow_bits = has_allow ? c->d_allow : nullptr;
    a.allow_stride = (uint32_t)shard_words(sh);
    {
        std::lock_guard<std::mutex> lk(sh->chain_mu);
        hipStream_t st = ix->serialize_scans ? sh->scan_stream : c->work;
        if (st != c->work) {  // the sweep must see the image, the thresholds, the masks and the zeroed counters
            HIPCHK(hipEventRecord(c->ev_up, c->work));
            HIPCHK(hipStreamWaitEvent(st, c->ev_up, 0));
        }
        if (ix->timing) HIPCHK(hipEventRecord(c->ev_scan0, st));
        if (bf16) HIPCHK(szg::launch_mq_score_bf16(ix->bits, a, nb, sh->cu_count, st));
        else HIPCHK(szg::launch_mq_score_i8(ix->bits, a, nb, sh->cu_count, st));
        if (ix->timing) {
            HIPCHK(hipEventRecord(c->ev_scan1, st));
            c->timed_scan = true;
            c->timed_n = groups;
        }
        if (st != c->work) {
            HIPCHK(hipEventRecord(c->ev_scan_done, st));
            HIPCHK(hipStreamWaitEvent(c->work, c->ev_scan_done, 0));
        }
    }
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    ix->stats.scan_launches += (uint64_t)groups;
    ix->stats.scan_bytes += (uint64_t)groups * sh->n_rows * (uint64_t)ix->row_bytes;
    ix->stats.mq_launches += (uint64_t)groups;
    ix->stats.mq_queries += (uint64_t)nq;
    ix->stats.mq_bf16_sweeps += bf16 ? 1 : 0;
    return SZG_OK;
}

}  // namespace szgi
