// scan_handle.cpp -- the handle: lifetime, per-shard contexts, mutations, tunables, statistics.
#include "scan_internal.h"

namespace szgi {

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
    HIPCHK(hipEventCreate(&c->ev_p0));
    HIPCHK(hipEventCreate(&c->ev_p1));
    HIPCHK(hipEventCreateWithFlags(&c->ev_scan_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    const size_t B = kMaxBatch;
    HIPCHK(hipHostMalloc((void **)&c->h_qsw, B * ix->qsw_bytes, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c->h_q64, B * sizeof(double) * ix->dim, hipHostMallocDefault));
    // hit counters of the collect sweeps, one 128-byte line per sweep of a launch
    const size_t n_count = (size_t)kMaxBatch * szg::kCandCountStride;  // (a batch may hold kMaxBatch sweeps)
    HIPCHK(hipHostMalloc((void **)&c->h_count, sizeof(uint32_t) * n_count, hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&c->d_qsw, B * ix->qsw_bytes));
    HIPCHK(hipMalloc((void **)&c->d_q64, B * sizeof(double) * ix->dim));
    HIPCHK(hipMalloc((void **)&c->d_count, sizeof(uint32_t) * n_count));
    return SZG_OK;
}

void ctx_free(Ctx *c)
{
    if (!c) return;
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (hipEvent_t e : {c->ev_scan0, c->ev_scan1, c->ev_all0, c->ev_all1, c->ev_scan_done, c->ev_up, c->ev_p0, c->ev_p1})
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
    c->work = c->stream;
    c->early_n = 0;
    return c;
}
Ctx *ctx_try_acquire(Shard *sh)
{
    std::lock_guard<std::mutex> lk(sh->mu);
    if (sh->free_ctx.empty()) return nullptr;
    Ctx *c = sh->free_ctx.back();
    sh->free_ctx.pop_back();
    c->work = c->stream;
    c->early_n = 0;
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
        sh->norm_valid = 0;
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

// The shard new rows go to.  Ranges stay contiguous in row order, so only the last shard
// that holds rows can grow -- or the next, still empty one can start, which it does only at
// a 64-row boundary (every shard's first row must be a multiple of 64: filter and tombstone
// bitmaps are split between shards by whole words) and once its predecessor holds 4M rows.
// a row was rewritten in place: the sketch pre-pass (if this index keeps one) re-sketches it at its next sync
void note_overwritten(szg_index *ix, uint64_t row)
{
    ix->gen++;
    if (!ix->sketch || ix->sk_need_full) return;  // no sketch yet (or a full rebuild pending): nothing to track
    if (ix->sk_dirty_rows.size() >= 4096) {       // more than a rebuild is worth
        ix->sk_need_full = true;
        ix->sk_dirty_rows.clear();
        return;
    }
    ix->sk_dirty_rows.push_back(row);
}

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

}  // namespace szgi

using namespace szgi;

extern "C" {

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
        (void)hipFree(sh->row_norm);
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
    if (ix) { ix->gen++; ix->sk_need_full = true; ix->sk_disabled = false; }
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
    if (ix) { ix->gen++; ix->sk_need_full = true; ix->sk_disabled = false; }
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
    if (!ix || !vector) return fail(SZG_E_INVALID, "null argument");
    uint64_t local;
    Shard *sh = shard_of(ix, row, &local);
    if (!sh) return fail(SZG_E_RANGE, "row out of range");
    note_overwritten(ix, row);
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
    if (e == hipSuccess && sh->row_norm && local < sh->norm_valid)
        e = szg::launch_row_norms(ix->bits, sh->rows, ix->layout, ix->dim, (float)ix->norm_bias, local, 1, sh->row_norm, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(SZG_E_DEVICE, "overwrite_f64", e);
    return SZG_OK;
    SZG_CATCH
}

int szg_index_overwrite(szg_index *ix, uint64_t row, const uint8_t *row_bytes)
{
    SZG_TRY
    if (!ix || !row_bytes) return fail(SZG_E_INVALID, "null argument");
    uint64_t local;
    Shard *sh = shard_of(ix, row, &local);
    if (!sh) return fail(SZG_E_RANGE, "row out of range");
    note_overwritten(ix, row);
    HIPCHK(hipSetDevice(sh->device));
    HIPCHK(hipDeviceSynchronize());
    const int rc = upload_rows(ix, sh, local, row_bytes, 1);
    if (rc == SZG_OK && sh->row_norm && local < sh->norm_valid) {
        HIPCHK(szg::launch_row_norms(ix->bits, sh->rows, ix->layout, ix->dim, (float)ix->norm_bias, local, 1, sh->row_norm, nullptr));
        HIPCHK(hipStreamSynchronize(nullptr));
    }
    return rc;
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
    if (n == "slack") {
        if (value < 0 || value > 4096) return fail(SZG_E_INVALID, "slack out of range");
        ix->slack_min = (int)value;
    } else if (n == "queries_per_launch") {
        if (value < 1 || value > szg::kMaxSweepsPerLaunch)
            return fail(SZG_E_INVALID, "queries_per_launch out of range");
        ix->queries_per_launch = (int)value;
    } else if (n == "query_batch") {
        if (value < 1 || value > kMaxBatch) return fail(SZG_E_INVALID, "query_batch out of range");
        ix->query_batch = (int)value;
    } else if (n == "radius_mq") {
        if (value < 0 || value > 1) return fail(SZG_E_INVALID, "radius_mq is 0 or 1");
        ix->radius_mq = (int)value;
    } else if (n == "finish_thread") {
        if (value < 0 || value > 1) return fail(SZG_E_INVALID, "finish_thread is 0 or 1");
        ix->finish_thread = (int)value;
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
    } else if (n == "multi_query") {
        ix->multi_query = value != 0;
    } else if (n == "mask_dense") {
        ix->mask_dense = value != 0;
    } else if (n == "coalesce") {
        ix->coalesce = value != 0;
    } else if (n == "force_matrix") {
        ix->force_matrix = value != 0;
    } else if (n == "force_no_refine") {
        ix->force_no_refine = value != 0;
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
    // the sketch index follows the same tunables (the last value of each, replayed when it is created)
    bool seen = false;
    for (auto &o : ix->opt_log)
        if (o.first == n) {
            o.second = value;
            seen = true;
        }
    if (!seen) ix->opt_log.emplace_back(n, value);
    if (ix->sketch) (void)szg_set_option(ix->sketch, name, value);
    return SZG_OK;
    SZG_CATCH
}

}  // extern "C"
