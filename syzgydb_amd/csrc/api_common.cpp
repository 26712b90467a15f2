// api_common.cpp -- error text, the container/heap replay and the cross-shard merge: host code only.
#include "scan_internal.h"

namespace szgi {

thread_local std::string g_last_error;

int fail(int code, const char *what, hipError_t e)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof(buf), "%s", what);
    g_last_error = buf;
    return code;
}

SiteTimers g_sites;

int64_t row_bytes_of(int bits, int dim)
{   // getVectorSize, collection.go:796-811
    switch (bits) {
    case 4: return ((int64_t)dim + 1) / 2;
    case 8: return dim;
    case 16: return (int64_t)dim * 2;
    case 32: return (int64_t)dim * 4;
    case 64: return (int64_t)dim * 8;
    default: return -1;
    }
}

// consider()'s top-k branch replayed over the candidates in visit order
// (collection.go:606-619), then the ascending pop loop (:694-697).
void replay_topk(std::vector<Cand> &cands, int k, std::vector<HeapItem> *result)
{
    std::sort(cands.begin(), cands.end(), [](const Cand &x, const Cand &y) { return x.row < y.row; });
    GoHeap h;
    for (const Cand &c : cands) h.consider_topk(c.row, c.dist, k);
    h.drain(result);
}

// True when the reference's answer may depend on its whole heap history: a NaN
// distance, or two exactly equal distances among the best k+1 candidates.
bool history_dependent(const double *dist, size_t n, int k)
{
    std::vector<double> d;
    d.reserve(n);
    for (size_t i = 0; i < n; i++) {
        if (std::isnan(dist[i])) return true;
        d.push_back(dist[i]);
    }
    const size_t m = std::min(d.size(), (size_t)k + 1);
    std::partial_sort(d.begin(), d.begin() + m, d.end());
    for (size_t i = 1; i < m; i++)
        if (d[i] == d[i - 1]) return true;
    return false;
}

double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// consider()'s top-k branch over the union of the lists in visit order; get(l, q, i, &row, &dist)
template <typename Count, typename Get>
int merge_lists(int k, int n_lists, int list_len, int n_queries, Count count_of, Get get, uint64_t *out_rows,
                double *out_dist, int32_t *out_count, uint8_t *out_history_dependent)
{
    std::vector<Cand> cands;
    std::vector<HeapItem> res;
    std::vector<double> d;
    for (int q = 0; q < n_queries; q++) {
        cands.clear();
        for (int l = 0; l < n_lists; l++) {
            const int n = std::min<int>(std::max<int>(count_of(l, q), 0), list_len);
            for (int i = 0; i < n; i++) {
                Cand c{0, 0.0, 0.0f};
                get(l, q, i, &c.row, &c.dist);
                cands.push_back(c);
            }
        }
        replay_topk(cands, k, &res);
        if (out_history_dependent) {
            d.resize(cands.size());
            for (size_t i = 0; i < cands.size(); i++) d[i] = cands[i].dist;
            out_history_dependent[q] = history_dependent(d.data(), d.size(), k) ? 1 : 0;
        }
        for (int i = 0; i < k; i++) {
            const bool have = i < (int)res.size();
            out_rows[(size_t)q * k + i] = have ? res[i].row : UINT64_MAX;
            out_dist[(size_t)q * k + i] = have ? res[i].priority : 0.0;
        }
        if (out_count) out_count[q] = (int32_t)res.size();
    }
    return SZG_OK;
}

}  // namespace szgi

using namespace szgi;

extern "C" {

int szg_abi_version(void) { return SZG_ABI_VERSION; }

const char *szg_last_error(void) { return g_last_error.c_str(); }

const char *szg_strerror(int code)
{
    switch (code) {
    case SZG_OK: return "ok";
    case SZG_E_INVALID: return "invalid argument";
    case SZG_E_NOMEM: return "out of memory";
    case SZG_E_DEVICE: return "HIP runtime error";
    case SZG_E_TRUNCATED: return "result truncated: more hits than capacity";
    case SZG_E_NODEVICE: return "no usable gfx950 device";
    case SZG_E_RANGE: return "row index out of range";
    case SZG_E_UNSUPPORTED: return "outside this build's limits";
    default: return "unknown error";
    }
}

int64_t szg_row_bytes(int quant_bits, int dim)
{
    if (dim <= 0) return -1;
    return row_bytes_of(quant_bits, dim);
}

/*
 * Cross-shard result assembly for one-process-per-GPU sharding: every rank
 * answers the query on its row range with szg_search_topk (list_len = k+1
 * results, rows already global via szg_index_set_row_base), the per-rank
 * lists are exchanged (RCCL all-gather) and this replays consider()'s top-k
 * branch over their union in visit order.  Pure host code.
 */
int szg_merge_topk(int k, int n_lists, int list_len, int n_queries, const uint64_t *rows,
                   const double *dist, const int32_t *counts, uint64_t *out_rows, double *out_dist,
                   int32_t *out_count, uint8_t *out_history_dependent)
{
    if (k <= 0 || n_lists <= 0 || list_len <= 0 || n_queries < 0 || !rows || !dist || !counts ||
        !out_rows || !out_dist)
        return fail(SZG_E_INVALID, "bad argument");
    try {
        return merge_lists(
            k, n_lists, list_len, n_queries, [&](int l, int q) { return counts[(size_t)l * n_queries + q]; },
            [&](int l, int q, int i, uint64_t *r, double *d) {
                const size_t at = ((size_t)l * n_queries + q) * list_len + i;
                *r = rows[at];
                *d = dist[at];
            },
            out_rows, out_dist, out_count, out_history_dependent);
    } catch (const std::bad_alloc &) {
        return fail(SZG_E_NOMEM, "out of memory");
    }
}

/* The same merge straight from the exchanged records (no repacking on the caller's side):
 * records[n_lists][n_queries][2*list_len + 1] int64 = list_len rows | list_len float64 bit
 * patterns | count -- exactly what each rank contributes to the all-gather. */
int szg_merge_topk_records(int k, int n_lists, int list_len, int n_queries, const int64_t *records,
                           uint64_t *out_rows, double *out_dist, int32_t *out_count,
                           uint8_t *out_history_dependent)
{
    if (k <= 0 || n_lists <= 0 || list_len <= 0 || n_queries < 0 || !records || !out_rows || !out_dist)
        return fail(SZG_E_INVALID, "bad argument");
    const size_t rec = 2 * (size_t)list_len + 1;
    try {
        return merge_lists(
            k, n_lists, list_len, n_queries,
            [&](int l, int q) { return (int)records[((size_t)l * n_queries + q) * rec + 2 * list_len]; },
            [&](int l, int q, int i, uint64_t *r, double *d) {
                const int64_t *p = records + ((size_t)l * n_queries + q) * rec;
                *r = (uint64_t)p[i];
                memcpy(d, &p[list_len + i], sizeof(double));
            },
            out_rows, out_dist, out_count, out_history_dependent);
    } catch (const std::bad_alloc &) {
        return fail(SZG_E_NOMEM, "out of memory");
    }
}

}  // extern "C"
