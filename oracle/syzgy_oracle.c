/*
 * syzgy_oracle.c -- CPU restatement of SyzgyDB's brute-force scan
 * (Collection.Search, Precision "exact") in plain C.
 *
 * TEST INFRASTRUCTURE ONLY -- see syzgy_oracle.h for who may use it and for
 * the parity-pinning status.  Every function cites the reference lines
 * (/root/reference/<file>:<lines>) it restates.  float64 everywhere,
 * sequential left-to-right sums, no FMA contraction (-ffp-contract=off).
 */
#include "syzgy_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

/* ---------------------------------------------------------------- quantizer */

/* quantization.go:5-23 */
uint64_t orc_quantize(double value, int bits)
{
    if (bits == 32) {
        float f = (float)value; /* float32(value): round-to-nearest-even */
        uint32_t u;
        memcpy(&u, &f, 4);
        return (uint64_t)u;
    }
    if (bits == 64) {
        uint64_t u;
        memcpy(&u, &value, 8);
        return u;
    }
    if (value < -1) {
        value = -1;
    } else if (value > 1) {
        value = 1;
    }
    int64_t maxInt = ((int64_t)1 << bits) - 1;
    double q = (value + 1) / 2 * (double)maxInt;
    return (uint64_t)round(q); /* math.Round: half away from zero */
}

/* quantization.go:25-36 */
double orc_dequantize(uint64_t value, int bits)
{
    if (bits == 32) {
        uint32_t u = (uint32_t)value;
        float f;
        memcpy(&f, &u, 4);
        return (double)f;
    }
    if (bits == 64) {
        double d;
        memcpy(&d, &value, 8);
        return d;
    }
    int64_t maxInt = ((int64_t)1 << bits) - 1;
    return ((double)value / (double)maxInt) * 2 - 1;
}

/* collection.go:796-811 */
int64_t orc_vector_size(int bits, int dim)
{
    switch (bits) {
    case 4: return ((int64_t)dim + 1) / 2;
    case 8: return dim;
    case 16: return (int64_t)dim * 2;
    case 32: return (int64_t)dim * 4;
    case 64: return (int64_t)dim * 8;
    default: return -1; /* reference panics */
    }
}

/* collection.go:713-743; binary.BigEndian puts */
void orc_encode_vector(const double *vec, int dim, int bits, uint8_t *out)
{
    int64_t size = orc_vector_size(bits, dim);
    if (size < 0) return;
    memset(out, 0, (size_t)size);
    for (int i = 0; i < dim; i++) {
        uint64_t q = orc_quantize(vec[i], bits);
        switch (bits) {
        case 4:
            if (i % 2 == 0) {
                out[i / 2] = (uint8_t)(q << 4);
            } else {
                out[i / 2] |= (uint8_t)(q & 0x0F);
            }
            break;
        case 8: out[i] = (uint8_t)q; break;
        case 16:
            out[i * 2] = (uint8_t)(q >> 8);
            out[i * 2 + 1] = (uint8_t)q;
            break;
        case 32:
            for (int b = 0; b < 4; b++) out[i * 4 + b] = (uint8_t)(q >> (24 - 8 * b));
            break;
        case 64:
            for (int b = 0; b < 8; b++) out[i * 8 + b] = (uint8_t)(q >> (56 - 8 * b));
            break;
        }
    }
}

/* collection.go:768-794 */
void orc_decode_vector(const uint8_t *data, int dim, int bits, double *out)
{
    for (int i = 0; i < dim; i++) {
        uint64_t q = 0;
        switch (bits) {
        case 4:
            if (i % 2 == 0) {
                q = (uint64_t)(data[i / 2] >> 4);
            } else {
                q = (uint64_t)(data[i / 2] & 0x0F);
            }
            break;
        case 8: q = data[i]; break;
        case 16: q = ((uint64_t)data[i * 2] << 8) | data[i * 2 + 1]; break;
        case 32:
            for (int b = 0; b < 4; b++) q = (q << 8) | data[i * 4 + b];
            break;
        case 64:
            for (int b = 0; b < 8; b++) q = (q << 8) | data[i * 8 + b];
            break;
        }
        out[i] = orc_dequantize(q, bits);
    }
}

/* ---------------------------------------------------------------- distances */

/* collection.go:812-819 */
double orc_euclidean(const double *a, const double *b, int n)
{
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        double diff = a[i] - b[i];
        sum += diff * diff;
    }
    return sqrt(sum); /* math.Sqrt = SQRTSD, correctly rounded */
}

/*
 * Go stdlib math/atan.go (xatan, satan) and math/asin.go (Asin, Acos).  The
 * amd64 port has no assembly for these; the pure-Go Cephes code runs.
 */
static double go_xatan(double x)
{
    const double P0 = -8.750608600031904122785e-01;
    const double P1 = -1.615753718733365076637e+01;
    const double P2 = -7.500855792314704667340e+01;
    const double P3 = -1.228866684490136173410e+02;
    const double P4 = -6.485021904942025371773e+01;
    const double Q0 = +2.485846490142306297962e+01;
    const double Q1 = +1.650270098316988542046e+02;
    const double Q2 = +4.328810604912902668951e+02;
    const double Q3 = +4.853903996359136964868e+02;
    const double Q4 = +1.945506571482613964425e+02;
    double z = x * x;
    z = z * ((((P0 * z + P1) * z + P2) * z + P3) * z + P4) /
        (((((z + Q0) * z + Q1) * z + Q2) * z + Q3) * z + Q4);
    z = x * z + x;
    return z;
}

static double go_satan(double x)
{
    const double Morebits = 6.123233995736765886130e-17; /* pi/2 = PIO2 + Morebits */
    const double Tan3pio8 = 2.41421356237309504880;      /* tan(3*pi/8) */
    if (x <= 0.66) {
        return go_xatan(x);
    }
    if (x > Tan3pio8) {
        return M_PI / 2 - go_xatan(1 / x) + Morebits;
    }
    return M_PI / 4 + go_xatan((x - 1) / (x + 1)) + 0.5 * Morebits;
}

static double go_asin(double x)
{
    if (x == 0) {
        return x;
    }
    int sign = 0;
    if (x < 0) {
        x = -x;
        sign = 1;
    }
    if (x > 1) {
        return NAN;
    }
    double temp = sqrt(1 - x * x);
    if (x > 0.7) {
        temp = M_PI / 2 - go_satan(temp / x);
    } else {
        temp = go_satan(x / temp);
    }
    if (sign) {
        temp = -temp;
    }
    return temp;
}

double orc_go_acos(double x)
{
    return M_PI / 2 - go_asin(x);
}

/* collection.go:821-832 ("Cosine" is the angular distance acos(cos)/pi) */
double orc_angular(const double *a, const double *b, int n)
{
    double dot = 0.0, m1 = 0.0, m2 = 0.0;
    for (int i = 0; i < n; i++) {
        dot += a[i] * b[i];
        m1 += a[i] * a[i];
        m2 += b[i] * b[i];
    }
    if (m1 == 0 || m2 == 0) {
        return 1.0;
    }
    return orc_go_acos(dot / (sqrt(m1) * sqrt(m2))) / M_PI;
}

static double distance_fn(int metric, const double *q, const double *d, int n)
{
    return metric == ORC_COSINE ? orc_angular(q, d, n) : orc_euclidean(q, d, n);
}

/* --------------------------------------------------- container/heap restated */

typedef struct {
    uint64_t row;
    double priority; /* = distance (collection.go:600, 611) */
} orc_item;

typedef struct {
    orc_item *a;
    int64_t len, cap;
} orc_pq;

/* collection.go:545-547: max-heap on distance */
static int pq_less(const orc_pq *h, int64_t i, int64_t j)
{
    return h->a[i].priority > h->a[j].priority;
}

static void pq_swap(orc_pq *h, int64_t i, int64_t j)
{
    orc_item t = h->a[i];
    h->a[i] = h->a[j];
    h->a[j] = t;
}

/* container/heap.up */
static void heap_up(orc_pq *h, int64_t j)
{
    for (;;) {
        int64_t i = (j - 1) / 2; /* parent; (0-1)/2 == 0 like Go */
        if (i == j || !pq_less(h, j, i)) break;
        pq_swap(h, i, j);
        j = i;
    }
}

/* container/heap.down */
static void heap_down(orc_pq *h, int64_t i0, int64_t n)
{
    int64_t i = i0;
    for (;;) {
        int64_t j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int64_t j = j1;
        int64_t j2 = j1 + 1;
        if (j2 < n && pq_less(h, j2, j1)) j = j2;
        if (!pq_less(h, j, i)) break;
        pq_swap(h, i, j);
        i = j;
    }
}

/* heap.Push = append (collection.go:553-556) + up */
static void heap_push(orc_pq *h, orc_item it)
{
    if (h->len == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->a = (orc_item *)realloc(h->a, (size_t)h->cap * sizeof(orc_item));
    }
    h->a[h->len++] = it;
    heap_up(h, h->len - 1);
}

/* heap.Pop = swap(0,n-1); down(0,n-1); remove last (collection.go:558-564) */
static orc_item heap_pop(orc_pq *h)
{
    int64_t n = h->len - 1;
    pq_swap(h, 0, n);
    heap_down(h, 0, n);
    orc_item it = h->a[n];
    h->len = n;
    return it;
}

/* ------------------------------------------------------------------- search */

int64_t orc_search_exact(const uint8_t *rows, uint64_t n_rows, int dim, int bits,
                         int metric, const double *query, int k, double radius,
                         const uint8_t *allow, uint64_t *out_rows, double *out_dist,
                         uint64_t capacity, uint64_t *points_searched)
{
    int64_t row_bytes = orc_vector_size(bits, dim);
    if (row_bytes < 0 || dim <= 0 || (metric != ORC_EUCLIDEAN && metric != ORC_COSINE)) return -1;
    orc_pq pq = {0, 0, 0};
    uint64_t searched = 0;
    double *vec = (double *)malloc(sizeof(double) * (size_t)dim);

    if (!(radius == 0 && k == 0)) { /* listing mode (:633-669) computes no distances */
        for (uint64_t r = 0; r < n_rows; r++) {
            /* consider(), collection.go:583-629 */
            orc_decode_vector(rows + r * (uint64_t)row_bytes, dim, bits, vec); /* :584 */
            searched++;                                                        /* :589 */
            if (allow && !allow[r]) continue;                                  /* :592-594 */
            double distance = distance_fn(metric, query, vec, dim);            /* :596 */
            if (radius > 0 && distance <= radius) {                            /* :598-603 */
                orc_item it = {r, distance};
                heap_push(&pq, it);
            } else if (radius > 0) {                                           /* :604-605 */
                continue;
            } else if (k > 0) {                                                /* :606-619 */
                if (pq.len <= k) {
                    if (pq.len < k || pq.a[0].priority > distance) {
                        orc_item it = {r, distance};
                        heap_push(&pq, it);
                        if (pq.len > k) heap_pop(&pq);
                    }
                }
            }
        }
    }
    /* :694-697: fill from the end => ascending */
    int64_t n = pq.len;
    for (int64_t i = n - 1; i >= 0; i--) {
        orc_item it = heap_pop(&pq);
        if ((uint64_t)i < capacity) {
            if (out_rows) out_rows[i] = it.row;
            if (out_dist) out_dist[i] = it.priority;
        }
    }
    if (points_searched) *points_searched = searched;
    free(vec);
    free(pq.a);
    return n;
}

void orc_all_distances(const uint8_t *rows, uint64_t n_rows, int dim, int bits,
                       int metric, const double *query, double *out_dist)
{
    int64_t row_bytes = orc_vector_size(bits, dim);
    if (row_bytes < 0) return;
    double *vec = (double *)malloc(sizeof(double) * (size_t)dim);
    for (uint64_t r = 0; r < n_rows; r++) {
        orc_decode_vector(rows + r * (uint64_t)row_bytes, dim, bits, vec);
        out_dist[r] = distance_fn(metric, query, vec, dim);
    }
    free(vec);
}

void orc_distances_for_rows(const uint8_t *rows, const uint64_t *row_ids, uint64_t n_ids,
                            int dim, int bits, int metric, const double *query,
                            double *out_dist)
{
    int64_t row_bytes = orc_vector_size(bits, dim);
    if (row_bytes < 0) return;
    double *vec = (double *)malloc(sizeof(double) * (size_t)dim);
    for (uint64_t i = 0; i < n_ids; i++) {
        orc_decode_vector(rows + row_ids[i] * (uint64_t)row_bytes, dim, bits, vec);
        out_dist[i] = distance_fn(metric, query, vec, dim);
    }
    free(vec);
}

/* ------------------------------------------------------------- visit order */

typedef struct {
    char s[24];
    uint64_t idx;
} orc_idstr;

static int idstr_cmp(const void *a, const void *b)
{
    return strcmp(((const orc_idstr *)a)->s, ((const orc_idstr *)b)->s);
}

/* spanfile.go:540-560: sort.Strings over fmt.Sprintf("%d", id) (collection.go:450) */
void orc_sorted_id_order(const uint64_t *ids, uint64_t n, uint64_t *perm)
{
    orc_idstr *v = (orc_idstr *)malloc(sizeof(orc_idstr) * (size_t)(n ? n : 1));
    for (uint64_t i = 0; i < n; i++) {
        snprintf(v[i].s, sizeof(v[i].s), "%llu", (unsigned long long)ids[i]);
        v[i].idx = i;
    }
    qsort(v, (size_t)n, sizeof(orc_idstr), idstr_cmp);
    for (uint64_t i = 0; i < n; i++) perm[i] = v[i].idx;
    free(v);
}

/* ---------------------------------------------------------- synthetic data */

uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

double orc_synth_value(uint64_t seed, uint64_t index)
{
    uint64_t m = orc_splitmix64(seed + index) >> 11;       /* 53 bits */
    return (double)m * (1.0 / 4503599627370496.0) - 1.0;   /* m * 2^-52 - 1, exact */
}

void orc_synth_vectors(uint64_t seed, uint64_t first_row, uint64_t n_rows, int dim, double *out)
{
    for (uint64_t r = 0; r < n_rows; r++)
        for (int i = 0; i < dim; i++)
            out[r * (uint64_t)dim + (uint64_t)i] =
                orc_synth_value(seed, (first_row + r) * (uint64_t)dim + (uint64_t)i);
}

void orc_synth_rows(uint64_t seed, uint64_t first_row, uint64_t n_rows, int dim, int bits,
                    uint8_t *out)
{
    int64_t row_bytes = orc_vector_size(bits, dim);
    if (row_bytes < 0) return;
    double *vec = (double *)malloc(sizeof(double) * (size_t)dim);
    for (uint64_t r = 0; r < n_rows; r++) {
        orc_synth_vectors(seed, first_row + r, 1, dim, vec);
        orc_encode_vector(vec, dim, bits, out + r * (uint64_t)row_bytes);
    }
    free(vec);
}

/* ------------------------------------------------------------ CPU baseline */

typedef struct {
    const uint8_t *rows;
    uint64_t n_rows;
    int dim, bits, metric, k, n_queries;
    const double *queries;
    uint64_t *out_rows;
    int next; /* guarded by mu */
    pthread_mutex_t *mu;
} bench_ctx;

static void *bench_worker(void *p)
{
    bench_ctx *c = (bench_ctx *)p;
    uint64_t *rows_out = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)c->k);
    double *dist_out = (double *)malloc(sizeof(double) * (size_t)c->k);
    for (;;) {
        pthread_mutex_lock(c->mu);
        int q = c->next++;
        pthread_mutex_unlock(c->mu);
        if (q >= c->n_queries) break;
        uint64_t searched;
        int64_t n = orc_search_exact(c->rows, c->n_rows, c->dim, c->bits, c->metric,
                                     c->queries + (size_t)q * (size_t)c->dim, c->k, 0.0, NULL,
                                     rows_out, dist_out, (uint64_t)c->k, &searched);
        if (c->out_rows) {
            for (int i = 0; i < c->k; i++)
                c->out_rows[(size_t)q * (size_t)c->k + (size_t)i] =
                    i < n ? rows_out[i] : UINT64_MAX;
        }
    }
    free(rows_out);
    free(dist_out);
    return NULL;
}

double orc_bench_topk(const uint8_t *rows, uint64_t n_rows, int dim, int bits, int metric,
                      const double *queries, int n_queries, int k, int threads,
                      uint64_t *out_rows)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    bench_ctx c = {rows, n_rows, dim, bits, metric, k, n_queries, queries, out_rows, 0, &mu};
    pthread_t th[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, bench_worker, &c);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------- faithful per-record baseline */

static uint32_t crc_table[256];
static int crc_ready = 0;

static uint32_t crc32_ieee(const uint8_t *p, size_t n)
{   /* hash/crc32.ChecksumIEEE (spanfile.go:836-838); byte-at-a-time table */
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = 1;
    }
    uint32_t c = 0xFFFFFFFFu;
    while (n--) c = crc_table[(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

static size_t put7(uint8_t *o, uint64_t n)
{   /* write7Code, spanfile.go:568-625 (up to 4 bytes is all that is needed here) */
    if (n < 0x7f) { o[0] = (uint8_t)n; return 1; }
    if (n < 0x3fff) { o[0] = (uint8_t)(((n >> 7) & 0x7f) | 0x80); o[1] = (uint8_t)(n & 0x7f); return 2; }
    if (n < 0x1fffff) {
        o[0] = (uint8_t)(((n >> 14) & 0x7f) | 0x80); o[1] = (uint8_t)(((n >> 7) & 0x7f) | 0x80);
        o[2] = (uint8_t)(n & 0x7f); return 3;
    }
    o[0] = (uint8_t)(((n >> 21) & 0x7f) | 0x80); o[1] = (uint8_t)(((n >> 14) & 0x7f) | 0x80);
    o[2] = (uint8_t)(((n >> 7) & 0x7f) | 0x80); o[3] = (uint8_t)(n & 0x7f); return 4;
}

static int get7(const uint8_t *b, size_t len, size_t *at, uint64_t *out)
{   /* read7Code, spanfile.go:627-636 */
    uint64_t r = 0;
    for (size_t o = *at; o < len; o++) {
        uint64_t d = b[o];
        r = (r << 7) | (d & 0x7f);
        if ((d & 0x80) == 0) { *at = o + 1; *out = r; return 1; }
    }
    return 0;
}

uint64_t orc_spans_build(const uint8_t *rows, uint64_t n_rows, int dim, int bits, int meta_len,
                         uint8_t *out, uint64_t out_cap, uint64_t *offsets)
{
    int64_t rb = orc_vector_size(bits, dim);
    if (rb < 0) return 0;
    uint64_t at = 0;
    for (uint64_t r = 0; r < n_rows; r++) {
        char id[24];
        int idl = snprintf(id, sizeof(id), "%llu", (unsigned long long)r);
        uint8_t hdr[64];
        size_t h = 8;
        h += put7(hdr + h, r + 2);          /* sequence number */
        h += put7(hdr + h, (uint64_t)idl);  /* record id */
        memcpy(hdr + h, id, (size_t)idl); h += (size_t)idl;
        hdr[h++] = 2;                       /* stream count */
        hdr[h++] = 0;                       /* stream 0 = metadata */
        h += put7(hdr + h, (uint64_t)meta_len);
        uint8_t s1[8];
        size_t s1n = 0;
        s1[s1n++] = 1;                      /* stream 1 = packed vector */
        s1n += put7(s1 + s1n, (uint64_t)rb);
        uint64_t len = h + (uint64_t)meta_len + s1n + (uint64_t)rb + 4;
        if (at + len > out_cap) return 0;
        uint8_t *o = out + at;
        memcpy(o, hdr, h);
        o[0] = 0x53; o[1] = 0x50; o[2] = 0x41; o[3] = 0x4E;  /* 'SPAN' */
        o[4] = (uint8_t)(len >> 24); o[5] = (uint8_t)(len >> 16); o[6] = (uint8_t)(len >> 8); o[7] = (uint8_t)len;
        memset(o + h, 'm', (size_t)meta_len);
        memcpy(o + h + meta_len, s1, s1n);
        memcpy(o + h + meta_len + s1n, rows + r * (uint64_t)rb, (size_t)rb);
        uint32_t crc = crc32_ieee(o, (size_t)len - 4);
        o[len - 4] = (uint8_t)(crc >> 24); o[len - 3] = (uint8_t)(crc >> 16);
        o[len - 2] = (uint8_t)(crc >> 8); o[len - 1] = (uint8_t)crc;
        offsets[r] = at;
        at += len;
    }
    return at;
}

double orc_bench_topk_faithful(const uint8_t *spans, const uint64_t *offsets, uint64_t n_rows, int dim,
                               int bits, int metric, const double *queries, int n_queries, int k,
                               uint64_t *out_rows)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int q = 0; q < n_queries; q++) {
        const double *query = queries + (size_t)q * (size_t)dim;
        orc_pq pq = {0, 0, 0};
        for (uint64_t r = 0; r < n_rows; r++) {
            /* getDocument -> ReadRecord -> parseSpan (spanfile.go:730-818) */
            const uint8_t *d = spans + offsets[r];
            uint32_t len = (uint32_t)d[4] << 24 | (uint32_t)d[5] << 16 | (uint32_t)d[6] << 8 | d[7];
            uint32_t want = (uint32_t)d[len - 4] << 24 | (uint32_t)d[len - 3] << 16 |
                            (uint32_t)d[len - 2] << 8 | d[len - 1];
            if (crc32_ieee(d, len - 4) != want) { free(pq.a); return -1.0; }  /* :757 */
            size_t at = 8;
            uint64_t seq, idl, sl;
            if (!get7(d, len, &at, &seq) || !get7(d, len, &at, &idl)) { free(pq.a); return -1.0; }
            at += (size_t)idl;
            int ns = d[at++];
            const uint8_t *vec = NULL;
            for (int i = 0; i < ns; i++) {
                at++;  /* stream id */
                if (!get7(d, len, &at, &sl)) { free(pq.a); return -1.0; }
                if (i == 1) vec = d + at;  /* streams by position (collection.go:476-477) */
                at += (size_t)sl;
            }
            /* decodeVector allocates a fresh []float64 per record (collection.go:769) */
            double *v = (double *)malloc(sizeof(double) * (size_t)dim);
            orc_decode_vector(vec, dim, bits, v);
            double distance = distance_fn(metric, query, v, dim);
            free(v);
            if (pq.len <= k) {
                if (pq.len < k || pq.a[0].priority > distance) {
                    orc_item it = {r, distance};
                    heap_push(&pq, it);
                    if (pq.len > k) heap_pop(&pq);
                }
            }
        }
        int64_t n = pq.len;
        for (int64_t i = n - 1; i >= 0; i--) {
            orc_item it = heap_pop(&pq);
            if (out_rows && i < k) out_rows[(size_t)q * (size_t)k + (size_t)i] = it.row;
        }
        free(pq.a);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------- LSH path (f3) */
/*
 * Restatement of the reference's default ("medium") search path: the random-hyperplane
 * forest of lshtree.go and the traversal lshTree.search (lshtree.go:283-351) driving
 * consider() (collection.go:583-629) per candidate.  Test infrastructure, like the rest of
 * this file.  The forest itself is an INPUT of the search: the reference builds it with Go's
 * math/rand (rand.Intn / NormFloat64, lshtree.go:39-45, :173-180), a stream that cannot be
 * reproduced here, so orc_lsh_build follows the reference's insert / split rules
 * (lshtree.go:102-251) with its own documented generator -- any forest those rules can
 * produce is a valid one, and parity of the SEARCH is defined on a given forest.
 * Documents are identified by their row in `rows` (the caller keeps row <-> id).
 */
typedef struct lsh_node {
    double *normal; /* dim, interior nodes (lshtree.go:48) */
    double b;
    double radius;
    struct lsh_node *left, *right;
    uint64_t *ids;
    int64_t n_ids, cap_ids;
} lsh_node;

struct orc_lsh {
    lsh_node **roots;
    int n_roots, threshold, dim, bits, metric;
    const uint8_t *rows; /* borrowed */
    uint64_t rng;
};

static uint64_t lsh_next(orc_lsh *t) { return orc_splitmix64(t->rng++); }
static int64_t lsh_intn(orc_lsh *t, int64_t n) { return (int64_t)(lsh_next(t) % (uint64_t)n); } /* stands in for rand.Intn */
static double lsh_norm(orc_lsh *t)
{   /* stands in for rand.NormFloat64: Box-Muller on two uniforms of the splitmix stream */
    double u1 = ((double)(lsh_next(t) >> 11) + 1.0) * (1.0 / 9007199254740993.0);
    double u2 = (double)(lsh_next(t) >> 11) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

/* lshtree.go:30-36 */
static double vector_length(const double *v, int n)
{
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum += v[i] * v[i];
    return sqrt(sum);
}

/* lshtree.go:135-144 */
static double dot_product(const double *a, const double *b, int n)
{
    double dot = 0.0;
    for (int i = 0; i < n; i++) dot += a[i] * b[i];
    return dot;
}

/* lshtree.go:55-74 */
static double distance_to_hyperplane(int method, const double *vector, double length, const double *normal,
                                     double b, int dim, int *right)
{
    double dist = dot_product(vector, normal, dim) - b;
    *right = 0;
    if (method == ORC_EUCLIDEAN) {
        if (dist > 0) *right = 1; else dist = -dist;
        return dist;
    }
    dist = orc_go_acos(dist / length) / M_PI; /* angular distance of two already-normalized vectors */
    if (dist > 0.5) {
        *right = 1;
        dist = 1 - dist;
    }
    return dist;
}

static lsh_node *lsh_leaf(void) { return (lsh_node *)calloc(1, sizeof(lsh_node)); }

static void lsh_push_id(lsh_node *n, uint64_t id)
{
    if (n->n_ids == n->cap_ids) {
        n->cap_ids = n->cap_ids ? n->cap_ids * 2 : 16;
        n->ids = (uint64_t *)realloc(n->ids, sizeof(uint64_t) * (size_t)n->cap_ids);
    }
    n->ids[n->n_ids++] = id;
}

static void lsh_doc(const orc_lsh *t, uint64_t id, double *out)
{
    orc_decode_vector(t->rows + id * (uint64_t)orc_vector_size(t->bits, t->dim), t->dim, t->bits, out);
}

/* lshtree.go:171-251 */
static lsh_node *lsh_split(orc_lsh *t, lsh_node *node)
{
    const int dim = t->dim;
    int64_t i1 = lsh_intn(t, node->n_ids), i2;
    do { i2 = lsh_intn(t, node->n_ids); } while (i2 == i1);
    double *d1 = (double *)malloc(sizeof(double) * (size_t)dim * 3), *d2 = d1 + dim, *v = d2 + dim;
    lsh_doc(t, node->ids[i1], d1);
    lsh_doc(t, node->ids[i2], d2);
    int same = 1; /* aboutEqual, :160-169 */
    for (int i = 0; i < dim; i++) if (fabs(d1[i] - d2[i]) > 1e-9) { same = 0; break; }
    if (same) { free(d1); return node; }
    double *normal = (double *)malloc(sizeof(double) * (size_t)dim);
    double b = 0.0;
    for (int i = 0; i < dim; i++) normal[i] = lsh_norm(t);       /* randomNormalizedVector :39-45 */
    {
        double norm = 0.0;
        for (int i = 0; i < dim; i++) norm += normal[i] * normal[i];
        if (norm != 0) { norm = sqrt(norm); for (int i = 0; i < dim; i++) normal[i] = normal[i] / norm; }
    }
    if (t->metric == ORC_EUCLIDEAN) {                              /* :205-207 */
        for (int i = 0; i < dim; i++) v[i] = (d1[i] + d2[i]) / 2;  /* midpoint :146-155 */
        b = sqrt(dot_product(v, v, dim));
    }
    lsh_node *l = lsh_leaf(), *r = lsh_leaf();
    double radius = 0.0;
    for (int64_t i = 0; i < node->n_ids; i++) {                    /* :217-232 */
        lsh_doc(t, node->ids[i], v);
        int right;
        double distance = distance_to_hyperplane(t->metric, v, vector_length(v, dim), normal, b, dim, &right);
        radius = fmax(radius, distance);
        lsh_push_id(right ? r : l, node->ids[i]);
    }
    free(d1);
    if (l->n_ids == 0 || r->n_ids == 0) {                          /* :236-238 */
        free(l->ids); free(r->ids); free(l); free(r); free(normal);
        return node;
    }
    lsh_node *n = lsh_leaf();
    n->normal = normal; n->b = b; n->radius = radius; n->left = l; n->right = r;
    free(node->ids); free(node);
    return n;
}

/* lshtree.go:118-133 */
static lsh_node *lsh_insert(orc_lsh *t, lsh_node *node, uint64_t id, const double *vector, double length)
{
    if (node->left == NULL) {
        lsh_push_id(node, id);
        if (node->n_ids > t->threshold) node = lsh_split(t, node);
        return node;
    }
    int right;
    double distance = distance_to_hyperplane(t->metric, vector, length, node->normal, node->b, t->dim, &right);
    node->radius = fmax(node->radius, distance);
    if (!right) node->left = lsh_insert(t, node->left, id, vector, length);
    else node->right = lsh_insert(t, node->right, id, vector, length);
    return node;
}

orc_lsh *orc_lsh_build(const uint8_t *rows, uint64_t n_rows, int dim, int bits, int metric, int threshold,
                       int num_trees, uint64_t seed)
{
    if (orc_vector_size(bits, dim) < 0 || threshold < 1 || num_trees < 1) return NULL;
    orc_lsh *t = (orc_lsh *)calloc(1, sizeof(orc_lsh));
    t->roots = (lsh_node **)calloc((size_t)num_trees, sizeof(lsh_node *));
    t->n_roots = num_trees; t->threshold = threshold; t->dim = dim; t->bits = bits; t->metric = metric;
    t->rows = rows; t->rng = seed;
    for (int i = 0; i < num_trees; i++) t->roots[i] = lsh_leaf();  /* newLSHTree :84-96 */
    double *v = (double *)malloc(sizeof(double) * (size_t)dim);
    for (uint64_t id = 0; id < n_rows; id++) {                     /* addPoint :98-116, one tree after the other */
        lsh_doc(t, id, v);
        double length = vector_length(v, dim);
        for (int i = 0; i < num_trees; i++) t->roots[i] = lsh_insert(t, t->roots[i], id, v, length);
    }
    free(v);
    return t;
}

static void lsh_free_node(lsh_node *n)
{
    if (!n) return;
    lsh_free_node(n->left); lsh_free_node(n->right);
    free(n->normal); free(n->ids); free(n);
}

void orc_lsh_free(orc_lsh *t)
{
    if (!t) return;
    for (int i = 0; i < t->n_roots; i++) lsh_free_node(t->roots[i]);
    free(t->roots); free(t);
}

static void lsh_count(const lsh_node *n, int64_t *nodes, int64_t *ids)
{
    (*nodes)++; *ids += n->n_ids;
    if (n->left) { lsh_count(n->left, nodes, ids); lsh_count(n->right, nodes, ids); }
}

void orc_lsh_sizes(const orc_lsh *t, int64_t *n_nodes, int64_t *n_ids)
{
    *n_nodes = 0; *n_ids = 0;
    for (int i = 0; i < t->n_roots; i++) lsh_count(t->roots[i], n_nodes, n_ids);
}

static int32_t lsh_flatten(const orc_lsh *t, const lsh_node *n, int32_t *next, int64_t *ids_at, int32_t *left,
                           int32_t *right, double *normals, double *b, int64_t *ids_off, int32_t *ids_cnt,
                           uint64_t *ids)
{
    int32_t me = (*next)++;
    b[me] = n->b;
    ids_off[me] = *ids_at; ids_cnt[me] = (int32_t)n->n_ids;
    for (int64_t i = 0; i < n->n_ids; i++) ids[(*ids_at)++] = n->ids[i];
    for (int i = 0; i < t->dim; i++) normals[(size_t)me * (size_t)t->dim + (size_t)i] = n->normal ? n->normal[i] : 0.0;
    left[me] = right[me] = -1;
    if (n->left) {
        left[me] = lsh_flatten(t, n->left, next, ids_at, left, right, normals, b, ids_off, ids_cnt, ids);
        right[me] = lsh_flatten(t, n->right, next, ids_at, left, right, normals, b, ids_off, ids_cnt, ids);
    }
    return me;
}

/* the forest as flat arrays (what the host mirror of the product traverses) */
void orc_lsh_export(const orc_lsh *t, int32_t *roots, int32_t *left, int32_t *right, double *normals, double *b,
                    int64_t *ids_off, int32_t *ids_cnt, uint64_t *ids)
{
    int32_t next = 0;
    int64_t at = 0;
    for (int i = 0; i < t->n_roots; i++)
        roots[i] = lsh_flatten(t, t->roots[i], &next, &at, left, right, normals, b, ids_off, ids_cnt, ids);
}

/* nodePriorityQueue, lshtree.go:353-381: container/heap, Less = priority > (max-heap) */
typedef struct { const lsh_node *node; double priority; } npq_item;
typedef struct { npq_item *a; int64_t len, cap; } npq;
static int npq_less(const npq *h, int64_t i, int64_t j) { return h->a[i].priority > h->a[j].priority; }
static void npq_swap(npq *h, int64_t i, int64_t j) { npq_item t = h->a[i]; h->a[i] = h->a[j]; h->a[j] = t; }
static void npq_push(npq *h, npq_item it)
{
    if (h->len == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (npq_item *)realloc(h->a, (size_t)h->cap * sizeof(npq_item)); }
    h->a[h->len++] = it;
    for (int64_t j = h->len - 1;;) { /* up */
        int64_t i = (j - 1) / 2;
        if (i == j || !npq_less(h, j, i)) break;
        npq_swap(h, i, j);
        j = i;
    }
}
static npq_item npq_pop(npq *h)
{
    int64_t n = h->len - 1;
    npq_swap(h, 0, n);
    for (int64_t i = 0;;) { /* down(0, n) */
        int64_t j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int64_t j = j1, j2 = j1 + 1;
        if (j2 < n && npq_less(h, j2, j1)) j = j2;
        if (!npq_less(h, j, i)) break;
        npq_swap(h, i, j);
        i = j;
    }
    h->len = n;
    return h->a[n];
}

enum { STOP_SEARCH = 0, POINT_ACCEPTED, POINT_CHECKED, POINT_IGNORED }; /* collection.go:19-24 */

/*
 * Search{Precision: "medium"} (collection.go:685-691 -> lshTree.search, lshtree.go:283-351) with
 * consider() (collection.go:583-629) as the callback.  visit_order (nullable, capacity n_rows)
 * receives the rows in the order consider() saw them.
 */
int64_t orc_lsh_search(const orc_lsh *t, uint64_t n_rows, const double *query, int k, double radius_arg,
                       const uint8_t *allow, uint64_t *out_rows, double *out_dist, uint64_t capacity,
                       uint64_t *points_searched, uint64_t *visit_order)
{
    const int dim = t->dim;
    double radius = radius_arg > 0 ? radius_arg : DBL_MAX;   /* collection.go:686-689, math.MaxFloat64 */
    double length = vector_length(query, dim);
    uint8_t *visited = (uint8_t *)calloc((size_t)(n_rows ? n_rows : 1), 1);
    double *vec = (double *)malloc(sizeof(double) * (size_t)dim);
    const int search_k = 200;
    int k_counter = 0, point_accepted = 0;
    uint64_t searched = 0;
    orc_pq results = {0, 0, 0};
    npq pq = {0, 0, 0};
    for (int i = 0; i < t->n_roots; i++) { npq_item it = {t->roots[i], 0}; npq_push(&pq, it); }
    while (pq.len > 0) {
        npq_item item = npq_pop(&pq);
        const lsh_node *node = item.node;
        const int leaf = node->left == NULL;
        if (item.priority < 0 && -item.priority > radius && leaf) continue;   /* :305-310 */
        if (k_counter >= search_k) break;                                     /* :312-314 */
        if (leaf) {
            for (int64_t ii = 0; ii < node->n_ids; ii++) {
                uint64_t id = node->ids[ii];
                if (visited[id]) continue;
                visited[id] = 1;
                /* consider(), collection.go:583-629 */
                int signal = POINT_CHECKED;
                lsh_doc(t, id, vec);
                if (visit_order) visit_order[searched] = id;
                searched++;
                if (allow && !allow[id]) {
                    signal = POINT_IGNORED;
                } else {
                    double distance = distance_fn(t->metric, query, vec, dim);
                    if (radius_arg > 0 && distance <= radius_arg) {
                        orc_item it = {id, distance};
                        heap_push(&results, it);
                        signal = POINT_ACCEPTED;
                    } else if (radius_arg > 0) {
                        signal = POINT_CHECKED;
                    } else if (k > 0) {
                        if (results.len <= k && (results.len < k || results.a[0].priority > distance)) {
                            orc_item it = {id, distance};
                            heap_push(&results, it);
                            if (results.len > k) heap_pop(&results);
                            radius = results.a[0].priority;                   /* :616 */
                            signal = POINT_ACCEPTED;
                        }
                    }
                }
                switch (signal) {                                             /* lshtree.go:323-334 */
                case POINT_ACCEPTED: k_counter = 0; point_accepted = 1; break;
                case POINT_CHECKED: if (point_accepted) k_counter++; break;
                default: break;
                }
            }
        } else {
            int right;
            double dist = distance_to_hyperplane(t->metric, query, length, node->normal, node->b, dim, &right);
            npq_item a, b;
            if (right) { a.node = node->right; a.priority = dist; b.node = node->left; b.priority = -dist; }
            else { a.node = node->left; a.priority = dist; b.node = node->right; b.priority = -dist; }
            npq_push(&pq, a);
            npq_push(&pq, b);
        }
    }
    int64_t n = results.len;
    for (int64_t i = n - 1; i >= 0; i--) {                                    /* collection.go:694-697 */
        orc_item it = heap_pop(&results);
        if ((uint64_t)i < capacity) {
            if (out_rows) out_rows[i] = it.row;
            if (out_dist) out_dist[i] = it.priority;
        }
    }
    if (points_searched) *points_searched = searched;
    free(visited); free(vec); free(results.a); free(pq.a);
    return n;
}
