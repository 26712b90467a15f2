// scan_query.cpp -- the query as the kernels want it, and the bounds on what their arithmetic loses.
#include "scan_internal.h"

namespace szgi {

szg::RowMap choose_map(int r16, bool tiled)
{
    if (tiled) return szg::RowMap{r16, 4, r16 / 4, 16, 1, 1};  // one 64-byte step of 16 rows per load   // Groups of L lanes per row, P pieces per lane.
    // 1) Exact power-of-two groups (L*P == r16): every lane always holds a piece (the
    //    kernel's dense phase), reductions are DPP.  The SMALLEST such L >= 8 wins: a
    //    group still reads whole 128-byte lines per load, and the fewer lanes share a
    //    row, the more pieces each walks between two row finishes (measured on
    //    1M x 768 f32: L=8 6.80 TB/s, L=16 6.73, L=32 6.47, L=64 6.44; 8-bit: L=8 5.9 vs
    //    L=16 5.75 vs L=32 3.8).  L=4 (64-byte segments) only when nothing wider is exact.
    for (int L : {8, 16, 32, 64, 4})
        if (r16 % L == 0) return szg::RowMap{r16, L, r16 / L, 64 / L, 1, 1};
    // 2) Otherwise maximise lane utilisation, with a bonus for power-of-two groups.
    szg::RowMap best{r16, 64, (r16 + 63) / 64, 1, 1, 0};
    double best_score = -1;
    const int pmax = std::max(1, (r16 + 63) / 64 + 8);
    for (int P = 1; P <= pmax; P++) {
        const int need = (r16 + P - 1) / P;  // lanes a row needs at P pieces per lane
        if (need > 64) continue;
        int cand[2] = {need, 1};
        while (cand[1] < need) cand[1] <<= 1;  // next power of two
        for (int L : cand) {
            if (L > 64) continue;
            const int gpw = 64 / L;
            const double util = (double)gpw * r16 / (64.0 * P);
            const bool pow2 = (L & (L - 1)) == 0;
            const double score = util * (pow2 ? 1.3 : 1.0);
            if (score > best_score + 1e-9) {
                best_score = score;
                best = szg::RowMap{r16, L, P, gpw, pow2 ? 1 : 0, 0};
            }
        }
    }
    best.dense = (best.L * best.P == r16 && best.gpw * best.L == 64) ? 1 : 0;
    return best;
}

// round to nearest (ties away from zero) without a libm call; NaN -> 0, clamped to +-lim.
// Any rounding rule serves: Q only has to be within 1/2 of v/qscale (key_eps).
inline long long round_clamp(double t, double lim)
{
    if (!(t == t)) return 0;
    if (t > lim) t = lim;
    if (t < -lim) t = -lim;
    return (long long)(t + (t >= 0 ? 0.5 : -0.5));
}

// Query as the scan wants it (see RowAcc in kernels_scan.hip):
//  * 16/32/64-bit rows: float (double for 64-bit), pre-normalised for cosine,
//    pre-scaled by maxInt for 16-bit euclid, laid out [chunk][piece][4];
//  * 8/4-bit rows: the prepared real query v (q/|q| for cosine, maxInt*q for
//    euclid) quantized to integers Q_i = round(v_i / qscale) and split into
//    balanced digit planes (3 x int8 radix 128, or 5 x int4 radix 16), one
//    16-byte plane word per 16-byte piece of the row.
// The constants every path needs (norms of the caller's and of the prepared query); returns the largest
// |prepared element| and the scale q -> prepared query.
static double prep_query_norms(const szg_index *ix, const double *q, QMeta *meta, double *scale_out)
{
    const int dim = ix->dim, bits = ix->bits;
    *meta = QMeta{};
    double m1 = 0.0;
    for (int i = 0; i < dim; i++) m1 += q[i] * q[i];
    meta->m1 = m1;
    double scale = 1.0;
    if (ix->metric == SZG_COSINE) {
        scale = m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0;
    } else if (bits <= 16) {
        scale = (double)((1u << bits) - 1u);
    }
    double nrm = 0.0, vmax = 0.0;
    for (int e = 0; e < dim; e++) {
        const double v = q[e] * scale;
        nrm += v * v;
        vmax = std::max(vmax, std::fabs(v));
    }
    meta->qnorm = std::sqrt(nrm);
    meta->qnorm2 = nrm;
    *scale_out = scale;
    return vmax;
}

// Shared sweeps stage their own image of the batch; the single-query form below is only needed if a query of the
// batch escalates (a collect sweep), so a batch prepares just the constants and builds that form on demand.
void prep_query_meta(const szg_index *ix, const double *q, QMeta *meta)
{
    double scale;
    (void)prep_query_norms(ix, q, meta, &scale);
}

void prep_query(const szg_index *ix, const double *q, uint8_t *out_sw, QMeta *meta)
{
    const int dim = ix->dim, bits = ix->bits;
    const int E = 128 / bits;
    const int r16 = ix->map.r16;
    memset(out_sw, 0, ix->qsw_bytes);
    double scale;
    const double vmax = prep_query_norms(ix, q, meta, &scale);
    if (bits == 8 || bits == 4) {
        const double Qmax = bits == 8 ? 1000000.0 : szg::kQmax4;
        const double qs = (vmax > 0 && std::isfinite(vmax)) ? vmax / Qmax : 1.0;
        meta->qscale = qs;
        double sumQ = 0.0;
        uint32_t *planes = reinterpret_cast<uint32_t *>(out_sw);
        for (int e = 0; e < dim; e++) {
            long long Q = round_clamp(q[e] * scale / qs, Qmax);
            sumQ += (double)Q;
            const int j = e / E, i = e % E;
            if (bits == 8) {
                const int d = i / 4, kb = i % 4;
                for (int x = 2; x >= 0; x--) {  // planes: 0 = h (x16384), 1 = m (x128), 2 = l
                    long long dig;
                    if (x > 0) {
                        dig = ((Q + 64) & 127) - 64;
                        Q = (Q - dig) >> 7;
                    } else {
                        dig = Q;
                    }
                    planes[((size_t)x * r16 + j) * 4 + d] |= (uint32_t)((uint8_t)(int8_t)dig) << (8 * kb);
                }
            } else {
                // byte b of the piece holds element 2b in its high nibble, 2b+1 in the low one
                const int bb = i / 2, d = bb / 4, kb = bb % 4;
                const int t4 = 2 * kb + ((i % 2 == 0) ? 1 : 0);
                for (int x = 0; x < szg::kPlanes4; x++) {  // plane x carries the digit of weight 16^x
                    long long dig;
                    if (x < szg::kPlanes4 - 1) {
                        dig = ((Q + 8) & 15) - 8;
                        Q = (Q - dig) >> 4;
                    } else {
                        dig = Q;
                    }
                    planes[((size_t)x * r16 + j) * 4 + d] |= (uint32_t)(dig & 0xF) << (4 * t4);
                }
            }
        }
        meta->qconst = sumQ;
        return;
    }
    for (int e = 0; e < dim; e++) {
        const double v = q[e] * scale;
        const int j = e / E, i = e % E;
        if (bits == 64) {
            reinterpret_cast<double *>(out_sw)[(size_t)j * 2 + i] = v;
        } else {
            const int c = i / 4, m = i % 4;
            reinterpret_cast<float *>(out_sw)[((size_t)c * r16 + j) * 4 + m] = (float)v;
        }
    }
}

// Bound on |scan key - real-number key| (see DESIGN.md "certification").
double key_eps(const szg_index *ix, double key, const QMeta &m)
{
    const double k = std::fabs(key);
    if (m.mq) {
        // shared sweep: float32 everywhere (quantized rows decode to exact integers first).
        // cosine: dot and norm each carry <= (dim+16) u relative error.  euclid: the key is
        // |x|^2 - 2 x.g + |g|^2, three float32 sums whose magnitudes are bounded by
        // (|x| + |g|)^2 <= (2|g| + sqrt(key))^2 -- an absolute bound, far looser than the
        // difference form's when rows sit far from the origin; certification then simply
        // escalates more often.
        // (64-bit rows are narrowed to float32 element by element first: one more rounding of 2^-24 per operand,
        // i.e. u |x||g| on the dot product and 2u |x|^2 on the norm -- four more units of n cover both)
        const double u = 0x1p-24, n = (double)ix->dim + 16.0 + (ix->bits == 64 ? 4.0 : 0.0);
        if (m.mq_bf16) {
            // bfloat16 sweep: each operand is rounded to 8 significant bits -- a bfloat16 keeps 7 fraction bits, so the
            // spacing at 1 is 2^-7 and round-to-nearest moves a value by at most 2^-8 of itself (the query once more
            // from float32) -- the products are exact in float32 and summed by the matrix core in float32.
            // |sum bf(x_i) bf(g_i) - sum x_i g_i| <= c |x| |g| (Cauchy-Schwarz) with
            // c = (1 + 2^-8)^2 (1 + 2^-24) - 1 < 1.01 * 2^-7; the float32 part of the bound is doubled (the
            // accumulation order and rounding of the matrix core are its own).  (Rounds 2-3 had 2^-9 per operand
            // here, i.e. half this band: a 2-dimensional corpus far from the origin, where both elements can round
            // the same way, showed a key 0.0044 off -- scripts/fuzz_gpu.py seed 311.  At >= 16 dimensions the
            // errors never lined up, which is why nothing had caught it.)
            const double c = 1.01 * 0x1p-7;
            if (ix->metric == SZG_COSINE) return c + 4.0 * n * u + 1e-6;
            // euclid: the key moves by 2 c |x| |g|, and |x| <= |g| + d with d^2 <= key + 2 c |x| |g|
            // gives |x| <= 1.1 |g| + sqrt(key) for this c; the last term keeps key - eps(key) monotone
            const double s = 2.0 * m.qnorm + std::sqrt(k);
            return 2.0 * c * m.qnorm * (1.1 * m.qnorm + std::sqrt(k)) + c * c * m.qnorm2 + 3.0 * n * u * s * s + 1e-30;
        }
        if (ix->metric == SZG_COSINE) return 2.0 * n * u;
        const double s = 2.0 * m.qnorm + std::sqrt(k);
        return 1.5 * n * u * s * s + 1e-30;
    }
    if (m.mq_int) {
        // as the integer branch below with the sweep's own quantization step; the row operand is
        // v' = v - 128 (8-bit rows) or the nibble x in 0..15 (4-bit rows)
        const double M = (double)((1u << ix->bits) - 1u);
        const double V = ix->bits == 8 ? 128.0 : 16.0, Qmax = szg::kMqQmax;
        const double fl = 16.0 * 0x1p-24 * m.mq_qscale * Qmax * V * (double)ix->dim;
        if (ix->metric == SZG_COSINE)
            return 0.5 * m.mq_qscale * std::sqrt((double)ix->dim) + fl / std::sqrt((double)ix->dim) + 0x1p-21;
        return m.mq_qscale * M * (double)ix->dim + 2.0 * fl +
               0x1p-21 * (k + m.qnorm2 + M * M * (double)ix->dim) + 1e-30;
    }
    if (ix->bits == 8 || ix->bits == 4) {
        // integer paths: the per-lane sums are exact.  What is left is (a) the query's
        // quantization, |v_i - qscale*Q_i| <= qscale/2, and (b) the float32 roundings of
        // the row finish: the plane combination and the reduction over the lanes act on
        // terms bounded by sum |Q_i||v'_i| <= Qmax*V*dim (V = 128 resp. 8), i.e. an
        // absolute error <= 16*2^-24 * qscale*Qmax*V*dim in units of sum v n.
        const double M = (double)((1u << ix->bits) - 1u);
        const double V = ix->bits == 8 ? 128.0 : 8.0;
        const double Qmax = ix->bits == 8 ? 1000000.0 : szg::kQmax4;
        const double fl = 16.0 * 0x1p-24 * m.qscale * Qmax * V * (double)ix->dim;
        if (ix->metric == SZG_COSINE)  // divided by |n| >= sqrt(dim) (every n is odd)
            return 0.5 * m.qscale * std::sqrt((double)ix->dim) + fl / std::sqrt((double)ix->dim) + 0x1p-21;
        return m.qscale * M * (double)ix->dim + 2.0 * fl +
               0x1p-21 * (k + m.qnorm2 + M * M * (double)ix->dim) + 1e-30;
    }
    const double u = ix->bits == 64 ? 0x1p-53 : 0x1p-24;
    const double n = (double)ix->dim + 16.0;
    if (ix->metric == SZG_COSINE) {
        return 2.0 * n * u + (ix->bits == 64 ? 0x1p-22 : 0.0);
    }
    return 2.0 * n * u * k + 8.0 * u * m.qnorm * std::sqrt(k) + (ix->bits == 64 ? 0x1p-22 * k : 0.0) +
           1e-37;
}

// ---- shared sweeps: B queries share one pass of the corpus ------------

// 8-bit rows in whole 64-byte steps (the tiled layout) through the bfloat16 sweep -- their codes are exact in bfloat16,
// 96 queries share a pass instead of 48 (kernels_mq.hip, part 108).  SZG_BF16_8BIT=0 keeps them on the int8 sweep.
static bool bf16_takes_8bit(const szg_index *ix, bool radius, int nq)
{
    if (radius || nq <= 48) return false;  // (a radius IS the threshold: the band of the rounded query would be collected too)
    static const bool on = []() {
        const char *e = getenv("SZG_BF16_8BIT");
        if (getenv("SZG_NO_ROW_NORMS")) return false;  // (the kernel takes the rows' norms from the resident array)
        return e ? atoi(e) != 0 : SZG_BF16_8BIT_DEFAULT != 0;
    }();
    return on && ix->bits == 8 && ix->layout.tiled && ix->mq_bf16;
}
bool mq_uses_i8(const szg_index *ix, bool radius, int nq)
{
    return (ix->bits == 8 || ix->bits == 4) && ix->mq_i8 && !bf16_takes_8bit(ix, radius, nq);
}
// the bfloat16 sweep: 64-, 32- and 16-bit rows of any dimension (not the experimental tiled layout of wide rows), and
// tiled 8-bit rows
bool mq_uses_bf16(const szg_index *ix, bool radius, int nq)
{
    if (bf16_takes_8bit(ix, radius, nq)) return true;
    if (!ix->mq_bf16 || ix->layout.tiled) return false;
    return ix->bits == 64 || ix->bits == 32 || ix->bits == 16;
}

// round to nearest even, as v_cvt_pk_bf16_f32 does (NaN stays NaN)
uint16_t bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// the prepared real query of the integer sweeps: q/|q| (cosine) or maxInt*q (euclid)
double mq_int_scale(const szg_index *ix, double m1)
{
    if (ix->metric == SZG_COSINE) return m1 > 0 ? 1.0 / std::sqrt(m1) : 0.0;
    return (double)((1u << ix->bits) - 1u);
}

// int8 sweep: quantization step, the integer query Q (dim values) and its digit sum
void prep_mq_int(const szg_index *ix, const double *q, QMeta *meta, int32_t *Qout)
{
    const double scale = mq_int_scale(ix, meta->m1);
    double vmax = 0.0;
    for (int e = 0; e < ix->dim; e++) vmax = std::max(vmax, std::fabs(q[e] * scale));
    const double Qmax = szg::kMqQmax;
    const double qs = (vmax > 0 && std::isfinite(vmax)) ? vmax / Qmax : 1.0;
    const double inv = scale / qs;
    long long sumQ = 0;
    for (int e = 0; e < ix->dim; e++) {
        const long long Q = round_clamp(q[e] * inv, Qmax);
        Qout[e] = (int32_t)Q;
        sumQ += Q;
    }
    meta->mq_int = true;
    meta->mq_qscale = qs;
    meta->mq_qconst = (double)sumQ;
}

int mq_blocks(const szg_index *ix, int nq, bool radius)
{   // query blocks of 16 the batch gets (nq = the queries left in the call), or 0 when the shared sweep does not apply
    if (!ix->multi_query || nq < ix->mq_min) return 0;
    const bool bf16 = mq_uses_bf16(ix, radius, nq), i8 = mq_uses_i8(ix, radius, nq);
    if (!bf16 && !i8) return 0;  // (switched off, or the experimental tiled layout of wide rows): one sweep per query
    int nb = std::min((nq + 15) / 16, std::min(ix->mq_blocks_max, bf16 ? 6 : 3));
    auto fits = [&](int n) {  // the image (+ tables, hit buffers, staging) must fit LDS
        if (bf16) return szg::mq_bf16_lds_bytes(ix->bits, ix->map.r16, n) <= 160u * 1024u;
        return szg::mq_i8_lds_bytes(ix->bits, ix->map.r16, n) <= 150u * 1024u;
    };
    while (nb > 0 && !fits(nb)) nb--;
    return nb;
}

}  // namespace szgi
