// kernels_exact.hip -- float64, reference-operation-order kernels for gfx950.
// Compiled with -ffp-contract=off: Go on amd64 never fuses a*b+c, and the
// reference's distances are sequential float64 sums (collection.go:812-832).
//
//  * rerank_kernel: for the handful of candidates the fused scan kept, decode
//    the row exactly as decodeVector/dequantize do (collection.go:768-794,
//    quantization.go:25-36) and evaluate the reference distance bit for bit.
//  * synth_kernel: counter-based synthetic corpus, quantized and packed exactly
//    as encodeDocument/quantize (collection.go:713-743, quantization.go:5-23).
//  * repack_kernel: page-in transform between the reference's big-endian
//    element encoding and the resident little-endian, 16-byte-pitched layout.
#include "kernels.h"

#pragma clang fp contract(off)

namespace szg {

namespace {

// ---- Go math.Acos (asin.go / atan.go, Cephes), restated ---------------------

__device__ double go_xatan(double x)
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
    double z = __dmul_rn(x, x);
    double num = __dadd_rn(__dmul_rn(P0, z), P1);
    num = __dadd_rn(__dmul_rn(num, z), P2);
    num = __dadd_rn(__dmul_rn(num, z), P3);
    num = __dadd_rn(__dmul_rn(num, z), P4);
    double den = __dadd_rn(z, Q0);
    den = __dadd_rn(__dmul_rn(den, z), Q1);
    den = __dadd_rn(__dmul_rn(den, z), Q2);
    den = __dadd_rn(__dmul_rn(den, z), Q3);
    den = __dadd_rn(__dmul_rn(den, z), Q4);
    z = __ddiv_rn(__dmul_rn(z, num), den);
    z = __dadd_rn(__dmul_rn(x, z), x);
    return z;
}

__device__ double go_satan(double x)
{
    const double Morebits = 6.123233995736765886130e-17;
    const double Tan3pio8 = 2.41421356237309504880;
    const double PiO2 = 1.57079632679489661923132169163975144;
    const double PiO4 = 0.785398163397448309615660845819875721;
    if (x <= 0.66) return go_xatan(x);
    if (x > Tan3pio8) return __dadd_rn(__dsub_rn(PiO2, go_xatan(__ddiv_rn(1.0, x))), Morebits);
    return __dadd_rn(__dadd_rn(PiO4, go_xatan(__ddiv_rn(__dsub_rn(x, 1.0), __dadd_rn(x, 1.0)))),
                     0.5 * Morebits);
}

__device__ double go_asin(double x)
{
    const double PiO2 = 1.57079632679489661923132169163975144;
    if (x == 0) return x;
    bool sign = false;
    if (x < 0) {
        x = -x;
        sign = true;
    }
    if (x > 1) return __longlong_as_double(0x7FF8000000000001ll);  // math.NaN()
    double temp = __dsqrt_rn(__dsub_rn(1.0, __dmul_rn(x, x)));
    if (x > 0.7)
        temp = __dsub_rn(PiO2, go_satan(__ddiv_rn(temp, x)));
    else
        temp = go_satan(__ddiv_rn(x, temp));
    if (sign) temp = -temp;
    return temp;
}

__device__ double go_acos(double x)
{
    const double PiO2 = 1.57079632679489661923132169163975144;
    return __dsub_rn(PiO2, go_asin(x));
}

// ---- exact element decode (resident layout -> reference float64 value) ------

template <int QBITS>
__device__ __forceinline__ double decode_elem_piece(const uint8_t *rp, int i);

template <int QBITS>
__device__ __forceinline__ double decode_elem(const uint8_t *rows, const RowLayout &lay, uint64_t row, int i)
{
    constexpr int E = 128 / QBITS;  // elements per 16-byte piece
    const uint8_t *pp = rows + piece_offset(lay, row, (uint32_t)(i / E));
    return decode_elem_piece<QBITS>(pp, i % E);
}

template <int QBITS>
__device__ __forceinline__ double decode_elem_piece(const uint8_t *rp, int i)
{
    if (QBITS == 64) {
        return reinterpret_cast<const double *>(rp)[i];
    } else if (QBITS == 32) {
        return (double)reinterpret_cast<const float *>(rp)[i];  // float64(Float32frombits)
    } else {
        uint32_t v;
        if (QBITS == 16) {
            v = reinterpret_cast<const uint16_t *>(rp)[i];
        } else if (QBITS == 8) {
            v = rp[i];
        } else {
            const uint8_t b = rp[i >> 1];
            v = (i & 1) ? (b & 0x0Fu) : (b >> 4);  // collection.go:774-779
        }
        const double maxInt = (double)((1u << QBITS) - 1u);
        // (float64(value)/float64(maxInt))*2 - 1, quantization.go:35
        return __dsub_rn(__dmul_rn(__ddiv_rn((double)v, maxInt), 2.0), 1.0);
    }
}

// ordered float64 sum  s = (((0 + p[0]) + p[1]) + ...)  by ONE lane: the order is the reference's, so the adds
// form one dependent chain (768 of them at dim 768) -- what can be taken off that chain is the LDS latency: the
// next eight values are fetched (two per ds_read_b128) while the current eight are being added.  p is 16-byte
// aligned.
__device__ __forceinline__ double ordered_sum(double s, const double *p, int n)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *p2 = reinterpret_cast<const d2 *>(p);
    int i = 0;
    if (n >= 8) {
        d2 a0 = p2[0], a1 = p2[1], a2 = p2[2], a3 = p2[3];
        for (; i + 16 <= n; i += 8) {
            const d2 b0 = p2[(i >> 1) + 4], b1 = p2[(i >> 1) + 5], b2 = p2[(i >> 1) + 6], b3 = p2[(i >> 1) + 7];
            s = __dadd_rn(s, a0.x);
            s = __dadd_rn(s, a0.y);
            s = __dadd_rn(s, a1.x);
            s = __dadd_rn(s, a1.y);
            s = __dadd_rn(s, a2.x);
            s = __dadd_rn(s, a2.y);
            s = __dadd_rn(s, a3.x);
            s = __dadd_rn(s, a3.y);
            a0 = b0;
            a1 = b1;
            a2 = b2;
            a3 = b3;
        }
        s = __dadd_rn(s, a0.x);
        s = __dadd_rn(s, a0.y);
        s = __dadd_rn(s, a1.x);
        s = __dadd_rn(s, a1.y);
        s = __dadd_rn(s, a2.x);
        s = __dadd_rn(s, a2.y);
        s = __dadd_rn(s, a3.x);
        s = __dadd_rn(s, a3.y);
        i += 8;
    }
    for (; i < n; i++) s = __dadd_rn(s, p[i]);
    return s;
}

// elements per LDS chunk of the rerank: at most 384 (9 KiB of products for cosine: 16 candidates per CU at once
// instead of 6 with whole 768-element rows, and the next chunk's gathers overlap this chunk's adds), in multiples of 8
__host__ __device__ inline int rerank_chunk(int dim) { return dim >= 384 ? 384 : ((dim + 7) & ~7); }

// One wave per candidate.  All lanes decode the row and form the per-element
// products (each product is rounded once, exactly as `a*b` in Go); lanes 0..2
// then add them up in index order, one accumulator each, as the loops of
// collection.go:812-832 do.
template <int QBITS, int METRIC>
__global__ __launch_bounds__(64) void rerank_kernel(const uint8_t *rows, RowLayout lay, int dim,
                                                    const double *query, const uint64_t *cands,
                                                    const uint32_t *n_dev, uint32_t n_dev_stride, uint32_t n_max,
                                                    RerankOut *out, const uint32_t *left_rows)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int CH = rerank_chunk(dim);               // elements per LDS chunk (the launch sized smem for it)
    double *p0 = reinterpret_cast<double *>(smem);  // x*y   (euclid: diff*diff)
    double *p1 = p0 + CH;                           // x*x   (cosine only)
    double *p2 = p1 + CH;                           // y*y
    const int lane = threadIdx.x;
    uint32_t n = n_max;
    if (n_dev) n = min(n_dev[(size_t)blockIdx.y * n_dev_stride], n_max);
    query += (size_t)blockIdx.y * dim;           // blockIdx.y = query of the batch
    if (cands) cands += (size_t)blockIdx.y * n_max;
    out += (size_t)blockIdx.y * n_max;
    for (uint32_t ci = blockIdx.x; ci < n; ci += gridDim.x) {
        // cands == nullptr: candidate ci is row ci (full exact replay of a shard)
        const uint64_t c = cands ? cands[ci] : (uint64_t)ci;
        if (c == kInvalidCand) {
            if (lane == 0) {
                RerankOut r;
                r.dist = 0.0;
                r.row = 0xFFFFFFFFu;
                r.ukey = 0xFFFFFFFFu;
                out[ci] = r;
            }
            continue;
        }
        const uint32_t row = (uint32_t)c;
        // left_rows: the "query" of candidate ci is the stored row left_rows[ci], decoded exactly --
        // c.distance(doc1.Vector, doc2.Vector) of computeAverageDistance (collection.go:372-398)
        const bool pair = left_rows != nullptr;
        const uint32_t lrow = pair ? left_rows[ci] : 0u;
        double s = 0.0;
        // Chunks of CH elements: every lane forms the products of its elements (PER = CH / 64 each), lanes 0..2 add
        // them up in index order.  The next chunk's elements are fetched from HBM before the current chunk's sums
        // start, so the gathers' latency hides behind the add chain.
        constexpr int PER = 6;  // CH <= 384
        double xv[PER], yv[PER];
        auto fetch = [&](int base) {
            const int m = min(CH, dim - base);
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = lane + u * 64;
                if (i < m) {
                    xv[u] = pair ? decode_elem<QBITS>(rows, lay, lrow, base + i) : query[base + i];
                    yv[u] = decode_elem<QBITS>(rows, lay, row, base + i);
                }
            }
        };
        fetch(0);
        for (int base = 0; base < dim; base += CH) {
            const int m = min(CH, dim - base);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = lane + u * 64;
                if (i < m) {
                    const double x = xv[u], y = yv[u];
                    if (METRIC == kEuclidean) {
                        const double diff = __dsub_rn(x, y);
                        p0[i] = __dmul_rn(diff, diff);
                    } else {
                        p0[i] = __dmul_rn(x, y);
                        p1[i] = __dmul_rn(x, x);
                        p2[i] = __dmul_rn(y, y);
                    }
                }
            }
            __syncthreads();
            if (base + CH < dim) fetch(base + CH);
            if (METRIC == kEuclidean) {
                if (lane == 0) s = ordered_sum(s, p0, m);
            } else {
                if (lane < 3) s = ordered_sum(s, p0 + (size_t)lane * CH, m);
            }
        }
        const double m1 = __shfl(s, 1);
        const double m2 = __shfl(s, 2);
        if (lane == 0) {
            double dist;
            if (METRIC == kEuclidean) {  // collection.go:812-819
                dist = __dsqrt_rn(s);
            } else {  // collection.go:821-832
                if (m1 == 0 || m2 == 0) {
                    dist = 1.0;
                } else {
                    const double Pi = 3.14159265358979323846264338327950288;
                    const double cosv = __ddiv_rn(s, __dmul_rn(__dsqrt_rn(m1), __dsqrt_rn(m2)));
                    dist = __ddiv_rn(go_acos(cosv), Pi);
                }
            }
            RerankOut r;
            r.dist = dist;
            r.row = row;
            r.ukey = cands ? (uint32_t)(c >> 32) : 0u;
            out[ci] = r;
        }
    }
}

// One LANE per candidate: the form for many candidates per query (a radius batch's hits, a full exact replay).  The
// kernel above gives a candidate a whole wave and spends most of it on three lanes walking the ordered sums -- right
// for the handful of candidates of a top-k query, 118 us for the 46 000 hits of a cfg5-shard radius batch.  Here every
// lane walks its own row piece by piece and keeps the three sums in registers, the same operations in the same order
// (each product rounded once, added in index order), so 64 candidates share a wave's issue slots and a batch's hits
// finish within about one chain's latency.  The query sits in LDS (all lanes read the same element: a broadcast);
// 4- and 8-bit codes decode through a table of the 16 / 256 values the formula of quantization.go:35 gives.
template <int QBITS, int METRIC>
__global__ __launch_bounds__(64) void rerank_lanes_kernel(const uint8_t *rows, RowLayout lay, int dim, const double *query,
                                                          const uint64_t *cands, const uint32_t *n_dev, uint32_t n_dev_stride,
                                                          uint32_t n_max, RerankOut *out)
{
    extern __shared__ __align__(16) uint8_t smem[];
    double *qs = reinterpret_cast<double *>(smem);  // the query
    double *tab = qs + ((dim + 1) & ~1);            // decode table (4- and 8-bit rows)
    constexpr int E = 128 / QBITS;                  // elements per 16-byte piece
    const int lane = threadIdx.x;
    uint32_t n = n_max;
    if (n_dev) n = min(n_dev[(size_t)blockIdx.y * n_dev_stride], n_max);
    if (blockIdx.x * 64u >= n) return;
    query += (size_t)blockIdx.y * dim;
    if (cands) cands += (size_t)blockIdx.y * n_max;
    out += (size_t)blockIdx.y * n_max;
    for (int i = lane; i < dim; i += 64) qs[i] = query[i];
    if (QBITS <= 8) {
        const double maxInt = (double)((1u << QBITS) - 1u);
        for (int v = lane; v < (1 << QBITS); v += 64) tab[v] = __dsub_rn(__dmul_rn(__ddiv_rn((double)v, maxInt), 2.0), 1.0);
    }
    __syncthreads();
    const int pieces = (dim + E - 1) / E;
    for (uint32_t c0 = blockIdx.x * 64u; c0 < n; c0 += gridDim.x * 64u) {
        const uint32_t ci = c0 + lane;
        if (ci >= n) continue;
        const uint64_t c = cands ? cands[ci] : (uint64_t)ci;
        RerankOut r;
        if (c == kInvalidCand) {
            r.dist = 0.0;
            r.row = 0xFFFFFFFFu;
            r.ukey = 0xFFFFFFFFu;
            out[ci] = r;
            continue;
        }
        const uint32_t row = (uint32_t)c;
        double s = 0.0, m1 = 0.0, m2 = 0.0;
        uint4 w = *reinterpret_cast<const uint4 *>(rows + piece_offset(lay, row, 0));
        for (int j = 0; j < pieces; j++) {
            const uint4 cur = w;
            if (j + 1 < pieces) w = *reinterpret_cast<const uint4 *>(rows + piece_offset(lay, row, (uint32_t)(j + 1)));
            const uint32_t ww[4] = {cur.x, cur.y, cur.z, cur.w};
            const int nk = min(E, dim - j * E);
#pragma unroll
            for (int i = 0; i < E; i++) {
                if (i >= nk) break;
                double y;
                if (QBITS == 64) {
                    y = __hiloint2double((int)ww[2 * (i & 1) + 1], (int)ww[2 * (i & 1)]);
                } else if (QBITS == 32) {
                    y = (double)__uint_as_float(ww[i & 3]);
                } else if (QBITS == 16) {
                    const uint32_t v = (ww[(i >> 1) & 3] >> (16 * (i & 1))) & 0xFFFFu;
                    y = __dsub_rn(__dmul_rn(__ddiv_rn((double)v, 65535.0), 2.0), 1.0);
                } else if (QBITS == 8) {
                    y = tab[(ww[(i >> 2) & 3] >> (8 * (i & 3))) & 0xFFu];
                } else {  // byte b holds element 2b in its high nibble, 2b + 1 in the low one (collection.go:774-779)
                    const uint32_t b = (ww[(i >> 3) & 3] >> (8 * ((i >> 1) & 3))) & 0xFFu;
                    y = tab[(i & 1) ? (b & 0x0Fu) : (b >> 4)];
                }
                const double x = qs[j * E + i];
                if (METRIC == kEuclidean) {
                    const double diff = __dsub_rn(x, y);
                    s = __dadd_rn(s, __dmul_rn(diff, diff));
                } else {
                    s = __dadd_rn(s, __dmul_rn(x, y));
                    m1 = __dadd_rn(m1, __dmul_rn(x, x));
                    m2 = __dadd_rn(m2, __dmul_rn(y, y));
                }
            }
        }
        double dist;
        if (METRIC == kEuclidean) {  // collection.go:812-819
            dist = __dsqrt_rn(s);
        } else {  // collection.go:821-832
            if (m1 == 0 || m2 == 0) {
                dist = 1.0;
            } else {
                const double Pi = 3.14159265358979323846264338327950288;
                const double cosv = __ddiv_rn(s, __dmul_rn(__dsqrt_rn(m1), __dsqrt_rn(m2)));
                dist = __ddiv_rn(go_acos(cosv), Pi);
            }
        }
        r.dist = dist;
        r.row = row;
        r.ukey = cands ? (uint32_t)(c >> 32) : 0u;
        out[ci] = r;
    }
}

// ---- synthetic corpus ---------------------------------------------------------

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ double synth_value(uint64_t seed, uint64_t index)
{
    const uint64_t m = splitmix64(seed + index) >> 11;
    return __dsub_rn(__dmul_rn((double)m, 1.0 / 4503599627370496.0), 1.0);
}

// quantization.go:5-23
template <int QBITS>
__device__ __forceinline__ uint64_t quantize(double value)
{
    if (QBITS == 32) return (uint64_t)__float_as_uint((float)value);
    if (QBITS == 64) return (uint64_t)__double_as_longlong(value);
    if (value < -1) value = -1; else if (value > 1) value = 1;
    const double maxInt = (double)((1u << QBITS) - 1u);
    const double q = __dmul_rn(__ddiv_rn(__dadd_rn(value, 1.0), 2.0), maxInt);
    return (uint64_t)round(q);  // half away from zero, like math.Round
}

// one thread per 16-byte piece of the resident row.  src == nullptr: synthetic
// values; otherwise row-major float64 vectors (bulk AddDocument: quantize + pack on
// the device exactly as encodeDocument does, collection.go:713-743)
template <int QBITS>
__global__ void synth_kernel(uint8_t *dst, RowLayout lay, uint64_t dst_first_row, int dim, uint64_t n_rows,
                             uint64_t seed, uint64_t first_row, const double *src)
{
    constexpr int E = 128 / QBITS;
    const uint32_t r16 = lay.pitch / 16;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t row = t / r16;
    const uint32_t j = (uint32_t)(t - row * r16);
    if (row >= n_rows) return;
    uint32_t w[4] = {0, 0, 0, 0};
    const uint64_t ebase = (first_row + row) * (uint64_t)dim;
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int e = (int)j * E + i;
        if (e >= dim) break;
        const double val = src ? src[row * (uint64_t)dim + (uint64_t)e] : synth_value(seed, ebase + (uint64_t)e);
        const uint64_t q = quantize<QBITS>(val);
        if (QBITS == 64) {
            w[2 * i] = (uint32_t)q;
            w[2 * i + 1] = (uint32_t)(q >> 32);
        } else if (QBITS == 32) {
            w[i] = (uint32_t)q;
        } else if (QBITS == 16) {
            w[i >> 1] |= (uint32_t)q << (16 * (i & 1));
        } else if (QBITS == 8) {
            w[i >> 2] |= (uint32_t)q << (8 * (i & 3));
        } else {  // 4-bit: even element in the high nibble of byte i/2
            const int b = i >> 1;
            w[b >> 2] |= (uint32_t)(q & 0xF) << (8 * (b & 3) + ((i & 1) ? 0 : 4));
        }
    }
    *reinterpret_cast<uint4 *>(dst + piece_offset(lay, dst_first_row + row, j)) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- page-in transform --------------------------------------------------------
// Byte b of an element of size es maps to byte es-1-b (big-endian <-> little-
// endian); 4- and 8-bit rows are copied.  One thread per destination byte; the
// aligned 32-bit fast path below covers every bench-sized case.

__global__ void repack_bytes_kernel(uint8_t *ref, uint32_t row_bytes, uint8_t *rows, RowLayout lay,
                                    uint64_t first_row, uint32_t es, uint64_t n_rows, int to_reference)
{   // one thread per byte of the DESTINATION row (pitch bytes resident, row_bytes reference)
    const uint32_t dst_pitch = to_reference ? row_bytes : lay.pitch;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t row = t / dst_pitch;
    const uint32_t b = (uint32_t)(t - row * dst_pitch);
    if (row >= n_rows) return;
    const uint32_t e = b / es, k = b - e * es;
    const uint32_t sb = e * es + (es - 1 - k);  // the mirrored byte on the other side
    if (to_reference) {
        ref[row * row_bytes + b] = rows[piece_offset(lay, first_row + row, sb >> 4) + (sb & 15)];
    } else {
        uint8_t v = 0;
        if (b < row_bytes) v = ref[row * row_bytes + sb];
        rows[piece_offset(lay, first_row + row, b >> 4) + (b & 15)] = v;
    }
}

// row_bytes % 16 == 0 and pitch == row_bytes: 16 bytes per thread
__global__ void repack_vec_kernel(uint4 *ref, uint8_t *rows, RowLayout lay, uint64_t first_row, uint32_t r16,
                                  uint64_t n_vec, uint32_t es, int to_reference)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_vec) return;
    const uint64_t row = t / r16;
    const uint32_t j = (uint32_t)(t - row * r16);
    uint4 *res = reinterpret_cast<uint4 *>(rows + piece_offset(lay, first_row + row, j));
    uint4 v = to_reference ? *res : ref[t];
    if (es == 2) {
        auto sw = [](uint32_t x) { return ((x & 0x00FF00FFu) << 8) | ((x >> 8) & 0x00FF00FFu); };
        v = make_uint4(sw(v.x), sw(v.y), sw(v.z), sw(v.w));
    } else if (es == 4) {
        v = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z),
                       __builtin_bswap32(v.w));
    } else if (es == 8) {
        v = make_uint4(__builtin_bswap32(v.y), __builtin_bswap32(v.x), __builtin_bswap32(v.w),
                       __builtin_bswap32(v.z));
    }
    if (to_reference)
        ref[t] = v;
    else
        *res = v;
}

// ---- 8-bit sketch of float32 rows (cosine) --------------------------------------
// One wave per row.  x_i / max|x| is quantized to the library's own 8-bit element form (byte v, value
// n = 2v - 255: odd integers; cosine does not see the scale), written in the sketch shard's resident
// layout.  The angular distance between the row and its sketch -- the reference's own metric,
// acos(cos)/pi, in float64 -- goes into a running maximum: by the triangle inequality on the sphere
// |d(q, x) - d(q, sketch)| <= d(x, sketch) for every query.  Rows without a direction (all zero) or with
// a non-finite element get a dummy sketch and are reported (they are always re-ranked).
// gscale > 0 (Euclidean collections): every row is scaled by the same 1 / gscale (gscale >= max |x_i| over the
// collection), the sketch stands for gscale * n / 255, and the running maximum is the Euclidean distance
// row <-> sketch; gscale == 0 (cosine): per-row scale, angular distance.  With max_only the kernel just
// reports the largest finite |x_i| of the rows (bits of a float) through max_ang and writes nothing.
__global__ __launch_bounds__(256) void sketch_build_kernel(const uint8_t *src, RowLayout src_lay, int dim, uint8_t *dst,
                                                           RowLayout dst_lay, uint64_t first_row, uint64_t n_rows,
                                                           const uint32_t *row_list, unsigned long long *max_ang,
                                                           uint32_t *exc_rows, uint32_t *exc_count, uint32_t exc_cap,
                                                           double gscale, int max_only)
{
    const int lane = threadIdx.x & 63;
    const uint64_t i = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const uint64_t row = row_list ? (uint64_t)row_list[i] : first_row + i;
    const int pieces = (dim + 15) / 16;  // 16-element pieces of the sketch row
    float mx = 0.f;
    bool bad = false;
    for (int p = lane; p < pieces; p += 64) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int e0 = p * 16 + t * 4;
            if (e0 >= dim) break;
            const float4 v = *reinterpret_cast<const float4 *>(src + piece_offset(src_lay, row, (uint32_t)(p * 4 + t)));
            const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (e0 + k >= dim) continue;
                const float ax = fabsf(x[k]);
                if (!(ax <= 3.4e38f)) bad = true;  // Inf or NaN
                mx = fmaxf(mx, ax);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    bad = __ballot(bad) != 0;
    if (max_only) {
        if (lane == 0 && !bad) atomicMax(max_ang, (unsigned long long)__float_as_uint(mx));
        return;
    }
    if (gscale > 0.0) {
        if ((double)mx > gscale) bad = true;  // (cannot happen after a max pass; such a row would be re-ranked always)
    } else {
        bad = bad || !(mx > 0.f);
    }
    const double inv = bad ? 0.0 : 1.0 / (gscale > 0.0 ? gscale : (double)mx);
    double dot = 0.0, nx = 0.0, nn = 0.0;
    for (int p = lane; p < pieces; p += 64) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int e0 = p * 16 + t * 4;
            if (e0 >= dim) break;
            const float4 v = *reinterpret_cast<const float4 *>(src + piece_offset(src_lay, row, (uint32_t)(p * 4 + t)));
            const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (e0 + k >= dim) continue;  // padding bytes stay 0, as the page-in leaves them
                int q = 128;
                if (!bad) {
                    q = (int)rint(((double)x[k] * inv + 1.0) * 127.5);
                    q = q < 0 ? 0 : (q > 255 ? 255 : q);
                    const double n = (double)(2 * q - 255), xd = (double)x[k];
                    if (gscale > 0.0) {
                        const double df = xd - gscale * n / 255.0;
                        dot += df * df;
                    } else {
                        dot += xd * n;
                        nx += xd * xd;
                        nn += n * n;
                    }
                }
                w[t] |= (uint32_t)q << (8 * k);
            }
        }
        *reinterpret_cast<uint4 *>(dst + piece_offset(dst_lay, row, (uint32_t)p)) = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dot += __shfl_xor(dot, o);
        nx += __shfl_xor(nx, o);
        nn += __shfl_xor(nn, o);
    }
    if (lane == 0) {
        if (bad) {
            const uint32_t at = atomicAdd(exc_count, 1u);
            if (at < exc_cap) exc_rows[at] = (uint32_t)row;
        } else if (gscale > 0.0) {
            atomicMax(max_ang, (unsigned long long)__double_as_longlong(sqrt(dot)));
        } else {
            double c = dot / sqrt(nx * nn);
            c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
            const double ang = acos(c) / 3.14159265358979323846264338327950288;
            atomicMax(max_ang, (unsigned long long)__double_as_longlong(ang));  // non-negative doubles order as integers
        }
    }
}

__global__ void fill_bits_kernel(uint64_t *bits, uint64_t n_rows, uint64_t n_words)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_words) return;
    const uint64_t lo = t * 64;
    uint64_t v = 0;
    if (lo + 64 <= n_rows) v = ~0ull;
    else if (lo < n_rows) v = (1ull << (n_rows - lo)) - 1ull;
    bits[t] = v;
}

// device-side float64 primitive probe (tests: bit-exactness of div/sqrt/acos)
__global__ void f64_probe_kernel(int op, const double *a, const double *b, double *out, uint64_t n)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double r;
    switch (op) {
    case 0: r = __ddiv_rn(a[t], b[t]); break;
    case 1: r = __dsqrt_rn(a[t]); break;
    case 2: r = go_acos(a[t]); break;
    case 3: r = round(a[t]); break;
    default: r = (double)(float)a[t]; break;
    }
    out[t] = r;
}

template <int QBITS>
hipError_t launch_rerank_q(int metric, const uint8_t *rows, RowLayout lay, int dim,
                           const double *q, const uint64_t *cands, const uint32_t *n_dev,
                           uint32_t n_max, int n_queries, RerankOut *out, hipStream_t stream,
                           const uint32_t *left_rows = nullptr, uint32_t n_dev_stride = 0)
{
    if (n_max == 0 || n_queries == 0) return hipSuccess;
    if (!left_rows && n_max >= 512u) {  // many candidates per query: one lane each
        const dim3 grid(std::min((n_max + 63u) / 64u, 4096u), n_queries);
        const size_t lds = ((size_t)((dim + 1) & ~1) + (QBITS <= 8 ? (size_t)1 << QBITS : 0)) * sizeof(double);
        if (metric == kCosine)
            hipLaunchKernelGGL((rerank_lanes_kernel<QBITS, kCosine>), grid, dim3(64), lds, stream, rows, lay, dim, q, cands, n_dev,
                               n_dev_stride, n_max, out);
        else
            hipLaunchKernelGGL((rerank_lanes_kernel<QBITS, kEuclidean>), grid, dim3(64), lds, stream, rows, lay, dim, q, cands,
                               n_dev, n_dev_stride, n_max, out);
        return hipGetLastError();
    }
    const dim3 grid(n_max < 4096u ? n_max : 4096u, n_queries);
    const size_t lds = (size_t)rerank_chunk(dim) * (metric == kCosine ? 3 : 1) * sizeof(double);
    if (metric == kCosine)
        hipLaunchKernelGGL((rerank_kernel<QBITS, kCosine>), grid, dim3(64), lds, stream, rows,
                           lay, dim, q, cands, n_dev, n_dev_stride, n_max, out, left_rows);
    else
        hipLaunchKernelGGL((rerank_kernel<QBITS, kEuclidean>), grid, dim3(64), lds, stream,
                           rows, lay, dim, q, cands, n_dev, n_dev_stride, n_max, out, left_rows);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_rerank(int qbits, int metric, const uint8_t *rows, RowLayout pitch, int dim,
                         const double *q, const uint64_t *cands, const uint32_t *n_dev,
                         uint32_t n_max, int n_queries, RerankOut *out, hipStream_t stream, uint32_t n_dev_stride)
{
    switch (qbits) {
    case 4: return launch_rerank_q<4>(metric, rows, pitch, dim, q, cands, n_dev, n_max, n_queries, out, stream, nullptr, n_dev_stride);
    case 8: return launch_rerank_q<8>(metric, rows, pitch, dim, q, cands, n_dev, n_max, n_queries, out, stream, nullptr, n_dev_stride);
    case 16: return launch_rerank_q<16>(metric, rows, pitch, dim, q, cands, n_dev, n_max, n_queries, out, stream, nullptr, n_dev_stride);
    case 32: return launch_rerank_q<32>(metric, rows, pitch, dim, q, cands, n_dev, n_max, n_queries, out, stream, nullptr, n_dev_stride);
    case 64: return launch_rerank_q<64>(metric, rows, pitch, dim, q, cands, n_dev, n_max, n_queries, out, stream, nullptr, n_dev_stride);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rerank_pairs(int qbits, int metric, const uint8_t *rows, RowLayout pitch, int dim,
                               const uint32_t *left_rows, const uint64_t *right_cands, uint32_t n_pairs,
                               RerankOut *out, hipStream_t stream)
{
    switch (qbits) {
    case 4: return launch_rerank_q<4>(metric, rows, pitch, dim, nullptr, right_cands, nullptr, n_pairs, 1, out, stream, left_rows);
    case 8: return launch_rerank_q<8>(metric, rows, pitch, dim, nullptr, right_cands, nullptr, n_pairs, 1, out, stream, left_rows);
    case 16: return launch_rerank_q<16>(metric, rows, pitch, dim, nullptr, right_cands, nullptr, n_pairs, 1, out, stream, left_rows);
    case 32: return launch_rerank_q<32>(metric, rows, pitch, dim, nullptr, right_cands, nullptr, n_pairs, 1, out, stream, left_rows);
    case 64: return launch_rerank_q<64>(metric, rows, pitch, dim, nullptr, right_cands, nullptr, n_pairs, 1, out, stream, left_rows);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sketch_build(const uint8_t *src, RowLayout src_lay, int dim, uint8_t *dst, RowLayout dst_lay,
                               uint64_t first_row, uint64_t n_rows, const uint32_t *row_list,
                               unsigned long long *max_ang, uint32_t *exc_rows, uint32_t *exc_count, uint32_t exc_cap,
                               double gscale, int max_only, hipStream_t stream)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t grid = (n_rows + 3) / 4;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sketch_build_kernel, dim3((unsigned)grid), dim3(256), 0, stream, src, src_lay, dim, dst, dst_lay,
                       first_row, n_rows, row_list, max_ang, exc_rows, exc_count, exc_cap, gscale, max_only);
    return hipGetLastError();
}

hipError_t launch_repack(int qbits, uint8_t *ref, uint32_t row_bytes, uint8_t *rows, RowLayout lay,
                         uint64_t first_row, uint64_t n_rows, int to_reference, hipStream_t stream)
{
    if (n_rows == 0) return hipSuccess;
    const uint32_t es = qbits <= 8 ? 1u : (uint32_t)qbits / 8u;
    if (row_bytes == lay.pitch && (row_bytes % 16) == 0) {
        const uint32_t r16 = row_bytes / 16;
        const uint64_t n_vec = n_rows * r16;
        const uint64_t grid = (n_vec + 255) / 256;
        hipLaunchKernelGGL(repack_vec_kernel, dim3((unsigned)grid), dim3(256), 0, stream,
                           reinterpret_cast<uint4 *>(ref), rows, lay, first_row, r16, n_vec, es, to_reference);
        return hipGetLastError();
    }
    const uint64_t total = n_rows * (to_reference ? row_bytes : lay.pitch);
    const uint64_t grid = (total + 255) / 256;
    hipLaunchKernelGGL(repack_bytes_kernel, dim3((unsigned)grid), dim3(256), 0, stream, ref, row_bytes, rows,
                       lay, first_row, es, n_rows, to_reference);
    return hipGetLastError();
}

hipError_t launch_synth(int qbits, uint8_t *rows, RowLayout lay, uint64_t dst_first_row, int dim,
                        uint64_t n_rows, uint64_t seed, uint64_t first_row, const double *src,
                        hipStream_t stream)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t total = n_rows * (lay.pitch / 16);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    switch (qbits) {
    case 4: hipLaunchKernelGGL(synth_kernel<4>, g, b, 0, stream, rows, lay, dst_first_row, dim, n_rows, seed, first_row, src); break;
    case 8: hipLaunchKernelGGL(synth_kernel<8>, g, b, 0, stream, rows, lay, dst_first_row, dim, n_rows, seed, first_row, src); break;
    case 16: hipLaunchKernelGGL(synth_kernel<16>, g, b, 0, stream, rows, lay, dst_first_row, dim, n_rows, seed, first_row, src); break;
    case 32: hipLaunchKernelGGL(synth_kernel<32>, g, b, 0, stream, rows, lay, dst_first_row, dim, n_rows, seed, first_row, src); break;
    case 64: hipLaunchKernelGGL(synth_kernel<64>, g, b, 0, stream, rows, lay, dst_first_row, dim, n_rows, seed, first_row, src); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fill_bits(uint64_t *bits, uint64_t n_rows, uint64_t n_words, hipStream_t stream)
{
    if (n_words == 0) return hipSuccess;
    const uint64_t grid = (n_words + 255) / 256;
    hipLaunchKernelGGL(fill_bits_kernel, dim3((unsigned)grid), dim3(256), 0, stream, bits, n_rows,
                       n_words);
    return hipGetLastError();
}

hipError_t launch_f64_probe(int op, const double *a, const double *b, double *out, uint64_t n,
                            hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t grid = (n + 255) / 256;
    hipLaunchKernelGGL(f64_probe_kernel, dim3((unsigned)grid), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}


// ---- a radius batch's hits, sorted on the device ------------------------------------------------------------------
// The host assembles a radius result as the hits sorted by distance (scan_radius.cpp: radius_assemble); sorting
// ~500 doubles costs it ~13 us per query, which is what a shared-sweep radius batch is bound by.  One block per
// query sorts the query's re-ranked hits by distance in LDS (bitonic over (ordered distance, index) pairs, the
// entries themselves staged beside them) -- lists of 2 .. kSortHitsMax entries; longer ones are left to the host.
// Not stable: equal distances are the host's business anyway (it replays the reference's heap for them).
namespace {
__device__ __forceinline__ uint64_t ordered_f64(double d)
{
    const uint64_t b = (uint64_t)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ __launch_bounds__(256) void sort_hits_kernel(RerankOut *out, const uint32_t *count, uint32_t count_stride,
                                                        uint32_t cap)
{
    __shared__ RerankOut ent[kSortHitsMax];
    __shared__ uint64_t key[kSortHitsMax];
    __shared__ uint16_t idx[kSortHitsMax];
    const uint32_t n = min(count[(size_t)blockIdx.x * count_stride], cap);
    if (n < 2 || n > (uint32_t)kSortHitsMax) return;
    uint32_t P = 2;
    while (P < n) P <<= 1;
    RerankOut *o = out + (size_t)blockIdx.x * cap;
    for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
        if (i < n) {
            ent[i] = o[i];
            // (strictly below the padding key: a NaN with an all-ones payload orders as ~0 too, and a padding index
            // sorted in front of it would copy an entry that was never staged)
            const uint64_t ok_ = ordered_f64(ent[i].dist);
            key[i] = ok_ == ~0ull ? ~0ull - 1ull : ok_;  // (no min(): its overloads take uint64_t through double)
        } else {
            key[i] = ~0ull;
        }
        idx[i] = (uint16_t)i;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const uint64_t a = key[i], b = key[l];
                    if ((a > b) == up) {
                        key[i] = b;
                        key[l] = a;
                        const uint16_t t = idx[i];
                        idx[i] = idx[l];
                        idx[l] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) o[i] = ent[idx[i]];
}
}  // namespace

// ---- resident row norms (16-bit rows) ---------------------------------------------------------------------------
// 16 lanes per row, 16 rows per block of 256 threads; every lane sums the squares of its 16-byte pieces' real
// elements (the padding codes of the last piece stay out, as in the sweeps), the 16 partial sums are added with DPP.
namespace {
__global__ __launch_bounds__(256) void row_norms16_kernel(const uint8_t *rows, uint32_t pitch, int dim, uint64_t first_row,
                                                          uint64_t n_rows, float *out)
{
    const uint64_t r = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    float s = 0.f;
    if (r < n_rows) {
        const uint8_t *rp = rows + (first_row + r) * (uint64_t)pitch;
        const int pieces = (dim + 7) / 8;
        for (int i = l; i < pieces; i += 16) {
            const uint4 w = reinterpret_cast<const uint4 *>(rp)[i];
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
            const int nk = min(8, dim - i * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t v = (ww[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                const float x = fmaf((float)v, 2.0f, -65535.0f);
                s = e < nk ? fmaf(x, x, s) : s;
            }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (r < n_rows && l == 0) out[first_row + r] = s;
}
}  // namespace

// 8- and 4-bit rows (either layout): the int8 sweeps' norm, formed exactly as they form it -- the integer
// 4 * sum (x'^2 + x') over every code of the pitched row (x' = v - 128, resp. nibble - 8), converted once, plus the
// handle's bias for the padding codes.  4 lanes per row (16 rows per wave: a tiled KiB is read in order).
namespace {
template <int RB>
__global__ __launch_bounds__(256) void row_norms_i8_kernel(const uint8_t *rows, RowLayout lay, int r16, float norm_bias,
                                                           uint64_t first_row, uint64_t n_rows, float *out)
{
    const uint64_t r = (uint64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int c = threadIdx.x & 3;
    int SQ = 0, SV = 0;
    if (r < n_rows) {
        for (int j = c; j < r16; j += 4) {
            const uint4 w = *reinterpret_cast<const uint4 *>(rows + piece_offset(lay, first_row + r, (uint32_t)j));
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (RB == 8) {
                    const int wn = (int)(ww[d] ^ 0x80808080u);
                    SQ = __builtin_amdgcn_sdot4(wn, wn, SQ, false);
                    SV = __builtin_amdgcn_sdot4(wn, 0x01010101, SV, false);
                } else {
                    const int wn = (int)(ww[d] ^ 0x88888888u);
                    SQ = __builtin_amdgcn_sdot8(wn, wn, SQ, false);
                    SV = __builtin_amdgcn_sdot8(wn, 0x11111111, SV, false);
                }
            }
        }
    }
    int nrm = 4 * (SQ + SV);
    nrm += __shfl_xor(nrm, 1);
    nrm += __shfl_xor(nrm, 2);
    if (r < n_rows && c == 0) out[first_row + r] = (float)nrm + norm_bias;
}
}  // namespace

hipError_t launch_row_norms(int bits, const uint8_t *rows, const RowLayout &lay, int dim, float norm_bias, uint64_t first_row,
                            uint64_t n_rows, float *out, hipStream_t stream)
{
    if (n_rows == 0) return hipSuccess;
    if (bits == 16 && !lay.tiled) {
        hipLaunchKernelGGL(row_norms16_kernel, dim3((unsigned)((n_rows + 15) / 16)), dim3(256), 0, stream, rows, lay.pitch, dim,
                           first_row, n_rows, out);
    } else if (bits == 8 || bits == 4) {
        const int r16 = (int)(lay.pitch / 16);
        const dim3 grid((unsigned)((n_rows + 63) / 64));
        if (bits == 8)
            hipLaunchKernelGGL(row_norms_i8_kernel<8>, grid, dim3(256), 0, stream, rows, lay, r16, norm_bias, first_row, n_rows, out);
        else
            hipLaunchKernelGGL(row_norms_i8_kernel<4>, grid, dim3(256), 0, stream, rows, lay, r16, norm_bias, first_row, n_rows, out);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_sort_hits(RerankOut *out, const uint32_t *count, uint32_t count_stride, uint32_t cap, int n_queries,
                            hipStream_t stream)
{
    if (n_queries <= 0) return hipSuccess;
    hipLaunchKernelGGL(sort_hits_kernel, dim3(n_queries), dim3(256), 0, stream, out, count, count_stride, cap);
    return hipGetLastError();
}

}  // namespace szg
