// kernels.h -- launch-side declarations shared by the HIP kernel files and the
// C-ABI implementation (scan_api.cpp).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace szg {

constexpr int kEuclidean = 0;  // collection.go:186-189
constexpr int kCosine = 1;

// A candidate is one packed 64-bit word: (ordered key << 32) | local row.
// Unsigned comparison of the word orders by key, ties by lower row.
constexpr uint64_t kInvalidCand = 0xFFFFFFFFFFFFFFFFull;

// float -> uint32 so that unsigned order == float order (NaN sorts last).
__host__ __device__ inline uint32_t ordered_key(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float key_from_ordered(uint32_t u)
{
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// How the 64 lanes of a wave are laid over rows of r16 16-byte pieces:
// groups of L lanes take one row each, every lane P pieces of it.
// Where a row's 16-byte pieces live in the resident mirror.
//   linear: row-major, `pitch` bytes per row.
//   tiled : tiles of 16 rows; inside a tile the 64-byte step s of all 16 rows is one
//           contiguous KiB ([step][row & 15][piece & 3]).  A wave instruction that reads one
//           64-byte step of 16 rows -- the walk of short rows and the MFMA operand layout of
//           the shared sweeps -- then reads 1 KiB contiguous instead of sixteen 64-byte
//           segments (a bare read of which tops out at 6.2 TB/s, scripts/readbw).  Needs
//           pitch % 64 == 0; rows are allocated in multiples of 16.
struct RowLayout {
    uint32_t pitch;
    uint32_t tiled;
    uint32_t steps;  // pitch / 64 when tiled
};
__host__ __device__ inline uint64_t piece_offset(const RowLayout &l, uint64_t row, uint32_t j)
{
    if (!l.tiled) return row * l.pitch + (uint64_t)j * 16;
    return (row >> 4) * ((uint64_t)l.steps * 1024) + (uint64_t)(j >> 2) * 1024 + (row & 15) * 64 + (j & 3) * 16;
}
__host__ __device__ inline uint64_t layout_bytes(const RowLayout &l, uint64_t rows)
{
    return l.tiled ? ((rows + 15) >> 4) * ((uint64_t)l.steps * 1024) : rows * (uint64_t)l.pitch;
}

struct RowMap {
    int r16;  // 16-byte pieces per (pitched) row
    int L;    // lanes per row
    int P;    // pieces per lane per row (L*P >= r16)
    int gpw;  // rows (lane groups) per wave iteration
    int pow2; // L is a power of two (groups are aligned: DPP reductions apply)
    int dense; // L*P == r16 and gpw*L == 64: every lane always has a piece of a row
};

// 4-bit rows, single-query scan: the query is quantized to balanced int4 digit planes of radix
// 16.  kPlanes4 digits carry |Q| < 16^kPlanes4 / 2; every plane costs one v_dot8_i32_i4 per
// dword of the row and one 16-byte LDS read per piece.  The quantization step enters the
// certification bound (key_eps) -- 4 planes (15 bits) still certify: the bound grows to
// ~4e-5 in cosine units against gaps of ~1e-2 between the k-th and the kp-th best key.
#ifndef SZG_PLANES4
#define SZG_PLANES4 4
#endif
constexpr int kPlanes4 = SZG_PLANES4;
constexpr double kQmax4 = kPlanes4 == 5 ? 480000.0 : 30000.0;

// Bytes of one prepared query as the scan stages it in LDS: float32 (float64 for
// 64-bit rows) per element, or integer digit planes of 16 bytes per piece for the
// exact-integer paths (3 int8 planes for 8-bit rows, 5 int4 planes for 4-bit rows).
__host__ __device__ inline size_t query_lds_bytes(int qbits, int r16)
{
    switch (qbits) {
    case 4: return (size_t)r16 * 16 * kPlanes4;
    case 8: return (size_t)r16 * 16 * 3;
    case 16: return (size_t)r16 * 8 * 4;
    case 32: return (size_t)r16 * 4 * 4;
    default: return (size_t)r16 * 2 * 8;
    }
}

constexpr int kMaxSweepsPerLaunch = 16;

// Hit counters of the fused selection sit one per 128-byte line: the waves claim their slots
// with returning atomics, mostly at the end of the sweep, and atomics on one line serialise
// (~12 ns each) -- 48 counters in two lines made a 46 us tail on a 190 us sweep.
constexpr int kCandCountStride = 32;  // 32-bit words

// Exact integer shared sweep (mq_score_i8_kernel): the query as kMqPlanes balanced int8 digit
// planes of radix 128.  Every plane costs one MFMA and one 16-byte LDS read per query block
// and 64-byte row step; 2 planes (|Q| <= 16000, ~14 bits) keep the certification bound at
// ~2e-4 in cosine units, still far inside the gap between the k-th and the kp-th best key.
#ifndef SZG_MQ_PLANES
#define SZG_MQ_PLANES 2
#endif
constexpr int kMqPlanes = SZG_MQ_PLANES;
constexpr double kMqQmax = kMqPlanes == 2 ? 16000.0 : 1000000.0;

// per-query constants of the integer paths, as the row finish consumes them
struct QConst {
    float qscale, qconst, qnorm2, norm_bias;
};

struct ScanArgs {
    const uint8_t *rows;        // resident mirror: n_rows x pitch bytes, little-endian elements
    uint32_t n_rows;
    uint32_t pitch;             // bytes, multiple of 16
    uint32_t tiled, steps;      // RowLayout of `rows`
    int dim;
    RowMap map;
    const uint64_t *live_bits;  // nullable: bit r == 0 -> tombstoned
    const uint64_t *allow_bits; // nullable: bit r == 0 -> filtered out (collection.go:592)
    const void *query;          // device, piece-swizzled (see prep_query in scan_api.cpp)
    int n_queries;              // queries walked back to back by one launch
    uint32_t query_stride;      // bytes between consecutive swizzled queries
    uint32_t allow_stride;      // words between consecutive queries' allow masks
    // integer paths (4/8-bit rows): query ~ qscale * Q, qconst = sum Q, norm_bias turns
    // the row's integer sums into sum n^2 without the padding; qnorm2 = sum g^2 (euclid)
    float qscale[kMaxSweepsPerLaunch], qconst[kMaxSweepsPerLaunch], qnorm2[kMaxSweepsPerLaunch];
    double norm_bias;
    int no_shape_kernels;       // tuning hook: always take the any-shape kernel
    int ring;                   // tuning hook: >= 8 forces the deep ring (0 = chosen from kp)
    int mask_dense;             // masked sweep that reads every row and applies the masks at the row finish
                                // (most rows pass); 0 = rows are tested before their loads are issued
    int kp;                     // candidates kept per list (top-k mode)
    uint64_t *block_lists;      // [n_queries][grid][kp] sorted ascending (top-k mode)
    // collect mode (radius search / escalation): every row of sweep s with key <= thr_ukeys[s] is appended to
    // collect_buf + s * collect_cap; the sweep's hit counter is collect_count[s * kCandCountStride] (it may
    // exceed collect_cap: the caller then reruns that query with a larger buffer)
    int collect;
    uint32_t thr_ukeys[kMaxSweepsPerLaunch];
    uint64_t *collect_buf;
    uint32_t collect_cap;
    uint32_t *collect_count;
};

// Fused dequantize + distance + select.  QBITS in {4,8,16,32,64}.
hipError_t launch_scan(int qbits, int metric, const ScanArgs &a, int grid, int block,
                       hipStream_t stream);
// LDS bytes launch_scan needs for (qbits, map, kp, block)
size_t scan_lds_bytes(int qbits, const RowMap &m, int kp, int block);

// Merge n_lists sorted lists of kp candidates into ceil(n_lists/merge_fan(kp)) lists.
int merge_fan(int kp);
// ---- multi-query sweep (kernels_mq.hip): 4/8/16/32-bit rows, cosine -------------
// queries per shared sweep: 48 (3 blocks of 16) for the float32 and int8 sweeps, whose LDS images are
// 4 and 2-4 bytes per element and query; 96 for the bfloat16 sweep (2 bytes)
constexpr int kMqMaxQueries = 96;
struct MqArgs {
    const uint8_t *rows;      // resident mirror
    uint32_t n_rows;
    uint32_t pitch;
    uint32_t tiled, steps;    // RowLayout of `rows`
    int r16;                  // 16-byte pieces per row
    int dim;
    const void *queries;      // device: LDS image [piece][query block][group of 4][16 queries][4 floats]
    int n_queries;            // <= 16 * query blocks
    int metric;               // kCosine: image = q/|q|, key = -cos.  kEuclidean: image = the scan's
                              // prepared query (maxInt*q for quantized rows), key = |n - image|^2
    float qnorm2[kMqMaxQueries];  // euclid: |image_q|^2 per query
    float qsum[kMqMaxQueries];    // 8-bit rows through the bfloat16 sweep: the sum of the query's (rounded) image values
    float norm_bias;          // integer sweep: turns 4*(sum v'^2 + sum v') into sum n^2 of the real elements
    float *keys;              // out: [n_queries][key_stride] ranking keys
    size_t key_stride;        // floats, multiple of 4, >= n_rows
    const uint8_t *zero16;    // 16 zero bytes (address idle lanes read)
    // fused selection (collect != 0): instead of writing the score matrix, every (query,
    // row) whose key is <= thr[query] and whose mask bits allow it is appended to the
    // query's candidate buffer; thr comes from a sweep of a prefix of the rows
    int n_groups;                // int8 sweep: query groups of 48 one launch walks (0 / 1: one); image g at
    uint32_t group_stride;       // queries + g * group_stride bytes, thr / keys / candidates indexed by 48 g + q
    int shape_kernels;           // int8 sweep: take the row-shape-specialised kernel where one exists
    int collect;
    const float *thr;            // [n_queries]
    uint64_t *cand_buf;          // [n_queries][cand_cap]  (ordered key << 32 | row)
    uint32_t *cand_count;        // [n_queries * kCandCountStride]; may exceed cand_cap (then the batch is redone)
    uint32_t cand_cap;
    const uint64_t *live_bits;   // nullable
    const uint64_t *allow_bits;  // nullable, per query
    uint32_t allow_stride;
    const float *row_norm;       // resident row norms (launch_row_norms): 16-bit rows (nullable: the direct bfloat16
                                 // sweep then sums its own) and 8-/4-bit rows (the shape kernels REQUIRE them; without,
                                 // the any-shape kernel runs)
};
// thr[q] = key of the kp-th entry of query q's sorted list (3.0e38 if the list is shorter)
hipError_t launch_mq_thr(const uint64_t *lists, int kp, int n_queries, float *thr, hipStream_t stream);
// per query: the kp best of its candidate buffer, sorted, as one list [n_queries][kp]
hipError_t launch_cand_select(const uint64_t *cand_buf, const uint32_t *cand_count, uint32_t cand_cap,
                              int kp, int n_queries, uint64_t *lists, hipStream_t stream);
// The tail of a fused-selection batch in one launch (kernels_mq.hip: cand_refine_kernel): the kp best of each query's
// collected candidates -- mode 0: by the collected key; 1 / 2 (bfloat16 sweeps, cosine / euclid): the candidates
// within the sweep's error band of the kp-th best are scored again in float32 first, band_edge[q] tells the host
// where the band ended -- followed by the query's n_sent sentinel rows: lists [n_queries][kp + n_sent].  A band that
// does not fit sets cand_count[q] to 0xFFFFFFFF (the batch is redone through the score matrix).
bool cand_refine_applies(int kp, uint32_t cand_cap, int dim, bool rescore);
hipError_t launch_cand_refine(int mode, const uint8_t *rows, RowLayout layout, int dim, const double *q64,
                              const double *qscale, const double *qnorm2, const uint64_t *cand_buf, uint32_t *cand_count,
                              uint32_t cand_cap, int kp, int n_queries, const uint64_t *sent, int n_sent,
                              uint64_t *lists, float *band_edge, int row_bits, hipStream_t stream);
// float32 re-score of the collected candidates of a bfloat16 sweep (32-bit rows, or 16-bit rows of whole 16-byte
// pieces, decoded to n = 2v - 65535): replaces the key in every candidate word; qscale[q] = 1/|q| (cosine) or the
// prepared query's scale (euclid: 1, or 65535 for 16-bit rows), the float32 query is (float)(q64 * qscale)
hipError_t launch_cand_rescore(int metric, const uint8_t *rows, RowLayout layout, int dim, const double *q64,
                               const double *qscale, uint64_t *cand_buf, const uint32_t *cand_count,
                               uint32_t cand_cap, int n_queries, int row_bits, hipStream_t stream);

// Exact integer shared sweep for 8-bit rows (v_mfma_i32_16x16x64_i8).  MqArgs.queries is
// the image [64-byte step][digit plane h,m,l][query block][lane = chunk*16 + query][16 bytes]
// of the queries' balanced int8 digit planes (prep_query), followed by the float table
// [qscale | qconst | qnorm2][48]; MqArgs.norm_bias as ScanArgs.norm_bias.
// row_bits = 4: two B operands per piece (high / low nibbles as unsigned bytes), the image
// is [step][plane][even, odd elements][query block][lane][16 bytes] and the table's qconst
// entries hold -15 * sum Q (n = 2x - 15).
size_t mq_i8_image_bytes(int row_bits, int r16, int nb);   // digit image only
size_t mq_i8_lds_bytes(int row_bits, int r16, int nb, int groups = 1);  // groups x (image + constants + thresholds) + hit buffers
hipError_t launch_mq_score_i8(int row_bits, const MqArgs &a, int nb, int grid, hipStream_t stream);
// bfloat16 shared sweep for 32-bit rows of whole 64-byte steps (v_mfma_f32_16x16x32_bf16): MqArgs.queries is the
// image [32-element step][query block][lane = k-group*16 + query][8 bf16 = elements 8*k-group + 0..7 of the step]
// (zeros where the row has ended).
size_t mq_bf16_image_bytes(int row_bits, int r16, int nb);
size_t mq_bf16_lds_bytes(int row_bits, int r16, int nb);
hipError_t launch_mq_score_bf16(int row_bits, const MqArgs &a, int nb, int grid, hipStream_t stream);  // 64-, 32- or 16-bit rows
hipError_t launch_mq_select(const float *keys, size_t key_stride, uint32_t n_rows,
                            const uint64_t *live_bits, const uint64_t *allow_bits,
                            uint32_t allow_stride, int kp, int n_queries, int blocks_per_query,
                            uint64_t *block_lists, hipStream_t stream, float *thr_out = nullptr,
                            uint32_t *count_zero = nullptr);  // thr_out: one block per query; also
                                                              // writes thr[q] and zeroes count_zero[q]

// Lists are [n_queries][n_lists][kp]; the output is [n_queries][n_out][kp].
hipError_t launch_merge(const uint64_t *in, int n_lists, int kp, int n_queries, uint64_t *out,
                        hipStream_t stream);

struct RerankOut {
    double dist;    // the reference's float64 distance (collection.go:812-832)
    uint32_t row;   // local row
    uint32_t ukey;  // the scan's ordered key for that row
};

// Exact float64 distances, reference operation order, for n candidates of each of
// n_queries queries: query_f64 [n_queries][dim], cands/out [n_queries][n_cands_max].
// n_cands_dev (nullable): the candidate count of query q is min(n_cands_dev[q * n_dev_stride], n_cands_max), read on
// the device (collect sweeps: the hit counters, no host round trip between the sweep and its re-rank).
// sorts each query's re-ranked hits out[q * cap .. + min(count[q * count_stride], cap)) by distance, ascending, when the
// list has 2 .. kSortHitsMax entries (longer lists are left as they are)
constexpr int kSortHitsMax = 2048;
hipError_t launch_sort_hits(RerankOut *out, const uint32_t *count, uint32_t count_stride, uint32_t cap, int n_queries,
                            hipStream_t stream);
hipError_t launch_rerank(int qbits, int metric, const uint8_t *rows, RowLayout layout, int dim,
                         const double *query_f64, const uint64_t *cands, const uint32_t *n_cands_dev,
                         uint32_t n_cands_max, int n_queries, RerankOut *out, hipStream_t stream,
                         uint32_t n_dev_stride = 0);

// The same for pairs of STORED rows (computeAverageDistance, collection.go:372-398): out[i] =
// c.distance(row left_rows[i], row (uint32)right_cands[i]), both decoded exactly on the device.
hipError_t launch_rerank_pairs(int qbits, int metric, const uint8_t *rows, RowLayout layout, int dim,
                               const uint32_t *left_rows, const uint64_t *right_cands, uint32_t n_pairs,
                               RerankOut *out, hipStream_t stream);

// Page-in transform: n_rows rows in the reference encoding at `ref` (big-endian 16/32/64-bit,
// row_bytes apart) <-> rows [first_row, first_row + n_rows) of the resident mirror `rows`.
// 8-bit sketch of float32 rows for the cosine pre-pass (kernels_exact.hip): rows [first_row, first_row + n_rows) of
// `src` (or the listed rows), written into `dst` in its resident layout; *max_ang = max over the rows of the angular
// distance row <-> sketch (bits of a double); rows without a direction or with a non-finite element are reported
hipError_t launch_sketch_build(const uint8_t *src, RowLayout src_lay, int dim, uint8_t *dst, RowLayout dst_lay,
                               uint64_t first_row, uint64_t n_rows, const uint32_t *row_list,
                               unsigned long long *max_ang, uint32_t *exc_rows, uint32_t *exc_count, uint32_t exc_cap,
                               double gscale, int max_only, hipStream_t stream);
// (gscale > 0: Euclidean form -- one scale for the whole collection, *max_ang = largest Euclidean distance row <->
//  sketch; max_only: report the largest finite |x_i| of the rows as float bits instead of building anything)
hipError_t launch_repack(int qbits, uint8_t *ref, uint32_t row_bytes, uint8_t *rows, RowLayout layout,
                         uint64_t first_row, uint64_t n_rows, int to_reference, hipStream_t stream);

// Rows [dst_first_row, +n_rows) of the mirror directly in the resident layout: synthetic
// (src == nullptr, see szg_index_synth; seed_first_row indexes the PRNG stream) or
// quantized + packed from float64 vectors on the device.
hipError_t launch_synth(int qbits, uint8_t *rows, RowLayout layout, uint64_t dst_first_row, int dim,
                        uint64_t n_rows, uint64_t seed, uint64_t seed_first_row, const double *src,
                        hipStream_t stream);

// Resident row norms (shared sweeps), out[first_row + i] for rows [first_row, first_row + n_rows) of the mirror:
// 16-bit rows (linear): the float32 sum of n^2, n = 2v - 65535, over the row's real elements, as the direct bfloat16
// sweep summed it; 8- and 4-bit rows (either layout): (float)(4 * sum (x'^2 + x')) + norm_bias, the int8 sweeps' norm.
hipError_t launch_row_norms(int bits, const uint8_t *rows, const RowLayout &lay, int dim, float norm_bias, uint64_t first_row,
                            uint64_t n_rows, float *out, hipStream_t stream);

// device float64 primitive probe (tests)
hipError_t launch_f64_probe(int op, const double *a, const double *b, double *out, uint64_t n,
                            hipStream_t stream);

// live-bit maintenance
hipError_t launch_fill_bits(uint64_t *bits, uint64_t n_rows, uint64_t n_words, hipStream_t stream);

}  // namespace szg
