// kernels_mq.hip -- shared (multi-query) sweeps for gfx950: B queries per pass of the corpus.
//
// The single-query scan (kernels_scan.hip) is HBM-bound: every query costs one full sweep.  When the caller hands
// over a batch (szg_search_topk with n_queries > 1, the reference's concurrent Searches under RLock,
// collection.go:570), the sweep is shared: the corpus streams through once and the B query x row dot products go to
// the matrix cores.
//
//   mq_score_bf16s_kernel  64-, 32- and 16-bit rows: rows and queries rounded to bfloat16 on the fly,
//                          v_mfma_f32_16x16x32_bf16; the sweep only RANKS -- its candidates are scored again in
//                          float32, re-ranked in float64 and certified against the bfloat16 bound.
//   mq_score_i8(s)_kernel  8- and 4-bit rows: exact integer arithmetic, v_mfma_i32_16x16x64_i8 on the row bytes
//                          against int8 digit planes of the query.
//   mq_thr_radix_kernel, mq_select_kernel, cand_refine_kernel, cand_rescore_kernel, cand_select_kernel
//                          thresholds of the fused selection, per-query selection over a score matrix or over the
//                          collected candidates, the float32 re-score of a bfloat16 sweep's band.
//
// A wave owns a tile of 16 rows; the D layout of the 16x16 product is column = lane & 15 (the tile's row),
// row = (lane >> 4) * 4 + reg (the query inside its block of 16).  (Round 4 removed the float32 MFMA form,
// v_mfma_f32_16x16x4_f32 at 62 % of its matrix roof: every width has an HBM-bound sweep now.)
#include "kernels.h"
#include "device_lists.h"

#include <algorithm>
#include <cstdlib>

// Built once per part (parallel build, like kernels_scan.hip): -DSZG_MQ_PART=1 / 2 carry the int8 sweeps for 8- / 4-bit
// rows, 3 / 116 / 164 the bfloat16 sweep for 32- / 16- / 64-bit rows, and the default (0) the selection kernels and
// the dispatchers.
#ifndef SZG_MQ_PART
#define SZG_MQ_PART 0
#endif
#define SZG_CAT2(a, b) a##b
#define SZG_CAT(a, b) SZG_CAT2(a, b)

namespace szg {

namespace {

#ifndef SZG_MQ_RING
#define SZG_MQ_RING 4  // 16-byte loads per lane in flight (4 vs 6: -1.5 % on the int8 sweeps, no change on f32)
#endif
#ifndef SZG_MQ8_WAVES
#define SZG_MQ8_WAVES 12  // waves per block (one block per CU) of the int8 sweeps (the 12-step shape kernel: 8)
#endif
#ifndef SZG_RESCORE_BLOCKS
#define SZG_RESCORE_BLOCKS 64  // blocks (of 4 waves) per query of the float32 re-score
#endif
#ifndef SZG_MQB_WAVES
#define SZG_MQB_WAVES 8  // waves per block (one block per CU) of the bfloat16 sweep: 8 x 2 steps x 2 KiB = 32 KiB in
#endif                    // flight per CU (16 waves or 3 steps: -3..-6 %, as on every streaming kernel here)
[[maybe_unused]] constexpr int kRingMq = SZG_MQ_RING;
[[maybe_unused]] constexpr int kMqbThreads = 64 * SZG_MQB_WAVES;
typedef int v4i32b __attribute__((ext_vector_type(4)));
[[maybe_unused]] constexpr int kMq8Threads = 64 * SZG_MQ8_WAVES;
[[maybe_unused]] constexpr int kMq8TableRows = 6;  // 48-float rows after a group's image: qscale, qconst, qnorm2 | thresholds, pre-test s, w

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// plain (cacheable) loads: the MFMA operand layout makes every lane group read
// 64-byte segments, and the other half of each 128-byte line is wanted one step
// later -- a non-temporal hint evicts it first (measured 1.22x HBM over-fetch)
__device__ __forceinline__ u32x4 load_nt(const uint8_t *p)
{
    return *reinterpret_cast<const u32x4 *>(p);
}
// tiled rows (4- and 8-bit): a wave instruction reads one whole KiB that is used once per sweep -- stream it past
// the caches.  The hint is a template argument, not a run-time flag: `if (nt) nontemporal_load(p) else load(p)` is
// folded by the optimiser into ONE plain load before inlining (the two arms read the same address and the merged
// instruction keeps only the metadata both carry), which is how the int8 sweeps came to run without the hint.
template <bool NT>
__device__ __forceinline__ u32x4 load_stream(const uint8_t *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    else return *reinterpret_cast<const u32x4 *>(p);
}

// Stage n16 16-byte words of a query image into LDS.  Written as load-all / store-all groups of six: with the plain
// `dst[i] = src[i]` loop every iteration waited for its own load, i.e. 18 L2 round trips back to back for a 147 KiB
// image (~20 us at the head of EVERY sweep launch, the prefix pass included, with HBM idle).
__device__ __forceinline__ void stage_image(uint4 *dst, const uint4 *src, int n16, int tid, int nthreads)
{
    constexpr int U = 6;
    int i = tid;
    for (; i + (U - 1) * nthreads < n16; i += U * nthreads) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = src[i + u * nthreads];
#pragma unroll
        for (int u = 0; u < U; u++) dst[i + u * nthreads] = v[u];
    }
    for (; i < n16; i += nthreads) dst[i] = src[i];
}

// Fused selection.  A (query, row) pair whose key is at or below the query's threshold goes
// into the wave's own little hit buffer in LDS (wave-synchronous append: ballot + prefix
// popcount, no atomics); when 64 are waiting -- normally only at the end of the kernel --
// each lane takes one, checks the row's mask bits and claims a slot in the query's global
// candidate buffer.  (One returning global atomic per hit, issued where the hit occurs,
// stalls the wave for a full memory round trip each time: measured +40 % on the sweep.)
constexpr int kHitCap = 64;
struct HitBuf {
    uint64_t *cand;  // [kHitCap]
    uint8_t *query;  // [kHitCap]
    int n;           // wave-uniform
};

__device__ __forceinline__ void hit_flush(const MqArgs &a, HitBuf &hb, int lane)
{
    if (lane < hb.n) {
        const uint64_t c = hb.cand[lane];
        const int q = hb.query[lane];
        const uint32_t r = (uint32_t)c;
        bool ok = true;
        if (a.live_bits) ok = (a.live_bits[r >> 6] >> (r & 63)) & 1;
        if (ok && a.allow_bits) ok = (a.allow_bits[(size_t)q * a.allow_stride + (r >> 6)] >> (r & 63)) & 1;
        if (ok) {
            const uint32_t idx = atomicAdd(a.cand_count + q * kCandCountStride, 1u);
            if (idx < a.cand_cap) a.cand_buf[(size_t)q * a.cand_cap + idx] = c;
        }
    }
    // gfx9: loads, stores and atomics share one vmcnt and retire out of order among
    // themselves; leaving these pending would turn every later ring wait into vmcnt(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    hb.n = 0;
}

__device__ __forceinline__ void hit_offer(const MqArgs &a, HitBuf &hb, int lane, bool hit, int q, uint64_t row,
                                          float key)
{
    const uint64_t m = __ballot(hit);
    if (!m) return;
    const int cnt = __popcll(m);
    if (hb.n + cnt > kHitCap) hit_flush(a, hb, lane);
    if (hit) {
        const int pos = hb.n + __popcll(m & ((1ull << lane) - 1ull));
        hb.cand[pos] = ((uint64_t)ordered_key(key) << 32) | (uint32_t)row;
        hb.query[pos] = (uint8_t)q;
    }
    hb.n += cnt;
}

// OR of a 32-bit value over the wave (uniform result): four DPP steps inside each row of 16
// lanes, then the four rows through scalar registers
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// The tile finish of the fused-selection sweeps.  A tile yields NB x 4 (query, row) keys per lane;
// almost none of them is at or below its query's threshold.  All keys are formed first (pure
// VALU work), the lanes' hit bits are OR-ed over the wave, and only the (query block, register)
// slots that hold a hit somewhere go through hit_offer: one wave-uniform branch per tile in the
// common case instead of one ballot and branch per slot.
template <int NB>
__device__ __forceinline__ void offer_tile_hits(const MqArgs &a, HitBuf &hb, int lane, int c, uint32_t hm,
                                                const float (&keys)[NB][4], uint64_t row, int qoff = 0)
{
    if (!__ballot(hm != 0)) return;
    // ONE copy of the offer (and of the flush inside it), walked over the slots that hold a hit somewhere in the wave:
    // `un` is wave-uniform, so the loop and the slot's key select are scalar-controlled.  (Round 3 unrolled the NB x 4
    // slots -- 24 inlined offers with a flush each, thousands of instructions in the middle of every sweep's loop: the
    // register allocator split the load ring's live ranges around them and copied freshly loaded registers at the
    // loop's end, which waits for every load in flight.)
    uint32_t un = wave_or_u32(hm);
    // (the keys as ONE register vector, indexed by the scalar slot number: v_movrels / s_set_gpr_idx, no memory.  A
    // chain of selects over the array was turned into a table in scratch memory, written by every tile.)
    typedef float keyvec __attribute__((ext_vector_type(NB <= 2 ? 8 : (NB <= 4 ? 16 : 32))));
    keyvec kv;
#pragma unroll
    for (int i = 0; i < NB * 4; i++) kv[i] = keys[i >> 2][i & 3];
    while (un) {
        const int s = __builtin_ctz(un);
        un &= un - 1u;
        const float key = kv[s];
        hit_offer(a, hb, lane, (hm >> s) & 1u, qoff + (s >> 2) * 16 + c * 4 + (s & 3), row, key);
    }
}

#if SZG_MQ_PART == 3 || SZG_MQ_PART == 116 || SZG_MQ_PART == 164
// ---- bfloat16 shared sweep: 32-, 16- and 64-bit rows ---------------------------------------------------------------------------
//
// The sweep only has to RANK: what it keeps is re-scored in float64 and certified against the
// bound of its own arithmetic (key_eps, bf16 branch), so its products need not carry 24 bits.
// Rows and queries are rounded to bfloat16 on the fly (v_cvt_pk_bf16_f32, round to nearest
// even: 7 fraction bits, relative error <= 2^-8 each, same exponent range as float32) and multiplied by
// v_mfma_f32_16x16x32_bf16 -- 16 x the rate of the float32 MFMA, which turns the 48-query
// sweep from matrix-bound (0.64 ms at 1M x 768) into a plain stream of the rows.  By
// Cauchy-Schwarz the dot product moves by at most (2^-7 + 2^-16) |x| |q|, i.e. 0.0078 in -cos:
// a band that holds on the order of a hundred rows of a million, all of which the float32 re-score sees.
//
// A wave owns a tile of 16 rows and multiplies 32 elements of them per step with one A operand per
// query block (image [32-element step][query block][lane = k-group*16 + query][8 bf16]).  Row
// norms (of the float32 values) are VALU side work.  Any dimension: rows are walked in
// 128-byte steps and the chunks of a short last step that lie past the row are read as zeros.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef SZG_MQB_RING
#define SZG_MQB_RING 2  // 32-byte (two-load) steps per lane in flight
#endif
constexpr int kRingB = SZG_MQB_RING;
#ifndef SZG_MQB_RING_PREFIX
#define SZG_MQB_RING_PREFIX 6
#endif
constexpr int kRingBPrefix = SZG_MQB_RING_PREFIX;

// Staged form: a load instruction reads 128 contiguous bytes of each of 8 rows (8 lanes x 16 bytes per
// row) instead of 64 bytes of each of 16 -- the streaming pattern the memory system likes better
// (scripts/readbw: 6.95 vs 6.2 TB/s) -- and the wave turns the two loads of a 32-element step into
// the MFMA operand layout through its own KiB of LDS: convert, ds_write_b64 in row-major order,
// ds_read_b128 as lane (row, k-group).  The image is in natural order: lane (query, k-group g)
// holds elements 8g..8g+7 of the step.
//
// QBITS = 16: the rows are 16-bit codes v, decoded on the fly to n = 2v - 65535 (exact in float32) and rounded to
// bfloat16 like float rows.  A 128-byte step then holds 64 elements = two MFMA K-steps: the wave stages and
// multiplies the lower and the upper four chunks one after the other through the same KiB.  A chunk read from the
// zero block (past a short last step) decodes to -65535 per element: zeros stand against it in the image, and
// those lanes stay out of the norm, and so do the padding codes inside the row's last piece (dim % 8 != 0).
template <int NB, int METRIC, bool COLLECT, int QBITS>
__global__ __launch_bounds__(kMqbThreads) void mq_score_bf16s_kernel(const MqArgs a)
{
    constexpr int KS = QBITS == 16 ? 2 : 1;  // 32-element MFMA K-steps per 128-byte step of a row (64-bit rows: half a one)
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int SS = (a.r16 + 7) / 8;             // 128-byte steps per row, the last one possibly short
    const int last_valid = a.r16 - 8 * (SS - 1);  // 16-byte chunks of the last step that belong to the row (1..8)
    const bool partial = last_valid < 8;
    const int n16 = (QBITS == 64 ? (SS + 1) / 2 : SS * KS) * NB * 64;  // a KiB per K-step and query block
    const int pad16 = QBITS == 16 ? a.r16 * 8 - a.dim : 0;  // 16-bit rows: padding codes in the row's last 16-byte piece
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.queries);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        stage_image(dst, src, n16, tid, blockDim.x);
        // table: [0, 96) thresholds, [96, 192) |q|^2
        if (COLLECT && tid < kMqMaxQueries)
            reinterpret_cast<float *>(smem + (size_t)n16 * 16)[tid] = tid < a.n_queries ? a.thr[tid] : -3.0e38f;
        if (METRIC != kCosine && tid >= 128 && tid < 128 + kMqMaxQueries)
            reinterpret_cast<float *>(smem + (size_t)n16 * 16)[tid - 32] = a.qnorm2[tid - 128];
    }
    const v4i32b *qimg = reinterpret_cast<const v4i32b *>(smem);
    const float *thr_lds = reinterpret_cast<const float *>(smem + (size_t)n16 * 16);
    HitBuf hb;
    uint8_t *stage;
    {
        uint8_t *base = smem + (size_t)n16 * 16 + 2 * kMqMaxQueries * sizeof(float);
        hb.cand = reinterpret_cast<uint64_t *>(base) + (size_t)wave * kHitCap;
        hb.query = base + (size_t)nwaves * kHitCap * 8 + (size_t)wave * kHitCap;
        hb.n = 0;
        stage = base + (size_t)nwaves * kHitCap * 9 + (size_t)wave * 1024;  // (kHitCap * 9 * nwaves is a multiple of 16)
    }

    const int trow = lane & 15, c = lane >> 4;  // MFMA role: row of the tile, k-group
    const int r8 = lane >> 3, ch = lane & 7;    // load role: rows r8 and 8 + r8, 16-byte chunk of the 128-byte step
    uint2 *w_a = reinterpret_cast<uint2 *>(stage + r8 * 64 + ch * 8);
    uint2 *w_b = reinterpret_cast<uint2 *>(stage + 512 + r8 * 64 + ch * 8);
    uint4 *w16_a = reinterpret_cast<uint4 *>(stage + r8 * 64 + (ch & 3) * 16);  // QBITS = 16: 8 bf16 per chunk, half a step at a time
    uint4 *w16_b = reinterpret_cast<uint4 *>(stage + 512 + r8 * 64 + (ch & 3) * 16);
    uint32_t *w64_a = reinterpret_cast<uint32_t *>(stage + r8 * 64 + ch * 4);  // QBITS = 64: 2 bf16 per chunk, 16 elements per step
    uint32_t *w64_b = reinterpret_cast<uint32_t *>(stage + 512 + r8 * 64 + ch * 4);
    const v4i32b *r_op = reinterpret_cast<const v4i32b *>(stage + trow * 64 + c * 16);

    const uint64_t n_tiles = ((uint64_t)a.n_rows + 15) / 16;
    const uint64_t tile_stride = (uint64_t)gridDim.x * nwaves;
    const uint64_t tile_first = (uint64_t)blockIdx.x * nwaves + wave;
    const uint64_t n_it = tile_first < n_tiles ? (n_tiles - tile_first + tile_stride - 1) / tile_stride : 0;
    const uint64_t NP = n_it * (uint64_t)SS;

    uint64_t itile = tile_first;
    int is = 0;
    uint64_t ctile = tile_first;
    int cs = 0;

    // the threshold pass (no COLLECT) sweeps a few tiles per wave on a few CUs: latency-bound, deeper ring
    constexpr int R = COLLECT ? kRingB : kRingBPrefix;
    u32x4 ring_a[R], ring_b[R];
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float nrm_a = 0.f, nrm_b = 0.f;
    uint32_t nz_a = 0, nz_b = 0;
    [[maybe_unused]] uint32_t nzl_a = 0, nzl_b = 0;  // 64-bit rows: the low words (whose bit 31 is data, not a sign)
    v4i32b qn[NB];

    const uint8_t *iptr_a, *iptr_b;
    auto set_rows = [&](uint64_t tile) {
        const uint64_t last = (uint64_t)a.n_rows - 1;  // past the end: a valid row, discarded
        iptr_a = a.rows + (size_t)min(tile * 16 + r8, last) * a.pitch + (size_t)ch * 16;
        iptr_b = a.rows + (size_t)min(tile * 16 + 8 + r8, last) * a.pitch + (size_t)ch * 16;
    };
    set_rows(tile_first);
    // a short step at the end of a row whose pitch is not a multiple of 128 bytes (any dimension that is not a
    // multiple of 32): the chunks past the row belong to the next row -- those lanes read the shard's zero block
    // instead (zeros for the products, the norm and the zero-row test alike; the padding inside the row's last
    // 16-byte piece is stored as zeros)
    const bool past = ch >= last_valid;

#define MQS_ISSUE(u)                                                                     \
    {                                                                                    \
        const bool z_ = partial && is == SS - 1 && past;                                 \
        ring_a[u] = load_stream<true>(z_ ? a.zero16 : iptr_a); /* whole 128-byte lines, used once: non-temporal */ \
        ring_b[u] = load_stream<true>(z_ ? a.zero16 : iptr_b);                           \
        if (++is == SS) {                                                                \
            is = 0;                                                                      \
            itile += tile_stride;                                                        \
            set_rows(itile);                                                             \
        } else {                                                                         \
            iptr_a += 128;                                                               \
            iptr_b += 128;                                                               \
        }                                                                                \
    }

#define MQS_CONSUME(u)                                                                   \
    {                                                                                    \
        const u32x4 va_ = ring_a[u], vb_ = ring_b[u];                                    \
        if constexpr (QBITS == 32) {                                                     \
            const float xa_[4] = {__uint_as_float(va_.x), __uint_as_float(va_.y), __uint_as_float(va_.z),   \
                                  __uint_as_float(va_.w)};                               \
            const float xb_[4] = {__uint_as_float(vb_.x), __uint_as_float(vb_.y), __uint_as_float(vb_.z),   \
                                  __uint_as_float(vb_.w)};                               \
            _Pragma("unroll") for (int i = 0; i < 4; i++) nrm_a = fmaf(xa_[i], xa_[i], nrm_a);   \
            _Pragma("unroll") for (int i = 0; i < 4; i++) nrm_b = fmaf(xb_[i], xb_[i], nrm_b);   \
            nz_a |= va_.x | va_.y;                                                       \
            nz_a |= va_.z | va_.w;                                                       \
            nz_b |= vb_.x | vb_.y;                                                       \
            nz_b |= vb_.z | vb_.w;                                                       \
            const bf16x2 t0_ = __builtin_convertvector(f32x2{xa_[0], xa_[1]}, bf16x2);   \
            const bf16x2 t1_ = __builtin_convertvector(f32x2{xa_[2], xa_[3]}, bf16x2);   \
            const bf16x2 t2_ = __builtin_convertvector(f32x2{xb_[0], xb_[1]}, bf16x2);   \
            const bf16x2 t3_ = __builtin_convertvector(f32x2{xb_[2], xb_[3]}, bf16x2);   \
            *w_a = make_uint2(__builtin_bit_cast(uint32_t, t0_), __builtin_bit_cast(uint32_t, t1_)); \
            *w_b = make_uint2(__builtin_bit_cast(uint32_t, t2_), __builtin_bit_cast(uint32_t, t3_)); \
            __builtin_amdgcn_wave_barrier();                                             \
            const v4i32b bop_ = *r_op;                                                   \
            __builtin_amdgcn_wave_barrier();                                             \
            const int qnext_ = lane + (cs + 1 == SS ? 0 : cs + 1) * (NB * 64);           \
            _Pragma("unroll") for (int b = 0; b < NB; b++)                               \
            {                                                                            \
                const v4i32b qc_ = qn[b];                                                \
                qn[b] = qimg[qnext_ + b * 64];                                           \
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qc_),          \
                                                                 __builtin_bit_cast(bf16x8, bop_), acc[b], 0, 0, 0); \
            }                                                                            \
        } else if constexpr (QBITS == 64) {                                              \
            /* two float64 elements per 16-byte chunk: narrowed to float32 (v_cvt_f32_f64; beyond the float32 range */ \
            /* -> inf or 0, and the row is forced into the candidates by its norm, as for float32 rows), the norm and */ \
            /* the bfloat16 operand are made of the float32 values.  A 128-byte step is HALF a K-step: the wave */ \
            /* stages two steps side by side in its KiB and multiplies after the second (or after a last odd one, */ \
            /* whose missing half is zeroed). */                                         \
            const float xa0_ = (float)__hiloint2double((int)va_.y, (int)va_.x);          \
            const float xa1_ = (float)__hiloint2double((int)va_.w, (int)va_.z);          \
            const float xb0_ = (float)__hiloint2double((int)vb_.y, (int)vb_.x);          \
            const float xb1_ = (float)__hiloint2double((int)vb_.w, (int)vb_.z);          \
            nrm_a = fmaf(xa0_, xa0_, nrm_a);                                             \
            nrm_a = fmaf(xa1_, xa1_, nrm_a);                                             \
            nrm_b = fmaf(xb0_, xb0_, nrm_b);                                             \
            nrm_b = fmaf(xb1_, xb1_, nrm_b);                                             \
            nz_a |= va_.y | va_.w;                                                       \
            nzl_a |= va_.x | va_.z;                                                      \
            nz_b |= vb_.y | vb_.w;                                                       \
            nzl_b |= vb_.x | vb_.z;                                                      \
            const int half_ = cs & 1;                                                    \
            const bool last_ = cs == SS - 1;                                             \
            w64_a[half_ * 8] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{xa0_, xa1_}, bf16x2)); \
            w64_b[half_ * 8] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{xb0_, xb1_}, bf16x2)); \
            if (last_ && half_ == 0) { /* (wave-uniform) an odd number of steps: no second half */ \
                w64_a[8] = 0u;                                                           \
                w64_b[8] = 0u;                                                           \
            }                                                                            \
            if (last_ || half_ == 1) {                                                   \
                __builtin_amdgcn_wave_barrier();                                         \
                const v4i32b bop_ = *r_op;                                               \
                __builtin_amdgcn_wave_barrier();                                         \
                const int qnext_ = lane + (last_ ? 0 : (cs >> 1) + 1) * (NB * 64);       \
                _Pragma("unroll") for (int b = 0; b < NB; b++)                           \
                {                                                                        \
                    const v4i32b qc_ = qn[b];                                            \
                    qn[b] = qimg[qnext_ + b * 64];                                       \
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qc_),      \
                                                                     __builtin_bit_cast(bf16x8, bop_), acc[b], 0, 0, 0); \
                }                                                                        \
            }                                                                            \
        } else {                                                                         \
            const uint32_t wa_[4] = {va_.x, va_.y, va_.z, va_.w}, wb_[4] = {vb_.x, vb_.y, vb_.z, vb_.w};    \
            const bool out_ = partial && cs == SS - 1 && past; /* read from the zero block: not part of the row */ \
            uint32_t pa_[4], pb_[4];                                                     \
            float xa_[8], xb_[8];                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                \
            {                                                                            \
                xa_[2 * i] = fmaf((float)(wa_[i] & 0xFFFFu), 2.0f, -65535.0f);           \
                xa_[2 * i + 1] = fmaf((float)(wa_[i] >> 16), 2.0f, -65535.0f);           \
                xb_[2 * i] = fmaf((float)(wb_[i] & 0xFFFFu), 2.0f, -65535.0f);           \
                xb_[2 * i + 1] = fmaf((float)(wb_[i] >> 16), 2.0f, -65535.0f);           \
                pa_[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{xa_[2 * i], xa_[2 * i + 1]}, bf16x2)); \
                pb_[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{xb_[2 * i], xb_[2 * i + 1]}, bf16x2)); \
            }                                                                            \
            float sa_ = 0.f, sb_ = 0.f;                                                  \
            if ((partial || pad16) && cs == SS - 1) { /* (wave-uniform) the row's last step: zero-block lanes and the */ \
                /* padding codes of the last piece decode to -65535 -- zeros stand against them in the image, and */ \
                /* they stay out of the norm (subtracting their squares afterwards would cost the small rows' norms */ \
                /* all their bits) */                                                    \
                const int nk_ = out_ ? 0 : (ch == last_valid - 1 ? 8 - pad16 : 8);       \
                _Pragma("unroll") for (int i = 0; i < 8; i++)                            \
                {                                                                        \
                    sa_ = i < nk_ ? fmaf(xa_[i], xa_[i], sa_) : sa_;                     \
                    sb_ = i < nk_ ? fmaf(xb_[i], xb_[i], sb_) : sb_;                     \
                }                                                                        \
            } else {                                                                     \
                _Pragma("unroll") for (int i = 0; i < 8; i++)                            \
                {                                                                        \
                    sa_ = fmaf(xa_[i], xa_[i], sa_);                                     \
                    sb_ = fmaf(xb_[i], xb_[i], sb_);                                     \
                }                                                                        \
            }                                                                            \
            nrm_a += sa_;                                                                \
            nrm_b += sb_;                                                                \
            nz_a = nz_b = 1u; /* a decoded code is odd: never a zero row */              \
            _Pragma("unroll") for (int h = 0; h < 2; h++)                                \
            {                                                                            \
                if ((ch >> 2) == h) {                                                    \
                    *w16_a = make_uint4(pa_[0], pa_[1], pa_[2], pa_[3]);                 \
                    *w16_b = make_uint4(pb_[0], pb_[1], pb_[2], pb_[3]);                 \
                }                                                                        \
                __builtin_amdgcn_wave_barrier();                                         \
                const v4i32b bop_ = *r_op;                                               \
                __builtin_amdgcn_wave_barrier();                                         \
                const int kn_ = cs * 2 + h + 1;                                          \
                const int qnext_ = lane + (kn_ == SS * 2 ? 0 : kn_) * (NB * 64);         \
                _Pragma("unroll") for (int b = 0; b < NB; b++)                           \
                {                                                                        \
                    const v4i32b qc_ = qn[b];                                            \
                    qn[b] = qimg[qnext_ + b * 64];                                       \
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qc_),      \
                                                                     __builtin_bit_cast(bf16x8, bop_), acc[b], 0, 0, 0); \
                }                                                                        \
            }                                                                            \
        }                                                                                \
        if (++cs == SS) {                                                                \
            finish_tile(ctile);                                                          \
            cs = 0;                                                                      \
            ctile += tile_stride;                                                        \
        }                                                                                \
    }

    auto finish_tile = [&](uint64_t tile) {
        // row norms: over the 8 chunk lanes of each row, then to the lanes of the MFMA result (column = row)
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            nrm_a += __shfl_xor(nrm_a, o);
            nrm_b += __shfl_xor(nrm_b, o);
            nz_a |= __shfl_xor(nz_a, o);
            nz_b |= __shfl_xor(nz_b, o);
            if constexpr (QBITS == 64) {
                nzl_a |= __shfl_xor(nzl_a, o);
                nzl_b |= __shfl_xor(nzl_b, o);
            }
        }
        const int src = (trow & 7) * 8;
        const float na = __shfl(nrm_a, src), nb2 = __shfl(nrm_b, src);
        const uint32_t za = __shfl(nz_a, src), zb = __shfl(nz_b, src);
        const float nrm = trow < 8 ? na : nb2;
        uint32_t nz = (trow < 8 ? za : zb) & 0x7FFFFFFFu;
        if constexpr (QBITS == 64) {
            const uint32_t zla = __shfl(nzl_a, src), zlb = __shfl(nzl_b, src);
            nz |= trow < 8 ? zla : zlb;
        }
        const uint64_t row = tile * 16 + trow;
        const float inv = __frsqrt_rn(nrm);
        // What depends on the ROW alone is settled once per lane, not once per (row, query) pair: a zero row (distance
        // 1.0, collection.go:828-830) or a norm beyond float32 (forced in: see RowAcc::finish) has one fixed key for
        // every query.  The two clamps (NaN and +inf -> the worst finite key) are ONE v_min_f32 -- minnum returns the
        // other operand for a NaN -- and the hit bits are combined without short-circuits: round 3's form compiled to
        // three exec-masked branches and ~12 vector instructions per pair, 40 % of the sweep's vector instructions on
        // 16-bit rows (PMC: 57 per K-step against the 30 of its step loop), on kernels whose SIMDs issue all the time.
        const bool use_fixed = METRIC == kCosine && (nrm == 0.f || !(nrm <= 3.0e38f));
        const float fixed = (nrm == 0.f && !nz) ? 1.0f : -2.0f;
        if (COLLECT || row < a.n_rows) {
            float keys[NB][4];
            uint32_t hm = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const float4 th = COLLECT ? *reinterpret_cast<const float4 *>(thr_lds + b * 16 + c * 4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                const float thv[4] = {th.x, th.y, th.z, th.w};
                const float4 qn4 = METRIC == kCosine ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                     : *reinterpret_cast<const float4 *>(thr_lds + kMqMaxQueries + b * 16 + c * 4);
                const float qnv[4] = {qn4.x, qn4.y, qn4.z, qn4.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float key;
                    if (METRIC == kCosine) {
                        key = -acc[b][r] * inv;
                        key = use_fixed ? fixed : key;
                    } else {
                        key = fmaf(-2.0f, acc[b][r], nrm + qnv[r]);
                    }
                    key = fminf(key, 3.0e38f);  // NaN, +inf -> 3e38
                    keys[b][r] = key;
                    if (COLLECT)  // (unused query slots carry a threshold of -3e38: never a hit)
                        hm |= (uint32_t)(key <= thv[r]) << (b * 4 + r);
                    else if (b * 16 + c * 4 + r < a.n_queries)
                        a.keys[(size_t)(b * 16 + c * 4 + r) * a.key_stride + row] = key;
                }
            }
            if (COLLECT) {
                hm = row < a.n_rows ? hm : 0u;
                offer_tile_hits<NB>(a, hb, lane, c, hm, keys, row);
            }
        }
        if (!COLLECT) __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the key stores: gfx9 counts loads and stores in ONE vmcnt, a pending store would turn every ring wait into vmcnt(0)
#pragma unroll
        for (int b = 0; b < NB; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        nrm_a = nrm_b = 0.f;
        nz_a = nz_b = 0;
        nzl_a = nzl_b = 0;
    };

    {
        uint64_t issued = R, consumed = 0;
#pragma unroll
        for (int u = 0; u < R; u++) {
            MQS_ISSUE(u)
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // the query image is complete (the rows do not depend on it)
#pragma unroll
        for (int b = 0; b < NB; b++) qn[b] = qimg[lane + b * 64];
        while (consumed + 2 * R <= NP) {
#pragma unroll
            for (int u = 0; u < R; u++) {
                MQS_CONSUME(u)
                MQS_ISSUE(u)
                __builtin_amdgcn_sched_barrier(0);
            }
            consumed += R;
            issued += R;
        }
        while (consumed < NP) {
#pragma unroll
            for (int u = 0; u < R; u++) {
                if (consumed < NP) {
                    MQS_CONSUME(u)
                    consumed++;
                    if (issued < NP) {
                        MQS_ISSUE(u)
                        issued++;
                    }
                }
            }
        }
    }
#undef MQS_ISSUE
#undef MQS_CONSUME
    if (COLLECT) hit_flush(a, hb, lane);
}

#endif  // SZG_MQ_PART == 3 || 116 || 164

#if SZG_MQ_PART == 116
// ---- 16-bit rows without the LDS stage: the codes arrive in the MFMA operand layout ---------------------------------
//
// A 16-byte chunk of a 16-bit row is eight codes -- exactly one lane's share (eight bfloat16) of the B operand of
// v_mfma_f32_16x16x32_bf16 (lane = k-group * 16 + row).  So the lanes load the chunks themselves, decode
// (n = 2v - 65535), round to bfloat16 in registers and multiply: no ds_write / ds_read / wait between the load and the
// matrix instruction, where the staged kernel above -- 56 VALU of decode, then write -> read -> wait TWICE per step --
// held 16-bit rows at 4.2-5.0 TB/s.
//
// WHICH chunk a lane loads is the round-4 lesson.  Loading the operand layout directly (lane = row & 15, chunk =
// lane >> 4: 64 bytes of each of 16 rows per instruction, the line's other half one instruction later) streams at
// 5.1-5.6 TB/s with NOTHING but the loads in the kernel (scripts/readbw/rowpat, mode 0), and the sweep sat at 5.3-5.5
// whatever was removed from its arithmetic (resident norms: 8 of 30 VALU per step gone, same time).  128 bytes of each
// of 8 rows per instruction (lane = row & 7, chunk = lane >> 3) streams at 7.0-7.2 (mode 2).  So a DOUBLE step loads X
// = rows 0-7 and Y = rows 8-15 of the tile, 128 bytes of each, and one DPP exchange per dword (row_ror:8 -- lane L
// takes from lane L ^ 8 -- under a bank mask) turns the pair into two operands in MFMA layout:
//     E[L] = L & 8 ? Y[L ^ 8] : X[L]      row L & 15, chunk 2 * (L >> 4)        (the even chunks of the 128 bytes)
//     O[L] = L & 8 ? Y[L] : X[L ^ 8]      row L & 15, chunk 2 * (L >> 4) + 1    (the odd chunks)
// The k order inside a matrix instruction is free as long as both operands agree, so the A operands are the SAME image
// read at permuted addresses: k-group g of E pairs with chunk 2g = K-step 2t + (g >> 1), k-group 2 (g & 1) of the
// image; O with the k-group after it.
#ifndef SZG_MQD_RING
#define SZG_MQD_RING 2  // PAIRS of 16-byte loads per lane in flight (2 x 2 KiB per wave)
#endif
#ifndef SZG_MQD_WAVES
#define SZG_MQD_WAVES 12  // waves per block (one block per CU): no staging KiB per wave, <= 168 registers: three per SIMD
#endif
constexpr int kMqdThreads = 64 * SZG_MQD_WAVES;
template <int NB, int METRIC, bool COLLECT>
__global__ __launch_bounds__(kMqdThreads) void mq_score_bf16d_kernel(const MqArgs a)
{
    constexpr int D = COLLECT ? SZG_MQD_RING : 4;  // (the threshold pass: a few tiles per wave, latency-bound)
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int DT = (a.r16 + 7) / 8;        // double steps (64 elements, 128 bytes of a row) per row, the last possibly short
    const bool partial = (a.r16 & 7) != 0;
    const int n16 = 2 * DT * NB * 64;      // the image holds an even number of K-steps (mq_bf16_image_bytes), zero-filled
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.queries);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        stage_image(dst, src, n16, tid, blockDim.x);
        // table: [0, 96) thresholds, [96, 192) |q|^2
        if (COLLECT && tid < kMqMaxQueries)
            reinterpret_cast<float *>(smem + (size_t)n16 * 16)[tid] = tid < a.n_queries ? a.thr[tid] : -3.0e38f;
        if (METRIC != kCosine && tid >= 128 && tid < 128 + kMqMaxQueries)
            reinterpret_cast<float *>(smem + (size_t)n16 * 16)[tid - 32] = a.qnorm2[tid - 128];
    }
    const v4i32b *qimg = reinterpret_cast<const v4i32b *>(smem);
    const float *thr_lds = reinterpret_cast<const float *>(smem + (size_t)n16 * 16);
    HitBuf hb;
    {
        uint8_t *base = smem + (size_t)n16 * 16 + 2 * kMqMaxQueries * sizeof(float);
        hb.cand = reinterpret_cast<uint64_t *>(base) + (size_t)wave * kHitCap;
        hb.query = base + (size_t)nwaves * kHitCap * 8 + (size_t)wave * kHitCap;
        hb.n = 0;
    }
    const int row8 = lane & 7, chunk = lane >> 3;  // as loaded: 128 bytes of each of 8 rows
    const int trow = lane & 15, c = lane >> 4;     // as multiplied (after the exchange), and the result's layout
    const uint64_t n_tiles = ((uint64_t)a.n_rows + 15) / 16;
    const uint64_t tile_stride = (uint64_t)gridDim.x * nwaves;
    const uint64_t tile_first = (uint64_t)blockIdx.x * nwaves + wave;
    const uint64_t n_it = tile_first < n_tiles ? (n_tiles - tile_first + tile_stride - 1) / tile_stride : 0;
    const uint64_t NP = n_it * (uint64_t)DT;
    const bool past = (DT - 1) * 8 + chunk >= a.r16;  // this lane's chunk of a short last double step lies beyond the row
    // the A operand of (double step t, half h, block b): qimg[t * 2 * NB * 64 + h * 16 + b * 64 + lane_e]
    const int lane_e = trow + 32 * (c & 1) + (c >> 1) * (NB * 64);

    uint64_t itile = tile_first, ctile = tile_first;
    int is = 0, cs = 0;
    u32x4 ring[2 * D];
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float nrm = 0.f;
    v4i32b qn[NB];
    auto row_ptr = [&](uint64_t tile, int half) -> const uint8_t * {
        const uint64_t r = min(tile * 16 + half * 8 + row8, (uint64_t)a.n_rows - 1);  // past the end: a valid row, discarded
        return a.rows + (size_t)r * a.pitch + (size_t)chunk * 16;
    };
    const uint8_t *ipx = row_ptr(tile_first, 0), *ipy = row_ptr(tile_first, 1);

#define MQD_ISSUE(u)                                                                     \
    {                                                                                    \
        const bool z_ = partial && is == DT - 1 && past;                                 \
        ring[2 * (u)] = load_stream<true>(z_ ? a.zero16 : ipx); /* (whole lines, used once: past the caches) */ \
        ring[2 * (u) + 1] = load_stream<true>(z_ ? a.zero16 : ipy);                      \
        if (++is == DT) {                                                                \
            is = 0;                                                                      \
            itile += tile_stride;                                                        \
            ipx = row_ptr(itile, 0);                                                     \
            ipy = row_ptr(itile, 1);                                                     \
        } else {                                                                         \
            ipx += 128;                                                                  \
            ipy += 128;                                                                  \
        }                                                                                \
    }

    // one operand (half h_ of the double step): decode, norm, NB matrix instructions, the next operands' reads
    // (Tried on the 64-byte form: the decode in packed float32 pairs -- v_pk_fma_f32, 20 instead of 28 vector
    // instructions per K-step -- 3-5 % SLOWER on the same box.  profiles/r04_bf16_16bit_experiments.txt.)
#define MQD_HALF(raw_, h_)                                                               \
    {                                                                                    \
        const uint32_t w_[4] = {raw_.x, raw_.y, raw_.z, raw_.w};                         \
        float x_[8];                                                                     \
        v4i32b bop_;                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                    \
        {                                                                                \
            x_[2 * i] = fmaf((float)(w_[i] & 0xFFFFu), 2.0f, -65535.0f);                 \
            x_[2 * i + 1] = fmaf((float)(w_[i] >> 16), 2.0f, -65535.0f);                 \
            bop_[i] = (int)__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x_[2 * i], x_[2 * i + 1]}, bf16x2)); \
        }                                                                                \
        /* resident norms (MqArgs::row_norm): this tile's 16 arrive while its steps run.  (Summing them here -- eight */ \
        /* more vector instructions per operand -- measured 3.5 % slower, and the threshold pass then spilled: */        \
        /* without the array the staged kernel runs, which sums its own.) */                                              \
        if ((h_) == 0 && cs == 0) nrm = a.row_norm[min(ctile * 16 + trow, (uint64_t)a.n_rows - 1)]; \
        const int qnext_ = lane_e + ((h_) == 0 ? cs * (2 * NB * 64) + 16 : (cs + 1 == DT ? 0 : cs + 1) * (2 * NB * 64)); \
        _Pragma("unroll") for (int b = 0; b < NB; b++)                                   \
        {                                                                                \
            const v4i32b qc_ = qn[b];                                                    \
            qn[b] = qimg[qnext_ + b * 64];                                               \
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qc_),              \
                                                             __builtin_bit_cast(bf16x8, bop_), acc[b], 0, 0, 0); \
        }                                                                                \
    }

#define MQD_CONSUME(u)                                                                   \
    {                                                                                    \
        const u32x4 vx_ = ring[2 * (u)], vy_ = ring[2 * (u) + 1];                         \
        u32x4 ve_, vo_;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                    \
        {   /* row_ror:8 = 0x128; bank mask 0x3: lanes 0-7 of every 16 are written, 0xC: lanes 8-15 */ \
            vo_[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)vy_[i], (int)vx_[i], 0x128, 0xF, 0x3, false); \
            ve_[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)vx_[i], (int)vy_[i], 0x128, 0xF, 0xC, false); \
        }                                                                                \
        MQD_HALF(ve_, 0)                                                                 \
        MQD_HALF(vo_, 1)                                                                 \
        if (++cs == DT) {                                                                \
            finish_tile(ctile);                                                          \
            cs = 0;                                                                      \
            ctile += tile_stride;                                                        \
        }                                                                                \
    }

    auto finish_tile = [&](uint64_t tile) {
        const uint64_t row = tile * 16 + trow;  // (the MFMA result's column = the row, as the operand's)
        const float inv = __frsqrt_rn(nrm);
        // (a decoded code is odd: the norm of a 16-bit row is neither 0 nor beyond float32 -- no fixed keys here; the
        // clamps are one v_min_f32 and the hit bits have no short-circuits: see mq_score_bf16s_kernel's finish)
        if (COLLECT || row < a.n_rows) {
            float keys[NB][4];
            uint32_t hm = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const float4 th = COLLECT ? *reinterpret_cast<const float4 *>(thr_lds + b * 16 + c * 4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                const float thv[4] = {th.x, th.y, th.z, th.w};
                const float4 qn4 = METRIC == kCosine ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                     : *reinterpret_cast<const float4 *>(thr_lds + kMqMaxQueries + b * 16 + c * 4);
                const float qnv[4] = {qn4.x, qn4.y, qn4.z, qn4.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float key = METRIC == kCosine ? -acc[b][r] * inv : fmaf(-2.0f, acc[b][r], nrm + qnv[r]);
                    key = fminf(key, 3.0e38f);  // NaN, +inf -> 3e38
                    keys[b][r] = key;
                    if (COLLECT)  // (unused query slots carry a threshold of -3e38: never a hit)
                        hm |= (uint32_t)(key <= thv[r]) << (b * 4 + r);
                    else if (b * 16 + c * 4 + r < a.n_queries)
                        a.keys[(size_t)(b * 16 + c * 4 + r) * a.key_stride + row] = key;
                }
            }
            if (COLLECT) {
                hm = row < a.n_rows ? hm : 0u;
                offer_tile_hits<NB>(a, hb, lane, c, hm, keys, row);
            }
        }
        if (!COLLECT) __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the key stores (one vmcnt for loads and stores)
#pragma unroll
        for (int b = 0; b < NB; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        nrm = 0.f;
    };

    {
        uint64_t issued = D, consumed = 0;
#pragma unroll
        for (int u = 0; u < D; u++) {
            MQD_ISSUE(u)
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // the query image is complete (the rows do not depend on it)
#pragma unroll
        for (int b = 0; b < NB; b++) qn[b] = qimg[lane_e + b * 64];
        while (consumed + 2 * D <= NP) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                MQD_CONSUME(u)
                MQD_ISSUE(u)
                __builtin_amdgcn_sched_barrier(0);
            }
            consumed += D;
            issued += D;
        }
        while (consumed < NP) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                if (consumed < NP) {
                    MQD_CONSUME(u)
                    consumed++;
                    if (issued < NP) {
                        MQD_ISSUE(u)
                        issued++;
                    }
                }
            }
        }
    }
#undef MQD_ISSUE
#undef MQD_HALF
#undef MQD_CONSUME
    if (COLLECT) hit_flush(a, hb, lane);
}
#endif  // SZG_MQ_PART == 116 (direct 16-bit)

#if SZG_MQ_PART == 108
// ---- 8-bit rows through the bfloat16 matrix instruction: 96 queries per pass -----------------------------------------
//
// An 8-bit code is EXACT in bfloat16: v - 128 = -128..127 has eight significant bits.  So the rows need no digit planes and no
// integer arithmetic to be multiplied exactly -- only the QUERY is rounded (to bfloat16, as for float rows), which the
// bfloat16 path's second stage (float32 re-score of the band, §4.2a) and bounds already cover.  What that buys: the
// image of 96 queries is 6 KiB per 32 elements instead of the int8 sweep's 2 planes x 3 KiB per 48 queries, i.e. ONE
// pass of the rows per 96 queries where the int8 sweep makes two, for the same number of matrix instructions.
// The row operand: lane (row = lane & 15, c = lane >> 4) loads its 16 bytes of the 64-byte step of a TILED row (one
// contiguous KiB per wave instruction) = 16 codes = its share of TWO B operands (codes 0-7 and 8-15); the A operands
// are the natural image at the permuted addresses of mq_score_bf16d_kernel.  With v' = v - 128, n = 2v' + 1:
// sum g n = 2 sum g v' + sum g; sum g (over the ROUNDED image) is a per-query constant staged beside the thresholds
// (MqArgs::qsum).
// Norms: the resident array (launch_row_norms, the int8 formula = sum n^2 of the real elements).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef SZG_MQD8_WAVES
#define SZG_MQD8_WAVES 12
#endif
#ifndef SZG_MQD8_RING
#define SZG_MQD8_RING 4
#endif
constexpr int kMqd8Threads = 64 * SZG_MQD8_WAVES;
template <int NB, int METRIC, bool COLLECT>
__global__ __launch_bounds__(kMqd8Threads) void mq_score_bf16d8_kernel(const MqArgs a)
{
    constexpr int D = COLLECT ? SZG_MQD8_RING : 6;
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int DT = (int)a.steps;        // 64-byte steps per (tiled) row = double K-steps
    const int n16 = 2 * DT * NB * 64;   // a KiB per 32-element K-step and query block
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.queries);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        stage_image(dst, src, n16, tid, blockDim.x);
        // table: [0, 96) thresholds, [96, 192) |g|^2, [192, 288) sum g
        float *tab = reinterpret_cast<float *>(smem + (size_t)n16 * 16);
        if (COLLECT && tid < kMqMaxQueries) tab[tid] = tid < a.n_queries ? a.thr[tid] : -3.0e38f;
        if (tid >= 128 && tid < 128 + kMqMaxQueries) tab[tid - 32] = a.qnorm2[tid - 128];
        if (tid >= 256 && tid < 256 + kMqMaxQueries) tab[tid - 64] = a.qsum[tid - 256];
    }
    const v4i32b *qimg = reinterpret_cast<const v4i32b *>(smem);
    const float *thr_lds = reinterpret_cast<const float *>(smem + (size_t)n16 * 16);
    HitBuf hb;
    {
        uint8_t *base = smem + (size_t)n16 * 16 + 3 * kMqMaxQueries * sizeof(float);
        hb.cand = reinterpret_cast<uint64_t *>(base) + (size_t)wave * kHitCap;
        hb.query = base + (size_t)nwaves * kHitCap * 8 + (size_t)wave * kHitCap;
        hb.n = 0;
    }
    const int trow = lane & 15, c = lane >> 4;
    const uint64_t n_tiles = ((uint64_t)a.n_rows + 15) / 16;
    const uint64_t tile_stride = (uint64_t)gridDim.x * nwaves;
    const uint64_t tile_first = (uint64_t)blockIdx.x * nwaves + wave;
    const uint64_t n_it = tile_first < n_tiles ? (n_tiles - tile_first + tile_stride - 1) / tile_stride : 0;
    const uint64_t NP = n_it * (uint64_t)DT;
    const int lane_e = trow + 32 * (c & 1) + (c >> 1) * (NB * 64);  // (see mq_score_bf16d_kernel)
    const size_t tile_bytes = (size_t)DT * 1024;
    const size_t lane_off = (size_t)trow * 64 + (size_t)c * 16;

    uint64_t itile = tile_first, ctile = tile_first;
    int is = 0, cs = 0;
    u32x4 ring[D];
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float nrm = 0.f;
    v4i32b qn[NB];
    const uint8_t *iptr = a.rows + (size_t)min(tile_first, n_tiles - 1) * tile_bytes + lane_off;

#define MQ8_ISSUE(u)                                                                     \
    {                                                                                    \
        ring[u] = load_stream<true>(iptr);                                               \
        if (++is == DT) {                                                                \
            is = 0;                                                                      \
            itile += tile_stride;                                                        \
            iptr = a.rows + (size_t)min(itile, n_tiles - 1) * tile_bytes + lane_off; /* (past the end: the last tile, discarded) */ \
        } else {                                                                         \
            iptr += 1024;                                                                \
        }                                                                                \
    }

    // one operand: two dwords = eight codes -> v - 128 as float -> bfloat16 pairs (exact), NB matrix instructions
// The codes are multiplied as v' = v - 128 (one xor per dword, then a sign-extending byte convert), NOT as v: with
// n = 2v' + 1 the accumulator holds sum g v' -- small when the row is (a zero vector is all codes 128) -- whereas
// sum g v - 127.5 sum g would cancel two numbers 128 x larger than their difference inside the matrix core's float32
// sums, an error key_eps' bfloat16 branch has no term for.  (The unsigned form measured 2 % faster.)
#define MQ8_PK(a_, b_) (int)__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{(float)(a_), (float)(b_)}, bf16x2))
#define MQ8_DECODE(w0_, w1_)                                                             \
        {                                                                                \
            const uint32_t s0_ = (w0_) ^ 0x80808080u, s1_ = (w1_) ^ 0x80808080u;         \
            bop_[0] = MQ8_PK((int8_t)s0_, (int8_t)(s0_ >> 8));                           \
            bop_[1] = MQ8_PK((int8_t)(s0_ >> 16), (int8_t)(s0_ >> 24));                  \
            bop_[2] = MQ8_PK((int8_t)s1_, (int8_t)(s1_ >> 8));                           \
            bop_[3] = MQ8_PK((int8_t)(s1_ >> 16), (int8_t)(s1_ >> 24));                  \
        }
#define MQ8_HALF(w0_, w1_, h_)                                                           \
    {                                                                                    \
        v4i32b bop_;                                                                     \
        MQ8_DECODE(w0_, w1_)                                                             \
        if ((h_) == 0 && cs == 0) nrm = a.row_norm[min(ctile * 16 + trow, (uint64_t)a.n_rows - 1)]; \
        const int qnext_ = lane_e + ((h_) == 0 ? cs * (2 * NB * 64) + 16 : (cs + 1 == DT ? 0 : cs + 1) * (2 * NB * 64)); \
        _Pragma("unroll") for (int b = 0; b < NB; b++)                                   \
        {                                                                                \
            const v4i32b qc_ = qn[b];                                                    \
            qn[b] = qimg[qnext_ + b * 64];                                               \
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qc_),              \
                                                             __builtin_bit_cast(bf16x8, bop_), acc[b], 0, 0, 0); \
        }                                                                                \
    }

#define MQ8_CONSUME(u)                                                                   \
    {                                                                                    \
        const u32x4 v_ = ring[u];                                                        \
        MQ8_HALF(v_.x, v_.y, 0)                                                          \
        MQ8_HALF(v_.z, v_.w, 1)                                                          \
        if (++cs == DT) {                                                                \
            finish_tile(ctile);                                                          \
            cs = 0;                                                                      \
            ctile += tile_stride;                                                        \
        }                                                                                \
    }

    auto finish_tile = [&](uint64_t tile) {
        const uint64_t row = tile * 16 + trow;
        const float inv = __frsqrt_rn(nrm);  // (every n is odd: the norm of an 8-bit row is at least its dimension)
        if (COLLECT || row < a.n_rows) {
            float keys[NB][4];
            uint32_t hm = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const float4 th = COLLECT ? *reinterpret_cast<const float4 *>(thr_lds + b * 16 + c * 4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                const float thv[4] = {th.x, th.y, th.z, th.w};
                const float4 qn4 = METRIC == kCosine ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                     : *reinterpret_cast<const float4 *>(thr_lds + kMqMaxQueries + b * 16 + c * 4);
                const float qnv[4] = {qn4.x, qn4.y, qn4.z, qn4.w};
                const float4 qs4 = *reinterpret_cast<const float4 *>(thr_lds + 2 * kMqMaxQueries + b * 16 + c * 4);
                const float qsv[4] = {qs4.x, qs4.y, qs4.z, qs4.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float dotn = fmaf(2.0f, acc[b][r], qsv[r]);  // sum g n = 2 sum g v' + sum g
                    float key = METRIC == kCosine ? -dotn * inv : fmaf(-2.0f, dotn, nrm + qnv[r]);
                    key = fminf(key, 3.0e38f);  // NaN, +inf -> 3e38
                    keys[b][r] = key;
                    if (COLLECT)  // (unused query slots carry a threshold of -3e38: never a hit)
                        hm |= (uint32_t)(key <= thv[r]) << (b * 4 + r);
                    else if (b * 16 + c * 4 + r < a.n_queries)
                        a.keys[(size_t)(b * 16 + c * 4 + r) * a.key_stride + row] = key;
                }
            }
            if (COLLECT) {
                hm = row < a.n_rows ? hm : 0u;
                offer_tile_hits<NB>(a, hb, lane, c, hm, keys, row);
            }
        }
        if (!COLLECT) __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the key stores (one vmcnt for loads and stores)
#pragma unroll
        for (int b = 0; b < NB; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        nrm = 0.f;
    };

    {
        uint64_t issued = D, consumed = 0;
#pragma unroll
        for (int u = 0; u < D; u++) {
            MQ8_ISSUE(u)
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // the query image is complete (the rows do not depend on it)
#pragma unroll
        for (int b = 0; b < NB; b++) qn[b] = qimg[lane_e + b * 64];
        while (consumed + 2 * D <= NP) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                MQ8_CONSUME(u)
                MQ8_ISSUE(u)
                __builtin_amdgcn_sched_barrier(0);
            }
            consumed += D;
            issued += D;
        }
        while (consumed < NP) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                if (consumed < NP) {
                    MQ8_CONSUME(u)
                    consumed++;
                    if (issued < NP) {
                        MQ8_ISSUE(u)
                        issued++;
                    }
                }
            }
        }
    }
#undef MQ8_ISSUE
#undef MQ8_HALF
#undef MQ8_CONSUME
    if (COLLECT) hit_flush(a, hb, lane);
}
#endif  // SZG_MQ_PART == 108

#if SZG_MQ_PART == 1 || SZG_MQ_PART == 2
// ---- exact integer shared sweep, 8-bit rows (part 1) and 4-bit rows (part 2) ---------------------------------------
//
// With v' = v - 128 (one xor per dword) the decoded element is n = 2v' + 1, and the
// prepared query is the integer vector Q = 16384 h + 128 m + l of balanced int8 digits
// (prep_query, the same planes the single-query integer path uses).  One
// v_mfma_i32_16x16x64_i8 per digit plane multiplies 64 elements of 16 rows with 16
// queries, exactly: B operand = the row bytes as they come from HBM (lane = chunk*16 +
// row holds 16 consecutive elements), A operand = the plane's bytes from LDS (lane =
// chunk*16 + query, same elements).  Both operands use the same lane -> K mapping, so the
// products pair element with element whatever the hardware's K order is.  The row norm
// comes from two v_dot4_i32_i8 per dword.  The finish is RowAcc<8>::finish's, so the key
// and its error bound (key_eps, integer branch) are the single-query path's.
typedef int v4i32 __attribute__((ext_vector_type(4)));

template <int NB, int METRIC, bool COLLECT, bool FAST = false, int RB = 8>
__global__ __launch_bounds__(kMq8Threads) void mq_score_i8_kernel(const MqArgs a)
{
    // RB = 8: one B operand per 16-byte piece (the bytes, xor 0x80).  RB = 4: two -- the
    // high nibbles (even elements) and the low nibbles (odd elements) as unsigned bytes
    // 0..15, against the digit planes of the even / odd elements; n = 2x - 15 turns
    // sum Q x into sum Q n on the host side of the constants table.
    constexpr int T = RB == 4 ? 2 : 1;
    constexpr int NPL = kMqPlanes;  // digit planes of the query (radix 128)
    // prefetch the A operands one step ahead -- where the registers are there: with three query blocks the prefetched
    // set (48 VGPRs for 4-bit rows) pushed these kernels over the 168 registers of 12 waves per CU and they spilled
    // 8-21 of them (round 3's builds; -Rpass-analysis=kernel-resource-usage, scripts/kernel_resources.sh)
    constexpr bool PF = NB < 3;
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r16 = a.r16;
    const int steps = (r16 + 3) / 4;  // 64-byte steps per row
    const RowLayout mlay{a.pitch, a.tiled, a.steps};
    const uint32_t istep = a.tiled ? 1024u : 64u;  // bytes from one 64-byte step of a row to the next
    const int n16 = steps * NPL * T * NB * 64;  // image, 16-byte words
    // One launch walks the passes of up to two query groups (48 queries each) back to back, as the
    // single-query scan walks its sweeps: a 0.13 ms pass at 1M rows otherwise pays its start-up and its
    // tail (9 %) once per launch.  Both groups' images are staged in LDS up front (2 x 73 KiB at 768
    // dims), so a wave that finishes its share of the first pass goes straight on to the second.
    const int n_groups = a.n_groups > 0 ? a.n_groups : 1;
    const size_t grp_lds = (size_t)n16 * 16 + 4 * 48 * sizeof(float);  // image | qscale, qconst, qnorm2 | thresholds
    for (int g = 0; g < n_groups; g++) {
        const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.queries) +
                                                           (size_t)g * a.group_stride);
        uint4 *dst = reinterpret_cast<uint4 *>(smem + (size_t)g * grp_lds);
        const int n = n16 + (3 * 48 * 4) / 16;  // + constants table
        stage_image(dst, src, n, tid, blockDim.x);
        if (COLLECT && tid < 48)
            reinterpret_cast<float *>(smem + (size_t)g * grp_lds + (size_t)n * 16)[tid] =
                g * 48 + tid < a.n_queries ? a.thr[g * 48 + tid] : -3.0e38f;
    }
    HitBuf hb;
    {
        uint8_t *base = smem + (size_t)n_groups * grp_lds;
        hb.cand = reinterpret_cast<uint64_t *>(base) + (size_t)wave * kHitCap;
        hb.query = base + (size_t)nwaves * kHitCap * 8 + (size_t)wave * kHitCap;
        hb.n = 0;
    }
    for (int grp = 0; grp < n_groups; grp++) {
    const int qoff = grp * 48;  // first query of the group
    const uint8_t *gbase = smem + (size_t)grp * grp_lds;
    // (the barrier that publishes the image comes after the ring's first loads have been issued:
    // the rows do not depend on it, and a 140 us sweep notices a 5 us start-up)
    const v4i32 *qimg = reinterpret_cast<const v4i32 *>(gbase);
    const float *qtab = reinterpret_cast<const float *>(gbase + (size_t)n16 * 16);
    const float *thr_lds = qtab + 3 * 48;

    const int trow = lane & 15;
    const int c = lane >> 4;
    const uint64_t n_tiles = ((uint64_t)a.n_rows + 15) / 16;
    const uint64_t tile_stride = (uint64_t)gridDim.x * nwaves;
    const uint64_t tile_first = (uint64_t)blockIdx.x * nwaves + wave;
    const uint64_t n_it = tile_first < n_tiles ? (n_tiles - tile_first + tile_stride - 1) / tile_stride : 0;
    const uint64_t NP = n_it * (uint64_t)steps;

    uint64_t itile = tile_first;
    int is = 0;
    uint64_t ctile = tile_first;
    int cs = 0;

    u32x4 ring[kRingMq];
    v4i32 acc[NPL][NB];
#pragma unroll
    for (int p = 0; p < NPL; p++)
#pragma unroll
        for (int b = 0; b < NB; b++) acc[p][b] = v4i32{0, 0, 0, 0};
    int SQ = 0, SV = 0;
    const int qstep8 = NPL * T * NB * 64;  // 16-byte words of the image per 64-byte step
    // A operands of the step about to be multiplied (fetched one step ahead when PF)
    v4i32 qn[NPL][T][NB];

    // FAST (r16 % 4 == 0): no range predicates, addresses advance by constants
    auto row_ptr = [&](uint64_t tile) -> const uint8_t * {
        const uint64_t r = min(tile * 16 + trow, (uint64_t)a.n_rows - 1);
        return a.rows + piece_offset(mlay, r, (uint32_t)c);
    };
    const uint8_t *iptr = row_ptr(tile_first);

#define MQ8F_ISSUE(u)                                                                    \
    {                                                                                    \
        ring[u] = load_stream<true>(iptr); /* FAST: tiled */                                      \
        if (++is == steps) {                                                             \
            is = 0;                                                                      \
            itile += tile_stride;                                                        \
            iptr = row_ptr(itile);                                                       \
        } else {                                                                         \
            iptr += istep;                                                               \
        }                                                                                \
    }

#define MQ8_ISSUE(u)                                                                     \
    {                                                                                    \
        const uint64_t row_ = itile * 16 + trow;                                         \
        const int j_ = is * 4 + c;                                                       \
        const bool ok_ = row_ < a.n_rows && j_ < r16;                                    \
        ring[u] = load_nt(ok_ ? a.rows + piece_offset(mlay, row_, (uint32_t)j_) : a.zero16); \
        if (++is == steps) {                                                             \
            is = 0;                                                                      \
            itile += tile_stride;                                                        \
        }                                                                                \
    }

    // PRED: the piece may be the dummy one (not part of the row): its operands become 0
#define MQ8_CONSUME_X(u, PRED)                                                           \
    {                                                                                    \
        const u32x4 v_ = ring[u];                                                        \
        const bool in_ = !(PRED) || cs * 4 + c < r16;                                    \
        const uint32_t raw_[4] = {v_.x, v_.y, v_.z, v_.w};                               \
        v4i32 bop_[T];                                                                   \
        int wn_[4];                                                                      \
        _Pragma("unroll") for (int d = 0; d < 4; d++)                                    \
        {                                                                                \
            if (RB == 8) {                                                               \
                wn_[d] = in_ ? (int)(raw_[d] ^ 0x80808080u) : 0;                         \
                bop_[0][d] = wn_[d];                                                     \
            } else {                                                                     \
                wn_[d] = in_ ? (int)(raw_[d] ^ 0x88888888u) : 0;                         \
                bop_[0][d] = in_ ? (int)((raw_[d] >> 4) & 0x0F0F0F0Fu) : 0;              \
                bop_[T - 1][d] = in_ ? (int)(raw_[d] & 0x0F0F0F0Fu) : 0;                 \
            }                                                                            \
        }                                                                                \
        v4i32 qc_[NPL][T][NB];                                                            \
        const int qcur_ = lane + cs * qstep8;                                            \
        const int qnext_ = lane + (cs + 1 == steps ? 0 : cs + 1) * qstep8;               \
        _Pragma("unroll") for (int p = 0; p < NPL; p++)                                   \
            _Pragma("unroll") for (int t = 0; t < T; t++)                                \
                _Pragma("unroll") for (int b = 0; b < NB; b++)                           \
                {                                                                        \
                    if (PF) {                                                            \
                        qc_[p][t][b] = qn[p][t][b];                                      \
                        qn[p][t][b] = qimg[qnext_ + ((p * T + t) * NB + b) * 64];        \
                    } else {                                                             \
                        qc_[p][t][b] = qimg[qcur_ + ((p * T + t) * NB + b) * 64];        \
                    }                                                                    \
                }                                                                        \
        _Pragma("unroll") for (int t = 0; t < T; t++)                                    \
            _Pragma("unroll") for (int p = 0; p < NPL; p++)                               \
                _Pragma("unroll") for (int b = 0; b < NB; b++)                           \
                {                                                                        \
                    acc[p][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qc_[p][t][b], bop_[t], acc[p][b], 0, 0, 0); \
                }                                                                        \
        _Pragma("unroll") for (int d = 0; d < 4; d++)                                    \
        {                                                                                \
            if (RB == 8) {                                                               \
                SQ = __builtin_amdgcn_sdot4(wn_[d], wn_[d], SQ, false);                  \
                SV = __builtin_amdgcn_sdot4(wn_[d], 0x01010101, SV, false);              \
            } else {                                                                     \
                SQ = __builtin_amdgcn_sdot8(wn_[d], wn_[d], SQ, false);                  \
                SV = __builtin_amdgcn_sdot8(wn_[d], 0x11111111, SV, false);              \
            }                                                                            \
        }                                                                                \
        if (++cs == steps) {                                                             \
            finish_tile8(ctile);                                                         \
            cs = 0;                                                                      \
            ctile += tile_stride;                                                        \
        }                                                                                \
    }
#define MQ8_CONSUME(u) MQ8_CONSUME_X(u, true)
#define MQ8F_CONSUME(u) MQ8_CONSUME_X(u, false)

    auto finish_tile8 = [&](uint64_t tile) {
        int nrm = 4 * (SQ + SV);
        nrm += __shfl_xor(nrm, 16);
        nrm += __shfl_xor(nrm, 32);
        const float norm = (float)nrm + a.norm_bias;
        const float inv = __frsqrt_rn(norm);
        const uint64_t row = tile * 16 + trow;
        if (COLLECT || row < a.n_rows) {
            float keys[NB][4];
            uint32_t hm = 0;
            const bool row_ok = row < a.n_rows;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                // this lane's four queries of block b are consecutive: 16-byte reads of the tables
                const int q0 = b * 16 + c * 4;
                const float4 qs4 = *reinterpret_cast<const float4 *>(qtab + q0);
                const float4 qc4 = *reinterpret_cast<const float4 *>(qtab + 48 + q0);
                const float4 qn4 = METRIC == kCosine ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                     : *reinterpret_cast<const float4 *>(qtab + 96 + q0);
                const float4 th4 = COLLECT ? *reinterpret_cast<const float4 *>(thr_lds + q0)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                const float qsv[4] = {qs4.x, qs4.y, qs4.z, qs4.w}, qcv[4] = {qc4.x, qc4.y, qc4.z, qc4.w};
                const float qnv[4] = {qn4.x, qn4.y, qn4.z, qn4.w}, thv[4] = {th4.x, th4.y, th4.z, th4.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int q = q0 + r;
                    float dot = (float)acc[0][b][r];  // plane 0 = the top digit
#pragma unroll
                    for (int p = 1; p < NPL; p++) dot = fmaf(128.0f, dot, (float)acc[p][b][r]);
                    const float d2 = fmaf(2.0f, dot, qcv[r]);  // sum Q n
                    float key;
                    if (METRIC == kCosine)
                        key = -(d2 * qsv[r]) * inv;
                    else
                        key = fmaf(-2.0f * qsv[r], d2, qnv[r] + norm);
                    // (finite by construction: integer sums, norm >= dim > 0 -- no NaN / inf clamps)
                    keys[b][r] = key;
                    if (COLLECT)
                        hm |= (uint32_t)(row_ok & (key <= thv[r])) << (b * 4 + r);  // unused queries: thr = -3e38
                    else if (qoff + q < a.n_queries)
                        a.keys[(size_t)(qoff + q) * a.key_stride + row] = key;
                }
            }
            if (COLLECT) offer_tile_hits<NB>(a, hb, lane, c, hm, keys, row, qoff);
        }
        if (!COLLECT) __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the key stores: gfx9 counts loads and stores in ONE vmcnt, a pending store would turn every ring wait into vmcnt(0)
#pragma unroll
        for (int p = 0; p < NPL; p++)
#pragma unroll
            for (int b = 0; b < NB; b++) acc[p][b] = v4i32{0, 0, 0, 0};
        SQ = 0;
        SV = 0;
    };

#define MQ8_RUN_RING(ISSUE, CONSUME)                                                     \
    {                                                                                    \
        uint64_t issued = kRingMq, consumed = 0;                                         \
        _Pragma("unroll") for (int u = 0; u < kRingMq; u++)                              \
        {                                                                                \
            ISSUE(u)                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                           \
        }                                                                                \
        if (grp == 0) __syncthreads(); /* the query images are complete */               \
        if (PF) {                                                                        \
            _Pragma("unroll") for (int p = 0; p < NPL; p++)                              \
                _Pragma("unroll") for (int t = 0; t < T; t++)                            \
                    _Pragma("unroll") for (int b = 0; b < NB; b++)                       \
                        qn[p][t][b] = qimg[((p * T + t) * NB + b) * 64 + lane];          \
        }                                                                                \
        while (consumed + 2 * kRingMq <= NP) {                                           \
            _Pragma("unroll") for (int u = 0; u < kRingMq; u++)                          \
            {                                                                            \
                CONSUME(u)                                                               \
                ISSUE(u)                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                       \
            }                                                                            \
            consumed += kRingMq;                                                         \
            issued += kRingMq;                                                           \
        }                                                                                \
        while (consumed < NP) {                                                          \
            _Pragma("unroll") for (int u = 0; u < kRingMq; u++)                          \
            {                                                                            \
                if (consumed < NP) {                                                     \
                    CONSUME(u)                                                           \
                    consumed++;                                                          \
                    if (issued < NP) {                                                   \
                        ISSUE(u)                                                         \
                        issued++;                                                        \
                    }                                                                    \
                }                                                                        \
            }                                                                            \
        }                                                                                \
    }
    if (FAST)
        MQ8_RUN_RING(MQ8F_ISSUE, MQ8F_CONSUME)
    else
        MQ8_RUN_RING(MQ8_ISSUE, MQ8_CONSUME)
    }  // groups
    if (COLLECT) hit_flush(a, hb, lane);  // once for both groups (see mq_score_i8s_kernel)
#undef MQ8F_ISSUE
#undef MQ8F_CONSUME
#undef MQ8_CONSUME_X
#undef MQ8_RUN_RING
#undef MQ8_ISSUE
#undef MQ8_CONSUME
}



// ---- the same sweep with the row shape fixed at compile time ----------------------------------------------------
//
// STEPS = 64-byte steps per row (12 for 768 8-bit dims, 6 for 768 4-bit or 384 8-bit, 3 for 384 4-bit).  The loop
// walks one TILE per iteration, its STEPS steps unrolled with slot = step % D (D divides STEPS), so every load
// address is `tile pointer + constant`, every A operand an LDS read at a constant offset, the ring wait a fixed
// vmcnt(D-1), and there is ONE copy of the tile finish (the rotating-slot loop above carries four, each with the
// inlined hit path: 13 000 lines of ISA).  Whole 64-byte steps of tiled rows, fused selection only; other shapes
// keep mq_score_i8_kernel.  Measured against it (1M rows, ms per 48-query pass): 768 dims 8-bit 0.122 / 0.134,
// 768 dims 4-bit 0.088 / 0.092, 384 dims 4-bit 0.053 / 0.062 (profiles/r03_i8_sweep_experiments.txt, which also
// has the probe -- scripts/readbw -- that found the int8 sweeps running without their non-temporal hint).
// Per-query constants of the shape kernels' hit PRE-TEST (see the kernel's tile finish).  With g = sum Q n (a float)
// the key is  cosine: -fl(fl(g qs) inv)   Euclidean: fl(fma(-2 qs, g, fl(qn + norm))),  and a hit is key <= thr.
//   cosine:     key <= thr  ==>  g inv >= (-thr - 4e-7 |thr|) / qs =: T             pre-test  fma(g, inv, w) >= 0, w = -T
//   Euclidean:  key <= thr  ==>  2 qs g - norm (1 - 6e-8) >= qn - thr - 2e-6 (qn + |thr|) =: V
//                                                          pre-test  fma(g, s, w) >= norm (1 - 2e-6), s = 2 qs, w = -V
// (two roundings of 2^-24 each in the cosine chain, one plus the rounded qn + norm in the Euclidean one; the margins
// are several times that, and the float forms of w are nudged two more ulps towards "pass").  A query the algebra
// does not cover (qs <= 0, a NaN anywhere) gets w = +inf: every tile takes the exact path for it.  An unused query
// slot (thr = -3e38) gets w = -inf.
template <int METRIC>
__device__ __forceinline__ void pretest_consts(float thr, float qs, float qn, float *ps, float *pw)
{
    float s = 0.0f, w;
    if (thr <= -3.0e38f) {
        w = -__builtin_inff();
    } else if (METRIC == kCosine) {
        const double T = (-(double)thr - 4.0e-7 * fabs((double)thr)) / (double)qs;
        w = (float)(-T);
        w += fabsf(w) * 2.4e-7f + 1.0e-37f;
        if (!(qs > 0.0f) || w != w) w = __builtin_inff();
    } else {
        const double V = (double)qn - (double)thr - 2.0e-6 * (fabs((double)qn) + fabs((double)thr));
        s = 2.0f * qs;
        w = (float)(-V);
        w += fabsf(w) * 2.4e-7f + 1.0e-37f;
        if (!(qs > 0.0f) || !(qn >= 0.0f) || w != w) w = __builtin_inff();
    }
    *ps = s;
    *pw = w;
}

// Waves per CU and ring depth (16-byte loads per lane in flight; divides STEPS) of the shape kernels.  768-byte rows
// (12 steps): 8 waves with 6 KiB each in flight -- 0.122 ms per 1M-row pass against 0.134 with 12 x 4, fewer waves
// queueing behind one another's tile finish.  Shorter rows have a finish per fewer bytes and want the 12 waves
// (384 bytes: 0.069 against 0.075; 192: 0.046 against 0.052), and so do 4-bit rows with twice the arithmetic per byte.
#ifndef SZG_S12_WAVES
#define SZG_S12_WAVES 8
#endif
#ifndef SZG_S12_RING
#define SZG_S12_RING 6
#endif
#ifndef SZG_S6_RING
#define SZG_S6_RING 3
#endif
#ifndef SZG_I8S_RN
#define SZG_I8S_RN 1  // the shape kernels take the rows' norms from the resident array (MqArgs::row_norm) instead of summing them
#endif
template <int RB, int STEPS>
constexpr int i8s_waves()
{   // (4-bit rows of 12 steps -- 1 536 dims -- at 12 waves per CU spilled 2-4 of their 168 registers: 8 waves, 256)
    return STEPS == 12 ? SZG_S12_WAVES : SZG_MQ8_WAVES;
}
template <int RB, int STEPS>
constexpr int i8s_ring()
{
    if (RB == 8 && STEPS == 12) return SZG_S12_RING;
    if (RB == 8 && STEPS == 6) return SZG_S6_RING;
#ifdef SZG_S6R4_RING
    if (RB == 4 && STEPS == 6) return SZG_S6R4_RING;
#endif
    return STEPS % 4 == 0 ? 4 : (STEPS % 3 == 0 ? 3 : (STEPS % 2 == 0 ? 2 : 1));
}
template <int NB, int METRIC, int RB, int STEPS>
__global__ __launch_bounds__((64 * i8s_waves<RB, STEPS>())) void mq_score_i8s_kernel(const MqArgs a)
{
    constexpr int T = RB == 4 ? 2 : 1;
    constexpr int NPL = kMqPlanes;
    constexpr int D = i8s_ring<RB, STEPS>();
    static_assert(NPL == 2, "the integer plane combine in the tile finish assumes two digit planes");
    constexpr int QSTEP = NPL * T * NB * 64;  // 16-byte words of the image per 64-byte step
    constexpr int N16 = STEPS * QSTEP;
    constexpr bool RN = SZG_I8S_RN != 0;
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const RowLayout mlay{a.pitch, a.tiled, a.steps};
    const uint32_t istep = a.tiled ? 1024u : 64u;
    const int n_groups = a.n_groups > 0 ? a.n_groups : 1;
    constexpr size_t grp_lds = (size_t)N16 * 16 + kMq8TableRows * 48 * sizeof(float);  // image | qscale, qconst, qnorm2 | thresholds, pre-test s, w
    for (int g = 0; g < n_groups; g++) {
        const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.queries) +
                                                           (size_t)g * a.group_stride);
        uint4 *dst = reinterpret_cast<uint4 *>(smem + (size_t)g * grp_lds);
        constexpr int n = N16 + (3 * 48 * 4) / 16;  // + constants table
        stage_image(dst, src, n, tid, blockDim.x);
        if (tid < 48) {
            float *tab = reinterpret_cast<float *>(smem + (size_t)g * grp_lds + (size_t)n * 16);
            const float thr = g * 48 + tid < a.n_queries ? a.thr[g * 48 + tid] : -3.0e38f;
            const float *qconsts = reinterpret_cast<const float *>(src + N16);  // qscale | qconst | qnorm2
            float ps, pw;
            pretest_consts<METRIC>(thr, qconsts[tid], qconsts[96 + tid], &ps, &pw);
            tab[tid] = thr;
            tab[48 + tid] = ps;
            tab[96 + tid] = pw;
        }
    }
    const int trow = lane & 15;
    const int c = lane >> 4;
    const uint64_t n_tiles = ((uint64_t)a.n_rows + 15) / 16;
    const uint64_t tile_stride = (uint64_t)gridDim.x * nwaves;
    const uint64_t tile_first = (uint64_t)blockIdx.x * nwaves + wave;
    const uint64_t n_it = tile_first < n_tiles ? (n_tiles - tile_first + tile_stride - 1) / tile_stride : 0;
    auto row_ptr = [&](uint64_t tile) -> const uint8_t * {
        const uint64_t r = min(tile * 16 + trow, (uint64_t)a.n_rows - 1);  // past the end: a valid row, discarded
        return a.rows + piece_offset(mlay, r, (uint32_t)c);
    };
    HitBuf hb;
    {
        uint8_t *base = smem + (size_t)n_groups * grp_lds;
        hb.cand = reinterpret_cast<uint64_t *>(base) + (size_t)wave * kHitCap;
        hb.query = base + (size_t)nwaves * kHitCap * 8 + (size_t)wave * kHitCap;
        hb.n = 0;
    }

    for (int grp = 0; grp < n_groups; grp++) {
        const int qoff = grp * 48;
        const uint8_t *gbase = smem + (size_t)grp * grp_lds;
        const v4i32 *qimg = reinterpret_cast<const v4i32 *>(gbase) + lane;
        const float *qtab = reinterpret_cast<const float *>(gbase + (size_t)N16 * 16);
        const float *thr_lds = qtab + 3 * 48;
        u32x4 ring[D];
        v4i32 acc[NPL][NB];
#pragma unroll
        for (int p = 0; p < NPL; p++)
#pragma unroll
            for (int b = 0; b < NB; b++) acc[p][b] = v4i32{0, 0, 0, 0};
        int SQ = 0, SV = 0;
        uint64_t tile = tile_first;
        const uint8_t *cur = row_ptr(tile);
        // the ring's first D steps (D <= STEPS: all inside the first tile)
#pragma unroll
        for (int u = 0; u < D; u++) {
            ring[u] = load_stream<true>(cur + (size_t)u * istep);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 0) __syncthreads();  // the query images are complete (the rows do not depend on them)
        // The A operands travel one PHASE ahead of the matrix instructions that use them.  A phase is the G = NPL x NB
        // operands of one (step, nibble half); while the G MFMAs of phase ph issue (G x 16 cycles), the G ds_read_b128
        // of phase ph + 1 are in flight into the other half of a double buffer.  Round 3's form left the reads to the
        // compiler, which issued each one or two instructions ahead of the MFMA that needs it (84 reads, 72 MFMAs and
        // an `s_waitcnt lgkmcnt` before nearly every one of them in the 4-bit 6-step kernel): every wave paid the LDS
        // latency once per couple of MFMAs and the other two waves of its SIMD were all that hid it.  sched_barriers pin
        // the order read-group / MFMA-group; the decode of the row bytes shares the MFMA groups' regions, where the
        // scheduler slots it into the matrix instructions' shadows.  (Needs an even number of phases per tile, so that
        // the buffer halves are compile-time facts: 8-bit rows of 3 steps keep the plain form.)  Same box, 1M x 768
        // 4-bit: 0.092 -> 0.084 ms per pass.
        constexpr int G = NPL * NB, PHASES = STEPS * T;
        constexpr bool PIPE = PHASES % 2 == 0;
        v4i32 qbuf[2][G];
        auto read_phase = [&](int ph, v4i32 (&dst)[G]) {
            const int st_ = ph / T, t_ = ph % T;
#pragma unroll
            for (int p = 0; p < NPL; p++)
#pragma unroll
                for (int b = 0; b < NB; b++) dst[p * NB + b] = qimg[st_ * QSTEP + ((p * T + t_) * NB + b) * 64];
        };
        if (PIPE) read_phase(0, qbuf[0]);
        for (uint64_t it = 0; it < n_it; it++, tile += tile_stride) {
            // (past the wave's last tile: its own tile again -- D loads nobody consumes)
            const uint8_t *nxt = it + 1 < n_it ? row_ptr(tile + tile_stride) : cur;
            // resident norms: the tile's 16 arrive while its steps run (the decode below then spends nothing on them:
            // 12 of its 24 vector instructions per 64-byte step of 4-bit rows, 8 of 12 for 8-bit rows)
            float norm_res = 0.f;
            if constexpr (RN) norm_res = a.row_norm[min(tile * 16 + trow, (uint64_t)a.n_rows - 1)];
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const u32x4 v_ = ring[st % D];
                // this slot's next load: the step D ahead, in this tile or the next
                ring[st % D] = st + D < STEPS ? load_stream<true>(cur + (size_t)(st + D) * istep)
                                              : load_stream<true>(nxt + (size_t)(st + D - STEPS) * istep);
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t raw_[4] = {v_.x, v_.y, v_.z, v_.w};
                v4i32 bop_[T];
                if constexpr (PIPE) {
                    // region 0: the first operand of the step, and the NEXT phase's reads
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        bop_[0][d] = RB == 8 ? (int)(raw_[d] ^ 0x80808080u) : (int)((raw_[d] >> 4) & 0x0F0F0F0Fu);
                    read_phase((st * T + 1) % PHASES, qbuf[(st * T + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // region 1: G MFMAs of phase st * T, with the rest of the decode in their shadows
                    // (the tile's first matrix instructions start from a literal zero: no accumulator clearing per tile)
#pragma unroll
                    for (int g = 0; g < G; g++)
                        acc[g / NB][g % NB] = __builtin_amdgcn_mfma_i32_16x16x64_i8(
                            qbuf[(st * T) & 1][g], bop_[0], st == 0 ? v4i32{0, 0, 0, 0} : acc[g / NB][g % NB], 0, 0, 0);
                    if constexpr (RB == 8) {
                        if constexpr (!RN) {
#pragma unroll
                            for (int d = 0; d < 4; d++) {
                                SQ = __builtin_amdgcn_sdot4(bop_[0][d], bop_[0][d], SQ, false);
                                SV = __builtin_amdgcn_sdot4(bop_[0][d], 0x01010101, SV, false);
                            }
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; d++) bop_[T - 1][d] = (int)(raw_[d] & 0x0F0F0F0Fu);
                        __builtin_amdgcn_sched_barrier(0);
                        // region 2: the reads of the phase after next;  region 3: the low nibbles' MFMAs + the norm
                        read_phase((st * T + 2) % PHASES, qbuf[(st * T + 2) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int g = 0; g < G; g++)
                            acc[g / NB][g % NB] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qbuf[(st * T + 1) & 1][g], bop_[T - 1],
                                                                                      acc[g / NB][g % NB], 0, 0, 0);
                        if constexpr (!RN) {
#pragma unroll
                            for (int d = 0; d < 4; d++) {
                                const int wn_ = (int)(raw_[d] ^ 0x88888888u);
                                SQ = __builtin_amdgcn_sdot8(wn_, wn_, SQ, false);
                                SV = __builtin_amdgcn_sdot8(wn_, 0x11111111, SV, false);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        if (RB == 8) {
                            const int wn_ = (int)(raw_[d] ^ 0x80808080u);
                            bop_[0][d] = wn_;
                            if constexpr (!RN) {
                                SQ = __builtin_amdgcn_sdot4(wn_, wn_, SQ, false);
                                SV = __builtin_amdgcn_sdot4(wn_, 0x01010101, SV, false);
                            }
                        } else {
                            const int wn_ = (int)(raw_[d] ^ 0x88888888u);
                            bop_[0][d] = (int)((raw_[d] >> 4) & 0x0F0F0F0Fu);
                            bop_[T - 1][d] = (int)(raw_[d] & 0x0F0F0F0Fu);
                            if constexpr (!RN) {
                                SQ = __builtin_amdgcn_sdot8(wn_, wn_, SQ, false);
                                SV = __builtin_amdgcn_sdot8(wn_, 0x11111111, SV, false);
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < T; t++)
#pragma unroll
                        for (int p = 0; p < NPL; p++)
#pragma unroll
                            for (int b = 0; b < NB; b++) {
                                const v4i32 qc_ = qimg[st * QSTEP + ((p * T + t) * NB + b) * 64];
                                acc[p][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(
                                    qc_, bop_[t], st == 0 && t == 0 ? v4i32{0, 0, 0, 0} : acc[p][b], 0, 0, 0);
                            }
                }
            }
            // ---- the tile is done: row norms across the 4 chunk lanes, then the hit test in two stages.  On a large
            // shard a tile of 16 rows x 48 queries holds a hit a few times in a hundred (a radius batch: far less), so
            // every tile pays only a PRE-TEST of ~4 VALU instructions per (row, query) -- one integer combine, one
            // convert, two fmas, a running max -- against per-query constants staged with a safety margin
            // (pretest_consts), and only a tile in which some lane passes it forms the keys proper and tests them
            // against the thresholds.  The pre-test passes whenever key <= thr would (the same inequality solved for
            // the integer dot product, the rounding of the key's float chain covered by the margin), so the hits are
            // exactly the one-stage test's.  (1M rows with the default 1 024 expected hits per query: more than half
            // the tiles hold a hit and the two-stage form measures the same as the one-stage form; 12.5M rows: see
            // profiles/r03_i8_sweep_experiments.txt, section 11.)
            float norm;
            if constexpr (RN) {
                norm = norm_res;
            } else {
                int nrm = 4 * (SQ + SV);
                nrm += __shfl_xor(nrm, 16);
                nrm += __shfl_xor(nrm, 32);
                norm = (float)nrm + a.norm_bias;
            }
            const float inv = __frsqrt_rn(norm);
            const uint64_t row = tile * 16 + trow;
            const bool row_ok = row < a.n_rows;
            // (d2 = sum Q n of a pair, the float its key is made of, is formed again in the rare second stage rather than
            // kept: twelve registers that the prefetched operands of the next tile need more)
            auto pair_d2 = [&](int b, int r, float qc) -> float {
                // planes combined as integers: |plane sums| < 2^24 and |dot| < 2^31 for these row shapes, so the one
                // conversion rounds exactly as fmaf(128, float(acc0), float(acc1)) does (the generic kernel's form,
                // which the prefix pass made the thresholds with)
                int di = acc[0][b][r];
#pragma unroll
                for (int p = 1; p < NPL; p++) di = di * 128 + acc[p][b][r];
                return fmaf(2.0f, (float)di, qc);
            };
            float best = -__builtin_inff();
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int q0 = b * 16 + c * 4;
                const float4 qc4 = *reinterpret_cast<const float4 *>(qtab + 48 + q0);
                const float4 ps4 = *reinterpret_cast<const float4 *>(thr_lds + 48 + q0);
                const float4 pw4 = *reinterpret_cast<const float4 *>(thr_lds + 96 + q0);
                const float qcv[4] = {qc4.x, qc4.y, qc4.z, qc4.w};
                const float psv[4] = {ps4.x, ps4.y, ps4.z, ps4.w}, pwv[4] = {pw4.x, pw4.y, pw4.z, pw4.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float g = pair_d2(b, r, qcv[r]);
                    const float e = METRIC == kCosine ? fmaf(g, inv, pwv[r]) : fmaf(g, psv[r], pwv[r]);
                    best = fmaxf(best, e);
                }
            }
            const bool pass = row_ok && best >= (METRIC == kCosine ? 0.0f : norm * (1.0f - 2.0e-6f));
            if (__ballot(pass)) {
                float keys[NB][4];
                uint32_t hm = 0;
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const int q0 = b * 16 + c * 4;
                    const float4 qs4 = *reinterpret_cast<const float4 *>(qtab + q0);
                    const float4 qc4 = *reinterpret_cast<const float4 *>(qtab + 48 + q0);
                    const float4 qn4 = METRIC == kCosine ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                         : *reinterpret_cast<const float4 *>(qtab + 96 + q0);
                    const float4 th4 = *reinterpret_cast<const float4 *>(thr_lds + q0);
                    const float qsv[4] = {qs4.x, qs4.y, qs4.z, qs4.w}, qcv[4] = {qc4.x, qc4.y, qc4.z, qc4.w};
                    const float qnv[4] = {qn4.x, qn4.y, qn4.z, qn4.w}, thv[4] = {th4.x, th4.y, th4.z, th4.w};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float d2 = pair_d2(b, r, qcv[r]);
                        float key;
                        if (METRIC == kCosine)
                            key = -(d2 * qsv[r]) * inv;
                        else
                            key = fmaf(-2.0f * qsv[r], d2, qnv[r] + norm);
                        keys[b][r] = key;
                        hm |= (row_ok & (key <= thv[r])) ? (1u << (b * 4 + r)) : 0u;  // unused queries: thr = -3e38
                    }
                }
                offer_tile_hits<NB>(a, hb, lane, c, hm, keys, row, qoff);
            }
            SQ = 0;
            SV = 0;
            cur = nxt;
        }
    }
    // one flush for both groups (the buffered query index carries the group): a flush is a returning atomic per hit
    // and a drained load queue -- a memory round trip with nothing in flight, which at the end of every pass cost 3 %
    hit_flush(a, hb, lane);
}

#endif  // SZG_MQ_PART == 1 || 2

#if SZG_MQ_PART == 0
// ---- per-query selection over the score matrix ----------------------------------

// grid (blocks per query, queries).  Each lane reads 4 keys at a time (16 bytes);
// a key that beats the wave's current kp-th best is inserted into the wave's list.
__global__ __launch_bounds__(1024) void mq_select_kernel(const float *keys, size_t key_stride,
                                                        uint32_t n_rows, const uint64_t *live_bits,
                                                        const uint64_t *allow_bits,
                                                        uint32_t allow_stride, int kp,
                                                        uint64_t *block_lists, float *thr_out,
                                                        uint32_t *count_zero)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int q = blockIdx.y;
    WaveList wl;
    wl.init(lists + (size_t)wave * kp, kp, lane);
    __syncthreads();
    const float *kq = keys + (size_t)q * key_stride;
    const uint64_t *allow = allow_bits ? allow_bits + (size_t)q * allow_stride : nullptr;
    const uint32_t n4 = (n_rows + 3) / 4;  // key_stride is a multiple of 4, the tail holds +inf
    const uint32_t stride = gridDim.x * blockDim.x;
    constexpr int U = 4;  // loads in flight per thread (one block per query walks its keys: latency-bound)
    for (uint32_t i0 = blockIdx.x * blockDim.x + tid; i0 < ((n4 + stride - 1) / stride) * stride; i0 += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = i0 + u * stride;
            v[u] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f);
            if (i < n4) v[u] = reinterpret_cast<const float4 *>(kq)[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = i0 + u * stride;
            if (i >= ((n4 + stride - 1) / stride) * stride) break;  // (uniform over the block)
            const float kk[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t row = i * 4 + e;
                bool ok = i < n4 && row < n_rows;
                if (ok && live_bits) ok = (live_bits[row >> 6] >> (row & 63)) & 1;
                if (ok && allow) ok = (allow[row >> 6] >> (row & 63)) & 1;
                const uint64_t cnd = ((uint64_t)ordered_key(kk[e]) << 32) | row;
                wl.offer(ok, cnd, lane);
            }
        }
    }
    wl.flush(lane);
    __syncthreads();
    uint64_t *out = block_lists + ((size_t)q * gridDim.x + blockIdx.x) * kp;
    block_merge_lists(lists, nwaves, kp, out, tid, blockDim.x);
    if (thr_out) {  // one block per query: its kp-th key is the query's collect threshold
        __syncthreads();
        if (tid == 0) {
            const uint64_t c = out[kp - 1];
            thr_out[q] = c == kInvalidCand ? 3.0e38f : key_from_ordered((uint32_t)(c >> 32));
            count_zero[q * kCandCountStride] = 0;
        }
    }
}

// thr[q] = key of the kp-th entry of query q's sorted candidate list
__global__ void mq_thr_kernel(const uint64_t *lists, int kp, int n_queries, float *thr)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    const uint64_t c = lists[(size_t)q * kp + (kp - 1)];
    thr[q] = c == kInvalidCand ? 3.0e38f : key_from_ordered((uint32_t)(c >> 32));
}

// one block per query: the kp best of the query's candidate buffer, sorted ascending
__global__ __launch_bounds__(256) void cand_select_kernel(const uint64_t *cand_buf, const uint32_t *cand_count,
                                                          uint32_t cand_cap, int kp, uint64_t *lists)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t *wl_lds = reinterpret_cast<uint64_t *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int q = blockIdx.x;
    WaveList wl;
    wl.init(wl_lds + (size_t)wave * kp, kp, lane);
    __syncthreads();
    const uint32_t n = min(cand_count[q * kCandCountStride], cand_cap);
    const uint64_t *src = cand_buf + (size_t)q * cand_cap;
    for (uint32_t i = tid; i < ((n + blockDim.x - 1) / blockDim.x) * blockDim.x; i += blockDim.x) {
        const bool ok = i < n;
        const uint64_t c = ok ? src[i] : kInvalidCand;
        wl.offer(ok, c, lane);
    }
    wl.flush(lane);
    __syncthreads();
    block_merge_lists(wl_lds, nwaves, kp, lists + (size_t)q * kp, tid, blockDim.x);
}


// Second stage of the bfloat16 sweep: the (few thousand) candidates it collected are scored again in
// float32 -- one wave per (query, candidate), the query as float32 in LDS -- and the key inside the
// candidate word is replaced, so that the selection and the certification that follow work with
// float32 keys (bound: key_eps, mq branch).  grid (blocks, queries); 32-bit rows, any dim.
// One 16-byte piece of a 32- or 16-bit row against the float32 query staged in LDS: 4 floats, or 8 codes decoded to
// n = 2v - 65535.  COS: dot, norm and the zero-row bits; else the squared difference (into dot).
template <bool COS>
__device__ __forceinline__ void rescore_use(const uint4 w, const float *qf, int piece, int row_bits, int dim,
                                            float &dot, float &nrm, uint32_t &nz)
{
    if (row_bits == 8) {  // sixteen codes, n = 2v - 255 (exact); the last piece's padding codes are not part of the row
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
        const int n8 = min(16, dim - piece * 16);
        const float *y8 = qf + (size_t)piece * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i < n8) {
                const float xv = fmaf((float)((ww[i >> 2] >> (8 * (i & 3))) & 0xFFu), 2.0f, -255.0f);
                if (COS) {
                    dot = fmaf(xv, y8[i], dot);
                    nrm = fmaf(xv, xv, nrm);
                } else {
                    const float d = xv - y8[i];
                    dot = fmaf(d, d, dot);
                }
            }
        }
        nz |= 1u;
        return;
    }
    float x[8];
    int n;
    if (row_bits == 16) {
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            x[2 * i] = fmaf((float)(ww[i] & 0xFFFFu), 2.0f, -65535.0f);
            x[2 * i + 1] = fmaf((float)(ww[i] >> 16), 2.0f, -65535.0f);
        }
        n = min(8, dim - piece * 8);  // (the last piece's padding codes decode to -65535: not part of the row)
        nz |= 1u;
    } else if (row_bits == 64) {  // two float64 elements, narrowed to float32 as the sweep narrows them
        x[0] = (float)__hiloint2double((int)w.y, (int)w.x);
        x[1] = (float)__hiloint2double((int)w.w, (int)w.z);
        x[2] = x[3] = x[4] = x[5] = x[6] = x[7] = 0.f;
        n = 2;
        nz |= ((w.y | w.w) & 0x7FFFFFFFu) | w.x | w.z;
    } else {
        x[0] = __uint_as_float(w.x); x[1] = __uint_as_float(w.y); x[2] = __uint_as_float(w.z); x[3] = __uint_as_float(w.w);
        x[4] = x[5] = x[6] = x[7] = 0.f;
        n = 4;
        nz |= (w.x | w.y | w.z | w.w) & 0x7FFFFFFFu;
    }
    const float *y = qf + (size_t)piece * (row_bits == 16 ? 8 : (row_bits == 64 ? 2 : 4));
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < n) {
            if (COS) {
                dot = fmaf(x[i], y[i], dot);
                nrm = fmaf(x[i], x[i], nrm);
            } else {
                const float d = x[i] - y[i];
                dot = fmaf(d, d, dot);
            }
        }
    }
}
__device__ __forceinline__ int rescore_epp(int row_bits)  // elements per 16-byte piece
{
    return row_bits == 8 ? 16 : (row_bits == 16 ? 8 : (row_bits == 64 ? 2 : 4));
}
template <bool COS>
__device__ __forceinline__ void rescore_piece(const uint8_t *rows, const RowLayout &lay, uint32_t row, const float *qf, int piece,
                                              int row_bits, int dim, float &dot, float &nrm, uint32_t &nz)
{
    rescore_use<COS>(*reinterpret_cast<const uint4 *>(rows + piece_offset(lay, row, (uint32_t)piece)), qf, piece, row_bits, dim,
                     dot, nrm, nz);
}

template <int METRIC>
__global__ __launch_bounds__(256) void cand_rescore_kernel(const uint8_t *rows, RowLayout lay, int dim,
                                                           const double *q64, const double *qscale,
                                                           uint64_t *cand_buf, const uint32_t *cand_count,
                                                           uint32_t cand_cap, int row_bits)
{
    extern __shared__ __align__(16) uint8_t smem[];
    float *qf = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.y;
    const uint32_t n = min(cand_count[q * kCandCountStride], cand_cap);
    const double sc = qscale[q];
    // whole 16-byte pieces: a 32-bit row's padding is stored as zeros, the query's staged as zeros (16-bit rows come
    // here with whole pieces only)
    const int epp = rescore_epp(row_bits), pieces = (dim + epp - 1) / epp;
    for (int i = tid; i < epp * pieces; i += blockDim.x) qf[i] = i < dim ? (float)(q64[(size_t)q * dim + i] * sc) : 0.0f;
    __syncthreads();
    uint64_t *cb = cand_buf + (size_t)q * cand_cap;
    for (uint32_t ci = blockIdx.x * 4 + wave; ci < n; ci += gridDim.x * 4) {
        const uint32_t row = (uint32_t)cb[ci];
        float dot = 0.f, nrm = 0.f;
        uint32_t nz = 0;
        for (int i = lane; i < pieces; i += 64) rescore_piece<METRIC == kCosine>(rows, lay, row, qf, i, row_bits, dim, dot, nrm, nz);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dot += __shfl_xor(dot, o);
            if (METRIC == kCosine) {
                nrm += __shfl_xor(nrm, o);
                nz |= __shfl_xor(nz, o);
            }
        }
        float key;
        if (METRIC == kCosine) {
            key = -dot * __frsqrt_rn(nrm);
            if (nrm == 0.f) key = nz ? -2.0f : 1.0f;
            if (!(nrm <= 3.0e38f)) key = -2.0f;  // norm overflow: forced in (see RowAcc::finish)
        } else {
            key = dot;
        }
        if (!(key == key)) key = 3.0e38f;
        if (key > 3.0e38f) key = 3.0e38f;
        if (lane == 0) cb[ci] = ((uint64_t)ordered_key(key) << 32) | row;
    }
}


// ---- thresholds of the fused selection: a radix select, not a sort ----------------------------------------------------
//
// The threshold pass only needs ONE number per query: a key thr such that at least kp eligible prefix rows have
// key <= thr, as small as cheaply possible.  Two histogram rounds over the ordered key's top 12 + 12 bits (LDS
// atomics, one block of 1024 threads per query) give the kp-th smallest key to 2^-16 relative -- rounded UP, so the
// kp-th key itself always passes.  The sorted-list selection this replaces (mq_select_kernel, one block per query)
// spent ~50 us per 96-query batch inserting into lists nobody read.
__global__ __launch_bounds__(1024) void mq_thr_radix_kernel(const float *keys, size_t key_stride, uint32_t n_rows,
                                                            const uint64_t *live_bits, const uint64_t *allow_bits,
                                                            uint32_t allow_stride, int kp, float *thr_out,
                                                            uint32_t *count_zero)
{
    __shared__ uint32_t hist[4096];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t sel_bin, sel_below;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const float4 *kq = reinterpret_cast<const float4 *>(keys + (size_t)q * key_stride);
    const uint64_t *allow = allow_bits ? allow_bits + (size_t)q * allow_stride : nullptr;
    const uint32_t n4 = (n_rows + 3) / 4;  // key_stride is a multiple of 4
    uint32_t prefix_bits = 0;              // the bins chosen so far (top bits of the ordered key)
    uint32_t below = 0;                    // eligible keys below the chosen bins
    for (int round = 0; round < 2; round++) {
        for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
        __syncthreads();
        const int shift = round == 0 ? 20 : 8;
        for (uint32_t i = tid; i < n4; i += 1024) {
            const float4 v = kq[i];
            const float kk[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t row = i * 4 + e;
                bool ok = row < n_rows;
                if (ok && live_bits) ok = (live_bits[row >> 6] >> (row & 63)) & 1;
                if (ok && allow) ok = (allow[row >> 6] >> (row & 63)) & 1;
                const uint32_t u = ordered_key(kk[e]);
                if (ok && (round == 0 || (u >> 20) == prefix_bits)) atomicAdd(&hist[(u >> shift) & 0xFFFu], 1u);
            }
        }
        __syncthreads();
        // the bin where the running count reaches kp: 4 bins per thread, scan over the wave, then over the 16 waves
        const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        const uint32_t mine = h0 + h1 + h2 + h3;
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        if (tid == 0) sel_bin = 0xFFFFFFFFu;
        __syncthreads();
        uint32_t before = below;
        for (int w = 0; w < wave; w++) before += wsum[w];
        const uint32_t excl = before + incl - mine;  // eligible keys before this thread's 4 bins (chosen bins included)
        if (excl < (uint32_t)kp && excl + mine >= (uint32_t)kp) {  // exactly one thread
            uint32_t acc = excl;
            const uint32_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (acc < (uint32_t)kp && acc + hh[b] >= (uint32_t)kp) {
                    sel_bin = 4 * tid + b;
                    sel_below = acc;
                }
                acc += hh[b];
            }
        }
        __syncthreads();
        if (sel_bin == 0xFFFFFFFFu) break;  // fewer than kp eligible keys: everything passes
        prefix_bits = round == 0 ? sel_bin : ((prefix_bits << 12) | sel_bin);
        below = sel_below;
        __syncthreads();
    }
    if (tid == 0) {
        float thr = 3.0e38f;
        if (sel_bin != 0xFFFFFFFFu) {
            thr = key_from_ordered((prefix_bits << 8) | 0xFFu);  // the top of the chosen 24-bit bin
            if (!(thr <= 3.0e38f)) thr = 3.0e38f;                // (NaN patterns sort last: keys are clamped anyway)
        }
        thr_out[q] = thr;
        count_zero[q * kCandCountStride] = 0;
    }
}

// ---- the tail of a fused-selection batch in ONE launch ---------------------------------------------------------------
//
// One block of 1024 threads per query, the query's collected candidates (<= kRefineMaxCands) in LDS:
//   1. the kp best by the sweep's key (per-wave lists + block rank-merge, as everywhere);
//   2. MODE > 0, bfloat16 sweeps: the sweep's key b of a candidate is within eps_b of its real-number key, so only
//      candidates with b <= t_kp + W (t_kp = the kp-th best sweep key, W = 2 eps_b) can be among the kp best by a
//      better key.  Those -- a few dozen of the ~1 000 collected -- are scored again in float32 (one wave per
//      candidate, the float32 query in LDS) and the kp best of THEM by float32 key are the list; the band's edge
//      E = t_kp + W goes to the host: every collected candidate outside the band has sweep key > E, which
//      certification needs (scan_topk.cpp: gather_topk).  Re-scoring every candidate, as the first form of this
//      stage did, cost 64 us per 96-query batch; the band costs a tenth.
//   3. the query's sentinel rows (its first k eligible rows in visit order, whatever their key: DESIGN.md 2) are
//      appended behind the list, so ONE rerank launch computes the float64 distances of both.
// MODE 0: selection only (int8 / float32 sweeps: the collected keys are final).  1: cosine band, 2: euclid band.
// A band that does not fit kRefineMaxBand (duplicate-heavy corpora) reports overflow through the hit counter, which
// sends the batch down the score-matrix path like an overflowing candidate buffer.
constexpr int kRefineThreads = 1024;
constexpr int kRefineMaxCands = 8192;
constexpr int kRefineMaxBand = 1024;
constexpr int kRefineMaxKp = 256;

template <int MODE>
__global__ __launch_bounds__(kRefineThreads) void cand_refine_kernel(const uint8_t *rows, RowLayout lay, int dim,
                                                                     const double *q64, const double *qscale,
                                                                     const double *qnorm2, const uint64_t *cand_buf,
                                                                     uint32_t *cand_count, uint32_t cand_cap, int kp,
                                                                     const uint64_t *sent, int n_sent, uint64_t *lists,
                                                                     float *band_edge, int row_bits)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kRefineThreads / 64;
    const int q = blockIdx.x;
    const uint32_t n = min(cand_count[q * kCandCountStride], cand_cap);
    uint64_t *cand = reinterpret_cast<uint64_t *>(smem);                  // [kRefineMaxCands]
    uint64_t *wl_lds = cand + kRefineMaxCands;                            // [NW][kp]
    uint64_t *top = wl_lds + (size_t)NW * kp;                             // [kp]
    uint64_t *band = top + kp;                                            // [kRefineMaxBand]
    float *qf = reinterpret_cast<float *>(band + kRefineMaxBand);         // [dim] (MODE > 0)
    __shared__ uint32_t n_band;
    const uint64_t *src = cand_buf + (size_t)q * cand_cap;
    uint64_t *out = lists + (size_t)q * (kp + n_sent);

    for (uint32_t i = tid; i < n; i += kRefineThreads) cand[i] = src[i];
    if (MODE > 0) {
        const double sc = qscale[q];
        const int epp0 = rescore_epp(row_bits);
        for (int i = tid; i < epp0 * ((dim + epp0 - 1) / epp0); i += kRefineThreads)
            qf[i] = i < dim ? (float)(q64[(size_t)q * dim + i] * sc) : 0.0f;  // (padding: zeros, as in the rows)
    }
    if (tid == 0) n_band = 0;
    float edge = 3.0e38f;  // fewer than kp candidates: everything collected is in the band
    if (MODE == 0) {
        WaveList wl;
        wl.init(wl_lds + (size_t)wave * kp, kp, lane);
        __syncthreads();
        for (uint32_t i = tid; i < ((n + kRefineThreads - 1) / kRefineThreads) * kRefineThreads; i += kRefineThreads) {
            const bool ok = i < n;
            wl.offer(ok, ok ? cand[i] : kInvalidCand, lane);
        }
        wl.flush(lane);
        __syncthreads();
        block_merge_lists(wl_lds, NW, kp, out, tid, kRefineThreads);
        for (int i = tid; i < n_sent; i += kRefineThreads) out[kp + i] = sent[(size_t)q * n_sent + i];
        return;
    }
    for (int i = tid; i < n_sent; i += kRefineThreads) out[kp + i] = sent[(size_t)q * n_sent + i];
    // The band needs ONE number of the sweep's keys: the kp-th smallest, t_kp (any value at or above it serves: the band
    // only grows).  Two histogram rounds over the ordered key's top 12 + 12 bits (as mq_thr_radix_kernel) instead of
    // round 3's sixteen sorted per-wave lists and their rank merge, which nothing else read: 60 -> ~30 us per batch.
    {
        uint32_t *hist = reinterpret_cast<uint32_t *>(qf + ((dim + 15) & ~15));  // [4096]
        __shared__ uint32_t wsum[NW];
        __shared__ uint32_t sel_bin, sel_below;
        uint32_t prefix_bits = 0, below = 0;
        bool found = n >= (uint32_t)kp;
        for (int round = 0; round < 2 && found; round++) {
            for (int i = tid; i < 4096; i += kRefineThreads) hist[i] = 0;
            __syncthreads();
            const int shift = round == 0 ? 20 : 8;
            for (uint32_t i = tid; i < n; i += kRefineThreads) {
                const uint32_t u = (uint32_t)(cand[i] >> 32);
                if (round == 0 || (u >> 20) == prefix_bits) atomicAdd(&hist[(u >> shift) & 0xFFFu], 1u);
            }
            __syncthreads();
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            if (tid == 0) sel_bin = 0xFFFFFFFFu;
            __syncthreads();
            uint32_t before = below;
            for (int w = 0; w < wave; w++) before += wsum[w];
            const uint32_t excl = before + incl - mine;
            if (excl < (uint32_t)kp && excl + mine >= (uint32_t)kp) {  // exactly one thread
                uint32_t acc = excl;
                const uint32_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if (acc < (uint32_t)kp && acc + hh[b] >= (uint32_t)kp) {
                        sel_bin = 4 * tid + b;
                        sel_below = acc;
                    }
                    acc += hh[b];
                }
            }
            __syncthreads();
            if (sel_bin == 0xFFFFFFFFu) {
                found = false;  // (cannot happen with n >= kp; uniform over the block)
            } else {
                prefix_bits = round == 0 ? sel_bin : ((prefix_bits << 12) | sel_bin);
                below = sel_below;
            }
            __syncthreads();
        }
        if (found) {
            float t = key_from_ordered((prefix_bits << 8) | 0xFFu);  // the top of the kp-th key's 24-bit bin
            if (!(t <= 3.0e38f)) t = 3.0e38f;
            const float c = 1.01f * 0x1p-7f, nu = ((float)dim + (row_bits == 64 ? 20.0f : 16.0f)) * 0x1p-24f;  // key_eps (scan_query.cpp), bfloat16 branch
            float eps;
            if (MODE == 1) {
                eps = c + 4.0f * nu + 1e-6f;
            } else {  // key_eps (scan_query.cpp), bfloat16 euclid branch
                const float qn = sqrtf((float)qnorm2[q]), rt = sqrtf(fmaxf(t, 0.0f));
                const float s2 = 2.0f * qn + rt;
                eps = 2.0f * c * qn * (1.1f * qn + rt) + c * c * qn * qn + 3.0f * nu * s2 * s2;
            }
            edge = t + 2.02f * eps;
            if (!(edge < 3.0e38f)) edge = 3.0e38f;
        }
    }
    const uint32_t uedge = ordered_key(edge);
    for (uint32_t i = tid; i < n; i += kRefineThreads) {
        if ((uint32_t)(cand[i] >> 32) <= uedge) {
            const uint32_t at = atomicAdd(&n_band, 1u);
            if (at < (uint32_t)kRefineMaxBand) band[at] = cand[i];
        }
    }
    __syncthreads();
    const uint32_t nb = n_band;
    if (nb > (uint32_t)kRefineMaxBand) {  // (uniform over the block)
        if (tid == 0) cand_count[q * kCandCountStride] = 0xFFFFFFFFu;  // the host redoes the batch (score matrix)
        for (int i = tid; i < kp; i += kRefineThreads) out[i] = kInvalidCand;
        return;
    }
    // float32 keys for the band: one wave per candidate, four candidates' row gathers in flight per wave (the rows were
    // streamed past the caches by the sweep: every gather is a full HBM round trip, and a wave that walked its ~7
    // candidates one after the other paid seven of them in a row)
    const int epp = rescore_epp(row_bits), pieces = (dim + epp - 1) / epp;
    constexpr int U = 4;
    for (uint32_t c0 = (uint32_t)wave * U; c0 < nb; c0 += NW * U) {
        uint32_t rowv[U];
        float dot[U], nrm[U];
        uint32_t nz[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            rowv[u] = (uint32_t)band[min(c0 + u, nb - 1)];  // (past the band's end: the last candidate again, not written)
            dot[u] = nrm[u] = 0.f;
            nz[u] = 0;
        }
        // ALL the loads of a trip first (4 candidates x up to 4 pieces per lane: a 768-dim float32 row is 3), then the
        // arithmetic: the form that used each piece as it came paid one HBM round trip per 64 pieces of a row --
        // 12.5 us per trip, 25-50 us of the kernel's 33-60
        constexpr int PI = 4;
        for (int i0 = lane; i0 < pieces; i0 += 64 * PI) {
            uint4 w[U][PI];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int pi = 0; pi < PI; pi++)
                    w[u][pi] = *reinterpret_cast<const uint4 *>(rows + piece_offset(lay, rowv[u], (uint32_t)min(i0 + 64 * pi, pieces - 1)));
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int pi = 0; pi < PI; pi++)
                    if (i0 + 64 * pi < pieces)
                        rescore_use<MODE == 1>(w[u][pi], qf, i0 + 64 * pi, row_bits, dim, dot[u], nrm[u], nz[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                dot[u] += __shfl_xor(dot[u], o);
                if (MODE == 1) {
                    nrm[u] += __shfl_xor(nrm[u], o);
                    nz[u] |= __shfl_xor(nz[u], o);
                }
            }
            float key;
            if (MODE == 1) {  // exactly cand_rescore_kernel's key (the certification bound is the float32 sweeps')
                key = -dot[u] * __frsqrt_rn(nrm[u]);
                if (nrm[u] == 0.f) key = nz[u] ? -2.0f : 1.0f;
                if (!(nrm[u] <= 3.0e38f)) key = -2.0f;  // norm overflow: forced in (see RowAcc::finish)
            } else {
                key = dot[u];
            }
            if (!(key == key)) key = 3.0e38f;
            if (key > 3.0e38f) key = 3.0e38f;
            if (lane == 0 && c0 + u < nb) band[c0 + u] = ((uint64_t)ordered_key(key) << 32) | rowv[u];
        }
    }
    __syncthreads();
    // the kp best of the band by float32 key: rank by counting (entries are unique: the row is part of the word)
    for (int i = tid; i < kp; i += kRefineThreads) out[i] = kInvalidCand;
    __syncthreads();
    for (uint32_t i = tid; i < nb; i += kRefineThreads) {
        const uint64_t mine = band[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nb; j++) rank += band[j] < mine ? 1u : 0u;
        if (rank < (uint32_t)kp) out[rank] = mine;
    }
    if (tid == 0) band_edge[q] = edge;
}

#endif  // SZG_MQ_PART == 0

}  // namespace

#if SZG_MQ_PART == 0
hipError_t launch_mq_thr(const uint64_t *lists, int kp, int n_queries, float *thr, hipStream_t stream)
{
    hipLaunchKernelGGL(mq_thr_kernel, dim3(1), dim3(64), 0, stream, lists, kp, n_queries, thr);
    return hipGetLastError();
}

hipError_t launch_cand_select(const uint64_t *cand_buf, const uint32_t *cand_count, uint32_t cand_cap,
                              int kp, int n_queries, uint64_t *lists, hipStream_t stream)
{
    const size_t lds = (size_t)4 * kp * sizeof(uint64_t);
    hipLaunchKernelGGL(cand_select_kernel, dim3(n_queries), dim3(256), lds, stream, cand_buf, cand_count,
                       cand_cap, kp, lists);
    return hipGetLastError();
}

hipError_t launch_cand_rescore(int metric, const uint8_t *rows, RowLayout lay, int dim, const double *q64,
                               const double *qscale, uint64_t *cand_buf, const uint32_t *cand_count,
                               uint32_t cand_cap, int n_queries, int row_bits, hipStream_t stream)
{
    if (row_bits != 32 && row_bits != 16 && row_bits != 64 && row_bits != 8) return hipErrorInvalidValue;
    const dim3 grid(SZG_RESCORE_BLOCKS, n_queries);  // x 4 waves: one candidate per wave and trip
    const size_t lds = (size_t)((dim + 15) & ~15) * sizeof(float);
    if (metric == kCosine)
        hipLaunchKernelGGL(cand_rescore_kernel<kCosine>, grid, dim3(256), lds, stream, rows, lay, dim, q64, qscale,
                           cand_buf, cand_count, cand_cap, row_bits);
    else
        hipLaunchKernelGGL(cand_rescore_kernel<kEuclidean>, grid, dim3(256), lds, stream, rows, lay, dim, q64,
                           qscale, cand_buf, cand_count, cand_cap, row_bits);
    return hipGetLastError();
}

bool cand_refine_applies(int kp, uint32_t cand_cap, int dim, bool rescore)
{
    return kp <= kRefineMaxKp && cand_cap <= (uint32_t)kRefineMaxCands && (!rescore || dim <= 4096);
}

hipError_t launch_cand_refine(int mode, const uint8_t *rows, RowLayout lay, int dim, const double *q64,
                              const double *qscale, const double *qnorm2, const uint64_t *cand_buf, uint32_t *cand_count,
                              uint32_t cand_cap, int kp, int n_queries, const uint64_t *sent, int n_sent,
                              uint64_t *lists, float *band_edge, int row_bits, hipStream_t stream)
{
    if (!cand_refine_applies(kp, cand_cap, dim, mode > 0)) return hipErrorInvalidValue;
    if (mode > 0 && row_bits != 32 && row_bits != 16 && row_bits != 64 && row_bits != 8) return hipErrorInvalidValue;
    const size_t lds = ((size_t)kRefineMaxCands + (size_t)(kRefineThreads / 64) * kp + kp + kRefineMaxBand) * sizeof(uint64_t) +
                       (mode > 0 ? (size_t)((dim + 15) & ~15) * sizeof(float) + 4096 * sizeof(uint32_t) : 0);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(n_queries), dim3(kRefineThreads), lds, stream, rows, lay, dim, q64, qscale, qnorm2,
                           cand_buf, cand_count, cand_cap, kp, sent, n_sent, lists, band_edge, row_bits);
        return hipGetLastError();
    };
    if (mode == 0) return go(&cand_refine_kernel<0>);
    if (mode == 1) return go(&cand_refine_kernel<1>);
    return go(&cand_refine_kernel<2>);
}

#endif


#if SZG_MQ_PART == 0
size_t mq_i8_image_bytes(int row_bits, int r16, int nb)
{
    return (size_t)((r16 + 3) / 4) * kMqPlanes * (row_bits == 4 ? 2 : 1) * nb * 1024;
}
size_t mq_i8_lds_bytes(int row_bits, int r16, int nb, int groups)
{   // per group: image + constants + thresholds; + the 12 waves' hit buffers
    return (size_t)groups * (mq_i8_image_bytes(row_bits, r16, nb) + kMq8TableRows * 48 * sizeof(float)) + (size_t)SZG_MQ8_WAVES * 64 * 9;
}
size_t mq_bf16_image_bytes(int row_bits, int r16, int nb)
{   // a KiB per 32-element K-step and query block; a 128-byte step of a row holds one (32-bit rows), two (16-bit) or
    // half a one (64-bit)
    if (row_bits == 8) return (size_t)((r16 + 3) / 4) * 2 * nb * 1024;  // (tiled rows: two K-steps per 64-byte step)
    const size_t steps = (size_t)((r16 + 7) / 8);
    return (row_bits == 64 ? (steps + 1) / 2 : steps * (row_bits == 16 ? 2 : 1)) * nb * 1024;
}
size_t mq_bf16_lds_bytes(int row_bits, int r16, int nb)
{   // + thresholds, |q|^2 table and the waves' hit buffers
    // (16-bit rows -- the direct kernel -- run 12 waves without a staging KiB; the staged kernels 8 with one each)
    return mq_bf16_image_bytes(row_bits, r16, nb) + 3 * kMqMaxQueries * sizeof(float) +  // (the third table: 8-bit rows' sum g)
           std::max((size_t)SZG_MQB_WAVES * (kHitCap * 9 + 1024), (size_t)12 * kHitCap * 9);
}
hipError_t launch_mq_score_bf16_rows32(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream);
hipError_t launch_mq_score_bf16_rows16(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream);
hipError_t launch_mq_score_bf16_rows64(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream);
hipError_t launch_mq_score_bf16_rows8(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream);
hipError_t launch_mq_score_bf16(int row_bits, const MqArgs &a, int nb, int grid, hipStream_t stream)
{
    const size_t lds = mq_bf16_lds_bytes(row_bits, a.r16, nb);
    if (row_bits == 32) return launch_mq_score_bf16_rows32(a, nb, grid, lds, stream);
    if (row_bits == 16) return launch_mq_score_bf16_rows16(a, nb, grid, lds, stream);
    if (row_bits == 64) return launch_mq_score_bf16_rows64(a, nb, grid, lds, stream);
    if (row_bits == 8) return launch_mq_score_bf16_rows8(a, nb, grid, lds, stream);
    return hipErrorInvalidValue;
}
hipError_t launch_mq_score_i8_rows8(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream);
hipError_t launch_mq_score_i8_rows4(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream);
hipError_t launch_mq_score_i8(int row_bits, const MqArgs &a, int nb, int grid, hipStream_t stream)
{
    const size_t lds = mq_i8_lds_bytes(row_bits, a.r16, nb, a.n_groups > 0 ? a.n_groups : 1);
    if (row_bits == 8) return launch_mq_score_i8_rows8(a, nb, grid, lds, stream);
    if (row_bits == 4) return launch_mq_score_i8_rows4(a, nb, grid, lds, stream);
    return hipErrorInvalidValue;
}
#endif  // SZG_MQ_PART == 0

#if SZG_MQ_PART == 3 || SZG_MQ_PART == 116 || SZG_MQ_PART == 164
namespace {
constexpr int kBfRowBits = SZG_MQ_PART == 3 ? 32 : (SZG_MQ_PART == 116 ? 16 : 64);
template <int NB, int METRIC, bool COLLECT>
hipError_t launch_mq_score_bf16_t(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
    auto *kern = &mq_score_bf16s_kernel<NB, METRIC, COLLECT, kBfRowBits>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kMqbThreads), lds, stream, a);
    return hipGetLastError();
}
#if SZG_MQ_PART == 116
template <int NB, int METRIC, bool COLLECT>
hipError_t launch_mq_score_bf16d_t(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
    auto *kern = &mq_score_bf16d_kernel<NB, METRIC, COLLECT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kMqdThreads), lds, stream, a);
    return hipGetLastError();
}
#endif
template <int NB>
hipError_t launch_mq_score_bf16_m(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
#if SZG_MQ_PART == 116
    static const bool staged16 = getenv("SZG_BF16_STAGED16") != nullptr;  // (A/B: the LDS-staged form for 16-bit rows)
    if (!staged16 && a.row_norm) {  // (no resident norms -- an allocation failed, SZG_NO_ROW_NORMS: the staged form sums its own)
        if (a.collect) {
            if (a.metric == kCosine) return launch_mq_score_bf16d_t<NB, kCosine, true>(a, grid, lds, stream);
            return launch_mq_score_bf16d_t<NB, kEuclidean, true>(a, grid, lds, stream);
        }
        if (a.metric == kCosine) return launch_mq_score_bf16d_t<NB, kCosine, false>(a, grid, lds, stream);
        return launch_mq_score_bf16d_t<NB, kEuclidean, false>(a, grid, lds, stream);
    }
#endif
    if (a.collect) {
        if (a.metric == kCosine) return launch_mq_score_bf16_t<NB, kCosine, true>(a, grid, lds, stream);
        return launch_mq_score_bf16_t<NB, kEuclidean, true>(a, grid, lds, stream);
    }
    if (a.metric == kCosine) return launch_mq_score_bf16_t<NB, kCosine, false>(a, grid, lds, stream);
    return launch_mq_score_bf16_t<NB, kEuclidean, false>(a, grid, lds, stream);
}
}  // namespace

#if SZG_MQ_PART == 3
hipError_t launch_mq_score_bf16_rows32(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream)
#elif SZG_MQ_PART == 116
hipError_t launch_mq_score_bf16_rows16(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream)
#else
hipError_t launch_mq_score_bf16_rows64(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream)
#endif
{
    if (a.tiled || a.n_rows == 0 || !a.zero16) return hipErrorInvalidValue;
    switch (nb) {
    case 1: return launch_mq_score_bf16_m<1>(a, grid, lds, stream);
    case 2: return launch_mq_score_bf16_m<2>(a, grid, lds, stream);
    case 3: return launch_mq_score_bf16_m<3>(a, grid, lds, stream);
    case 4: return launch_mq_score_bf16_m<4>(a, grid, lds, stream);
    case 5: return launch_mq_score_bf16_m<5>(a, grid, lds, stream);
    case 6: return launch_mq_score_bf16_m<6>(a, grid, lds, stream);
    default: return hipErrorInvalidValue;
    }
}
#endif  // SZG_MQ_PART == 3 || 116 || 164

#if SZG_MQ_PART == 108
namespace {
template <int NB, int METRIC, bool COLLECT>
hipError_t launch_mq_score_bf16d8_t(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
    auto *kern = &mq_score_bf16d8_kernel<NB, METRIC, COLLECT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kMqd8Threads), lds, stream, a);
    return hipGetLastError();
}
template <int NB>
hipError_t launch_mq_score_bf16d8_m(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
    if (a.collect) {
        if (a.metric == kCosine) return launch_mq_score_bf16d8_t<NB, kCosine, true>(a, grid, lds, stream);
        return launch_mq_score_bf16d8_t<NB, kEuclidean, true>(a, grid, lds, stream);
    }
    if (a.metric == kCosine) return launch_mq_score_bf16d8_t<NB, kCosine, false>(a, grid, lds, stream);
    return launch_mq_score_bf16d8_t<NB, kEuclidean, false>(a, grid, lds, stream);
}
}  // namespace

hipError_t launch_mq_score_bf16_rows8(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream)
{
    if (!a.tiled || a.steps == 0 || a.n_rows == 0 || !a.row_norm) return hipErrorInvalidValue;
    switch (nb) {
    case 1: return launch_mq_score_bf16d8_m<1>(a, grid, lds, stream);
    case 2: return launch_mq_score_bf16d8_m<2>(a, grid, lds, stream);
    case 3: return launch_mq_score_bf16d8_m<3>(a, grid, lds, stream);
    case 4: return launch_mq_score_bf16d8_m<4>(a, grid, lds, stream);
    case 5: return launch_mq_score_bf16d8_m<5>(a, grid, lds, stream);
    case 6: return launch_mq_score_bf16d8_m<6>(a, grid, lds, stream);
    default: return hipErrorInvalidValue;
    }
}
#endif  // SZG_MQ_PART == 108

#if SZG_MQ_PART == 1 || SZG_MQ_PART == 2
namespace {
constexpr int kRowBits = SZG_MQ_PART == 1 ? 8 : 4;
template <int NB, int METRIC, bool COLLECT, bool FAST>
hipError_t launch_mq_score_i8_t(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(&mq_score_i8_kernel<NB, METRIC, COLLECT, FAST, kRowBits>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mq_score_i8_kernel<NB, METRIC, COLLECT, FAST, kRowBits>), dim3(grid), dim3(kMq8Threads), lds,
                       stream, a);
    return hipGetLastError();
}
template <int NB, int METRIC, int STEPS>
hipError_t launch_mq_score_i8s_t(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mq_score_i8s_kernel<NB, METRIC, kRowBits, STEPS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mq_score_i8s_kernel<NB, METRIC, kRowBits, STEPS>), dim3(grid), dim3(64 * i8s_waves<kRowBits, STEPS>()), lds, stream, a);
    return hipGetLastError();
}
template <int NB, int METRIC>
bool launch_mq_score_i8s(const MqArgs &a, int grid, size_t lds, hipStream_t stream, hipError_t *e)
{
    switch (a.r16 / 4) {  // the row shapes with a kernel of their own (768 / 384 dims, 8- and 4-bit)
    case 12: *e = launch_mq_score_i8s_t<NB, METRIC, 12>(a, grid, lds, stream); return true;
    case 6: *e = launch_mq_score_i8s_t<NB, METRIC, 6>(a, grid, lds, stream); return true;
    case 3: *e = launch_mq_score_i8s_t<NB, METRIC, 3>(a, grid, lds, stream); return true;
    default: return false;
    }
}
template <int NB>
hipError_t launch_mq_score_i8_m(const MqArgs &a, int grid, size_t lds, hipStream_t stream)
{
    if (a.collect) {
        if constexpr (NB == 3) {  // full query groups: the row shapes with a kernel of their own
            if (a.shape_kernels && a.tiled && a.r16 % 4 == 0 && a.n_rows > 0 && (!SZG_I8S_RN || a.row_norm)) {
                hipError_t e = hipSuccess;
                if (a.metric == kCosine ? launch_mq_score_i8s<NB, kCosine>(a, grid, lds, stream, &e)
                                        : launch_mq_score_i8s<NB, kEuclidean>(a, grid, lds, stream, &e))
                    return e;
            }
        }
        if (a.tiled && a.r16 % 4 == 0 && a.n_rows > 0) {  // whole 64-byte steps: the predicate-free kernel
            if (a.metric == kCosine) return launch_mq_score_i8_t<NB, kCosine, true, true>(a, grid, lds, stream);
            return launch_mq_score_i8_t<NB, kEuclidean, true, true>(a, grid, lds, stream);
        }
        if (a.metric == kCosine) return launch_mq_score_i8_t<NB, kCosine, true, false>(a, grid, lds, stream);
        return launch_mq_score_i8_t<NB, kEuclidean, true, false>(a, grid, lds, stream);
    }
    if (a.metric == kCosine) return launch_mq_score_i8_t<NB, kCosine, false, false>(a, grid, lds, stream);
    return launch_mq_score_i8_t<NB, kEuclidean, false, false>(a, grid, lds, stream);
}
}  // namespace

#if SZG_MQ_PART == 1
hipError_t launch_mq_score_i8_rows8(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream)
#else
hipError_t launch_mq_score_i8_rows4(const MqArgs &a, int nb, int grid, size_t lds, hipStream_t stream)
#endif
{
    switch (nb) {
    case 1: return launch_mq_score_i8_m<1>(a, grid, lds, stream);
    case 2: return launch_mq_score_i8_m<2>(a, grid, lds, stream);
    case 3: return launch_mq_score_i8_m<3>(a, grid, lds, stream);
    default: return hipErrorInvalidValue;
    }
}
#endif  // SZG_MQ_PART == 1 || 2

#if SZG_MQ_PART == 0
hipError_t launch_mq_select(const float *keys, size_t key_stride, uint32_t n_rows,
                            const uint64_t *live_bits, const uint64_t *allow_bits,
                            uint32_t allow_stride, int kp, int n_queries, int blocks_per_query,
                            uint64_t *block_lists, hipStream_t stream, float *thr_out, uint32_t *count_zero)
{
    if (thr_out && blocks_per_query != 1) return hipErrorInvalidValue;
    if (thr_out && kp >= 1) {  // the threshold pass: radix select (no lists)
        hipLaunchKernelGGL(mq_thr_radix_kernel, dim3(n_queries), dim3(1024), 0, stream, keys, key_stride, n_rows,
                           live_bits, allow_bits, allow_stride, kp, thr_out, count_zero);
        return hipGetLastError();
    }
    // the threshold pass walks a query's prefix keys with ONE block (it publishes the threshold): latency-bound, so
    // it gets 16 waves instead of 4 when their lists fit (50 -> ~15 us for a 96-query batch)
    const int threads = thr_out && (size_t)16 * kp * sizeof(uint64_t) <= 48u * 1024u ? 1024 : 256;
    const size_t lds = (size_t)(threads / 64) * kp * sizeof(uint64_t);
    hipLaunchKernelGGL(mq_select_kernel, dim3(blocks_per_query, n_queries), dim3(threads), lds, stream, keys,
                       key_stride, n_rows, live_bits, allow_bits, allow_stride, kp, block_lists, thr_out,
                       count_zero);
    return hipGetLastError();
}

#endif

}  // namespace szg
