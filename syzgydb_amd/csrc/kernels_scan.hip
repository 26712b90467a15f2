// kernels_scan.hip -- the fused brute-force scan for gfx950 (MI355X).
//
// One launch sweeps the resident corpus once and replaces the reference's HOT
// LOOP (collection.go:672-684): per row it dequantizes (collection.go:768-794,
// quantization.go:25-36), accumulates the distance terms of
// euclideanDistance / angularDistance (collection.go:812-832) and feeds a
// running selection (collection.go:598-619).
//
// The kernel is HBM-bound by construction: every lane issues 16-byte loads of
// the packed rows, lanes of one row group read contiguous bytes, the query
// lives in LDS (piece-swizzled so ds_read_b128 is conflict-free), per-row sums
// are reduced with wave shuffles and candidates go to per-wave sorted lists in
// LDS that are merged block-wide at the end.  Ranking keys are monotone
// surrogates of the reference distance (-cos or squared L2); the exact
// float64 distance is recomputed for the few survivors by kernels_exact.hip.
#include "kernels.h"

namespace szg {

namespace {

constexpr int kWave = 64;
#ifndef SZG_RING
#define SZG_RING 8
#endif
constexpr int kRing = SZG_RING;  // 16-byte loads each lane keeps in flight

template <int QBITS>
struct Traits {
    static constexpr int E = 128 / QBITS;  // elements per 16-byte piece
    using acc_t = float;
    static constexpr int QB = 4;  // bytes per query element in LDS
};
template <>
struct Traits<64> {
    static constexpr int E = 2;
    using acc_t = double;
    static constexpr int QB = 8;
};

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// streaming 16-byte load of one piece of a packed row (read once per scan)
__device__ __forceinline__ u32x4 load_piece(const uint8_t *p)
{
#ifdef SZG_PLAIN_LOADS
    return *reinterpret_cast<const u32x4 *>(p);
#else
    // non-temporal: the corpus is far larger than L2 + Infinity Cache and each byte
    // is used once per scan; measured 6.18 vs 5.66 TB/s on the 1M x 768 fp32 scan
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
#endif
}

// value of the lane a DPP control pairs this lane with (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror: after the four steps every lane of an aligned
// 2/4/8/16-lane group holds the group's sum)
template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double v)
{
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return ((uint64_t)hi << 32) | lo;
}

// ---- per-piece accumulation -------------------------------------------------
// a0: dot (cosine) or sum of squared differences (euclid); a1: sum of squares of
// the row (cosine only); nz: OR of the row's magnitude bits (cosine, float rows).

template <int METRIC>
__device__ __forceinline__ void acc4(const float4 q, float x0, float x1, float x2, float x3,
                                     float &a0, float &a1)
{
    if (METRIC == kCosine) {
        a0 = fmaf(q.x, x0, a0);
        a0 = fmaf(q.y, x1, a0);
        a0 = fmaf(q.z, x2, a0);
        a0 = fmaf(q.w, x3, a0);
        a1 = fmaf(x0, x0, a1);
        a1 = fmaf(x1, x1, a1);
        a1 = fmaf(x2, x2, a1);
        a1 = fmaf(x3, x3, a1);
    } else {
        float d0 = q.x - x0, d1 = q.y - x1, d2 = q.z - x2, d3 = q.w - x3;
        a0 = fmaf(d0, d0, a0);
        a0 = fmaf(d1, d1, a0);
        a0 = fmaf(d2, d2, a0);
        a0 = fmaf(d3, d3, a0);
    }
}

template <int QBITS, int METRIC>
struct Piece;

// 32-bit: resident elements are little-endian IEEE floats.
template <int METRIC>
struct Piece<32, METRIC> {
    __device__ static __forceinline__ void run(const uint4 raw, const uint8_t *q, int j, int r16,
                                               int dim, float &a0, float &a1, uint32_t &nz)
    {
        const float4 qv = reinterpret_cast<const float4 *>(q)[j];
        acc4<METRIC>(qv, __uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z),
                     __uint_as_float(raw.w), a0, a1);
        if (METRIC == kCosine) nz |= (raw.x | raw.y | raw.z | raw.w) & 0x7FFFFFFFu;
    }
};

// 64-bit: native float64 arithmetic (FP64 VALU is ample at HBM rate).
template <int METRIC>
struct Piece<64, METRIC> {
    __device__ static __forceinline__ void run(const uint4 raw, const uint8_t *q, int j, int r16,
                                               int dim, double &a0, double &a1, uint32_t &nz)
    {
        const double2 qv = reinterpret_cast<const double2 *>(q)[j];
        const double x0 = __longlong_as_double(((long long)raw.y << 32) | (long long)raw.x);
        const double x1 = __longlong_as_double(((long long)raw.w << 32) | (long long)raw.z);
        if (METRIC == kCosine) {
            a0 = fma(qv.x, x0, a0);
            a0 = fma(qv.y, x1, a0);
            a1 = fma(x0, x0, a1);
            a1 = fma(x1, x1, a1);
            nz |= (raw.x | raw.z) | ((raw.y | raw.w) & 0x7FFFFFFFu);
        } else {
            const double d0 = qv.x - x0, d1 = qv.y - x1;
            a0 = fma(d0, d0, a0);
            a0 = fma(d1, d1, a0);
        }
    }
};

// Quantized kinds decode to the odd integer n = 2v - maxInt (exact in float);
// dequantize(v) = n / maxInt (quantization.go:34-35 up to rounding), and the
// common 1/maxInt cancels in -cos and is folded into the query for euclid.
template <int QBITS>
__device__ __forceinline__ float qn(uint32_t v)
{
    constexpr float M = (float)((1u << QBITS) - 1u);
    return fmaf((float)v, 2.0f, -M);
}

template <int QBITS, int METRIC, int E>
__device__ __forceinline__ void acc_quant(float (&n)[E], const uint8_t *q, int j, int r16, int dim,
                                          float &a0, float &a1)
{
    const int e0 = j * E;
    if (e0 + E > dim) {  // tail piece of the row: padding decodes to -maxInt, mask it
#pragma unroll
        for (int i = 0; i < E; i++)
            if (e0 + i >= dim) n[i] = 0.0f;
    }
    const float4 *q4 = reinterpret_cast<const float4 *>(q);
#pragma unroll
    for (int c = 0; c < E / 4; c++) {
        const float4 qv = q4[c * r16 + j];
        acc4<METRIC>(qv, n[4 * c], n[4 * c + 1], n[4 * c + 2], n[4 * c + 3], a0, a1);
    }
}

template <int METRIC>
struct Piece<16, METRIC> {
    __device__ static __forceinline__ void run(const uint4 raw, const uint8_t *q, int j, int r16,
                                               int dim, float &a0, float &a1, uint32_t &nz)
    {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        float n[8];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            n[2 * d] = qn<16>(w[d] & 0xFFFFu);
            n[2 * d + 1] = qn<16>(w[d] >> 16);
        }
        acc_quant<16, METRIC, 8>(n, q, j, r16, dim, a0, a1);
    }
};

template <int METRIC>
struct Piece<8, METRIC> {
    __device__ static __forceinline__ void run(const uint4 raw, const uint8_t *q, int j, int r16,
                                               int dim, float &a0, float &a1, uint32_t &nz)
    {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        float n[16];
#pragma unroll
        for (int d = 0; d < 4; d++) {
#pragma unroll
            for (int k = 0; k < 4; k++) n[4 * d + k] = qn<8>((w[d] >> (8 * k)) & 0xFFu);
        }
        acc_quant<8, METRIC, 16>(n, q, j, r16, dim, a0, a1);
    }
};

// 4-bit: byte b holds element 2b in its high nibble, 2b+1 in its low nibble
// (collection.go:774-779).
template <int METRIC>
struct Piece<4, METRIC> {
    __device__ static __forceinline__ void run(const uint4 raw, const uint8_t *q, int j, int r16,
                                               int dim, float &a0, float &a1, uint32_t &nz)
    {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        float n[32];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t hi = (w[d] >> 4) & 0x0F0F0F0Fu;
            const uint32_t lo = w[d] & 0x0F0F0F0Fu;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                n[8 * d + 2 * k] = qn<4>((hi >> (8 * k)) & 0xFFu);
                n[8 * d + 2 * k + 1] = qn<4>((lo >> (8 * k)) & 0xFFu);
            }
        }
        acc_quant<4, METRIC, 32>(n, q, j, r16, dim, a0, a1);
    }
};

// ---- per-wave sorted candidate list in LDS ----------------------------------
// Only the owning wave touches its list and DS operations of one wave complete
// in issue order, so no s_barrier is needed; the wave barriers stop the
// compiler from moving one lane's store across another lane's load (it only
// reasons per thread).  Plain (non-volatile) pointers keep the accesses ds_*
// instructions: a volatile generic pointer would turn them into flat_* ops,
// which drain the whole global-load queue on every use.

__device__ __forceinline__ uint64_t list_insert(uint64_t *list, int kp, uint64_t c, int lane)
{
    int pos = 0;
    for (int base = 0; base < kp; base += kWave) {
        const int e = base + lane;
        const bool lt = e < kp && list[e] < c;
        pos += __popcll(__ballot(lt));
    }
    // shift [pos, kp-2] one slot up, highest chunk first
    for (int base = ((kp - 1) / kWave) * kWave; base >= 0; base -= kWave) {
        if (base + kWave - 1 <= pos) break;
        const int e = base + lane;
        const bool mv = e > pos && e < kp;
        uint64_t v = 0;
        if (mv) v = list[e - 1];
        __builtin_amdgcn_wave_barrier();
        if (mv) list[e] = v;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) list[pos] = c;
    __builtin_amdgcn_wave_barrier();
    return list[kp - 1];
}

// number of entries of sorted list[0..n) that are < c
__device__ __forceinline__ int lower_count(const uint64_t *list, int n, uint64_t c)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid] < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- the scan ---------------------------------------------------------------

template <int QBITS, int METRIC, int D, bool COLLECT, bool MASKED>
__global__ __launch_bounds__(512) void scan_kernel(const ScanArgs a)
{
    using T = Traits<QBITS>;
    using acc_t = typename T::acc_t;
    extern __shared__ __align__(16) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r16 = a.map.r16;
    const int qbytes = r16 * T::E * T::QB;  // multiple of 16

    uint64_t *lists = reinterpret_cast<uint64_t *>(smem + qbytes);
    uint64_t *mylist = lists + (size_t)wave * a.kp;
    const int L = a.map.L, P = a.map.P, gpw = a.map.gpw;
    const int grp = lane / L;
    const int lig = lane - grp * L;
    const bool active = grp < gpw;
    const uint64_t stride = (uint64_t)gridDim.x * nwaves * gpw;
    const uint64_t row_first = ((uint64_t)blockIdx.x * nwaves + wave) * gpw;

    // Query-major batch: one launch walks the corpus once per query, back to
    // back.  Blocks move from query to query on their own (block-level barriers
    // only), so there is no chip-wide tail or launch gap between queries.
    for (int qi = 0; qi < a.n_queries; qi++) {
    if (qi) __syncthreads();  // the previous query's lists have been merged
    {   // stage the query into LDS (it is L2-resident after the first block)
        const uint4 *src = reinterpret_cast<const uint4 *>(
            reinterpret_cast<const uint8_t *>(a.query) + (size_t)qi * a.query_stride);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (int i = tid; i < qbytes / 16; i += blockDim.x) dst[i] = src[i];
    }
    if (!COLLECT)
        for (int i = lane; i < a.kp; i += kWave) mylist[i] = kInvalidCand;
    __syncthreads();
    const uint64_t *allow_bits = a.allow_bits ? a.allow_bits + (size_t)qi * a.allow_stride : nullptr;
    uint64_t worst = kInvalidCand;

    // is the row of this lane's group at wave-row `row0` to be scanned?
    auto row_valid = [&](uint64_t row0) -> bool {
        const uint64_t r = row0 + grp;
        bool v = active && r < a.n_rows;
        if (MASKED && v) {
            const uint32_t rr = (uint32_t)r;
            if (a.live_bits) v = (a.live_bits[rr >> 6] >> (rr & 63)) & 1;         // removed
            if (v && allow_bits) v = (allow_bits[rr >> 6] >> (rr & 63)) & 1;      // filter
        }
        return v;
    };

    // One row is done: reduce the group's L lanes, form the key, select.
    auto finish_row = [&](uint64_t row0, bool valid, acc_t a0, acc_t a1, uint32_t nz) {
        if (a.map.pow2) {
            // aligned power-of-two groups: DPP inside the 16-lane row, permutes across rows
            if (L >= 2) { a0 += dpp_xchg<0xB1>(a0); if (METRIC == kCosine) a1 += dpp_xchg<0xB1>(a1); }
            if (L >= 4) { a0 += dpp_xchg<0x4E>(a0); if (METRIC == kCosine) a1 += dpp_xchg<0x4E>(a1); }
            if (L >= 8) { a0 += dpp_xchg<0x141>(a0); if (METRIC == kCosine) a1 += dpp_xchg<0x141>(a1); }
            if (L >= 16) { a0 += dpp_xchg<0x140>(a0); if (METRIC == kCosine) a1 += dpp_xchg<0x140>(a1); }
            if (L >= 32) { a0 += __shfl_xor(a0, 16); if (METRIC == kCosine) a1 += __shfl_xor(a1, 16); }
            if (L >= 64) { a0 += __shfl_xor(a0, 32); if (METRIC == kCosine) a1 += __shfl_xor(a1, 32); }
        } else {
            for (int w = L; w > 1;) {
                const int half = (w + 1) >> 1;
                const acc_t o0 = __shfl_down(a0, half);
                const acc_t o1 = __shfl_down(a1, half);
                if (lig + half < w) {
                    a0 += o0;
                    a1 += o1;
                }
                w = half;
            }
        }
        float key;
        if (METRIC == kCosine) {
            // query is pre-normalised, so key = -cos.  A zero row is distance 1.0
            // (collection.go:828-830) == cos -1; an underflowed norm is forced in.
            const bool zero = a1 == (acc_t)0;
            if (sizeof(acc_t) == 4)
                key = -(float)a0 * __frsqrt_rn((float)a1);
            else
                key = (float)(-a0 / sqrt(a1));
            if (__ballot(zero && valid && lig == 0)) {  // rare: tell true zeros from underflow
                for (int w = L; w > 1;) {
                    const int half = (w + 1) >> 1;
                    const uint32_t oz = __shfl_down(nz, half);
                    if (lig + half < w) nz |= oz;
                    w = half;
                }
                if (zero) key = nz ? -2.0f : 1.0f;
            }
        } else {
            key = (float)a0;
        }
        if (!(key == key)) key = 3.0e38f;       // NaN: worst finite
        if (key > 3.0e38f) key = 3.0e38f;       // +inf (overflow): worst finite
        const uint32_t row = (uint32_t)(row0 + grp);
        const uint64_t c = ((uint64_t)ordered_key(key) << 32) | row;
        const bool leader = valid && lig == 0;

        if (COLLECT) {
            const bool hit = leader && (uint32_t)(c >> 32) <= a.thr_ukey;
            const uint64_t m = __ballot(hit);
            if (m) {
                const int first = __ffsll((long long)m) - 1;
                uint32_t base = 0;
                if (lane == first) base = atomicAdd(a.collect_count, (uint32_t)__popcll(m));
                base = __shfl(base, first);
                if (hit) {
                    const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (idx < a.collect_cap) a.collect_buf[idx] = c;
                }
            }
        } else {
            uint64_t m = __ballot(leader && c < worst);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint64_t cc = shfl_u64(c, src);
                if (cc < worst) worst = list_insert(mylist, a.kp, cc, lane);
            }
        }
    };

    // The piece ring: every lane keeps D 16-byte loads in flight, walking its
    // rows piece by piece (row-major), whatever the row size.  Slot u is consumed
    // and immediately re-issued with the piece D steps ahead, so HBM requests
    // keep flowing while a finished row is reduced and selected.  Loads are
    // unconditional (idle lanes read a dummy address) and the steady-state loop
    // is branch-free apart from the row-finish, so the compiler can emit
    // counted s_waitcnt vmcnt(D-1) instead of draining the queue.
    u32x4 ring[D];
    uint32_t okmask = 0;
    const uint64_t n_it = row_first < a.n_rows ? (a.n_rows - row_first + stride - 1) / stride : 0;
    const uint64_t NP = n_it * (uint64_t)P;  // pieces this wave walks
    // issue cursor
    uint64_t irow0 = row_first;
    int ip = 0;
    bool ivalid = row_valid(irow0);
    bool inext = MASKED ? row_valid(irow0 + stride) : false;
    const uint8_t *irp = a.rows + (uint64_t)(uint32_t)(irow0 + grp) * a.pitch;
    // consume cursor
    uint64_t crow0 = row_first;
    int cp = 0;
    bool cvalid = false;
    acc_t a0 = 0, a1 = 0;
    uint32_t nz = 0;

#define SZG_ISSUE(u)                                                                    \
    {                                                                                   \
        const int j_ = ip * L + lig;                                                    \
        const bool ok_ = ivalid && j_ < r16;                                            \
        ring[u] = load_piece(ok_ ? irp + (size_t)j_ * 16 : a.rows);                     \
        okmask = (okmask & ~(1u << (u))) | ((uint32_t)ok_ << (u));                      \
        if (++ip == P) {                                                                \
            ip = 0;                                                                     \
            irow0 += stride;                                                            \
            irp = a.rows + (uint64_t)(uint32_t)(irow0 + grp) * a.pitch;                 \
            if (MASKED) {                                                               \
                ivalid = inext;                                                         \
                inext = row_valid(irow0 + stride);                                      \
            } else {                                                                    \
                ivalid = active && irow0 + grp < a.n_rows;                              \
            }                                                                           \
        }                                                                               \
    }

#define SZG_CONSUME(u)                                                                  \
    {                                                                                   \
        const int j_ = cp * L + lig;                                                    \
        const bool ok_ = (okmask >> (u)) & 1u;                                          \
        if (cp == 0) cvalid = ok_; /* piece 0 of the group's first lane is in range */  \
        if (ok_) {                                                                      \
            const u32x4 v_ = ring[u];                                                   \
            Piece<QBITS, METRIC>::run(make_uint4(v_.x, v_.y, v_.z, v_.w), smem, j_, r16, \
                                      a.dim, a0, a1, nz);                               \
        }                                                                               \
        if (++cp == P) {                                                                \
            finish_row(crow0, cvalid, a0, a1, nz);                                      \
            a0 = 0;                                                                     \
            a1 = 0;                                                                     \
            nz = 0;                                                                     \
            cp = 0;                                                                     \
            crow0 += stride;                                                            \
        }                                                                               \
    }

    uint64_t issued = 0, consumed = 0;
#pragma unroll
    for (int u = 0; u < D; u++) {
        if (issued < NP) {
            SZG_ISSUE(u)
            issued++;
        }
    }
    // steady state: every slot consumed is re-issued
    while (consumed + 2 * D <= NP) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            SZG_CONSUME(u)
            SZG_ISSUE(u)
        }
        consumed += D;
        issued += D;
    }
    // drain
    while (consumed < NP) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            if (consumed < NP) {
                SZG_CONSUME(u)
                consumed++;
                if (issued < NP) {
                    SZG_ISSUE(u)
                    issued++;
                }
            }
        }
    }
#undef SZG_ISSUE
#undef SZG_CONSUME

    if (COLLECT) continue;

    // block-wide k-select: rank-merge the waves' sorted lists (entries are unique)
    __syncthreads();
    uint64_t *out = a.block_lists + ((size_t)qi * gridDim.x + blockIdx.x) * a.kp;
    for (int i = tid; i < a.kp; i += blockDim.x) out[i] = kInvalidCand;
    __syncthreads();
    const int total = nwaves * a.kp;
    for (int it = tid; it < total; it += blockDim.x) {
        const int w = it / a.kp;
        const int i = it - w * a.kp;
        const uint64_t c = lists[it];
        if (c == kInvalidCand) continue;
        int rank = i;
        for (int w2 = 0; w2 < nwaves; w2++) {
            if (w2 == w) continue;
            rank += lower_count(lists + (size_t)w2 * a.kp, a.kp, c);
            if (rank >= a.kp) break;
        }
        if (rank < a.kp) out[rank] = c;
    }
    }  // for qi
}

// ---- merge of sorted candidate lists ----------------------------------------

// minimum of a 64-bit value over the wave, returned in every lane: four DPP
// steps reduce each row of 16 lanes, four readlanes finish across rows.
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#define SZG_DPP_MIN(ctrl)                                                                  \
    {                                                                                      \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(                        \
            (int)(uint32_t)v, (int)(uint32_t)v, ctrl, 0xF, 0xF, false);                    \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(                        \
            (int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), ctrl, 0xF, 0xF, false);    \
        const uint64_t o_ = ((uint64_t)hi_ << 32) | lo_;                                   \
        v = o_ < v ? o_ : v;                                                               \
    }
    SZG_DPP_MIN(0xB1)   // quad_perm [1,0,3,2]
    SZG_DPP_MIN(0x4E)   // quad_perm [2,3,0,1]
    SZG_DPP_MIN(0x141)  // row_half_mirror
    SZG_DPP_MIN(0x140)  // row_mirror
#undef SZG_DPP_MIN
    uint64_t m = kInvalidCand;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, r * 16);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), r * 16);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        m = o < m ? o : m;
    }
    return m;
}

// Tournament merge: one wave per group of up to 64 sorted lists, one list per
// lane; kp rounds of "take the smallest head".  Lists are staged in LDS with
// coalesced loads first, so each round is a DPP reduction plus one LDS read.
__global__ __launch_bounds__(256) void merge_heads_kernel(const uint64_t *in, int n_lists, int kp,
                                                          uint64_t *out)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);
    const int lane = threadIdx.x & 63;
    const int first = blockIdx.x * kWave;
    const int mine = min(kWave, n_lists - first);
    const int total = mine * kp;
    in += (size_t)blockIdx.y * n_lists * kp;     // blockIdx.y = query of the batch
    out += (size_t)blockIdx.y * gridDim.x * kp;
    for (int i = threadIdx.x; i < total; i += blockDim.x) lists[i] = in[(size_t)first * kp + i];
    __syncthreads();
    if (threadIdx.x >= kWave) return;  // the other waves only helped staging
    int pos = 0;
    uint64_t head = lane < mine ? lists[(size_t)lane * kp] : kInvalidCand;
    uint64_t *o = out + (size_t)blockIdx.x * kp;
    uint64_t keep = kInvalidCand;  // lane i keeps output i (+64 per round of 64)
    for (int r = 0; r < kp; r++) {
        const uint64_t m = wave_min_u64(head);
        if (head == m && m != kInvalidCand) {  // entries are unique: exactly one lane advances
            pos++;
            head = pos < kp ? lists[(size_t)lane * kp + pos] : kInvalidCand;
        }
        if ((r & 63) == lane) keep = m;
        if ((r & 63) == 63 || r == kp - 1) {
            const int idx = (r & ~63) + lane;
            if (idx <= r) o[idx] = keep;
            keep = kInvalidCand;
        }
    }
}

// Rank merge (any kp): every entry finds its rank by binary searches in the
// other lists.  More work, fully parallel; used when kp is large.
__global__ __launch_bounds__(1024) void merge_kernel(const uint64_t *in, int n_lists, int kp,
                                                     int fan, uint64_t *out)
{
    extern __shared__ __align__(16) uint8_t smem[];
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);
    const int first = blockIdx.x * fan;
    const int mine = min(fan, n_lists - first);
    const int total = mine * kp;
    in += (size_t)blockIdx.y * n_lists * kp;     // blockIdx.y = query of the batch
    out += (size_t)blockIdx.y * gridDim.x * kp;
    for (int i = threadIdx.x; i < total; i += blockDim.x) lists[i] = in[(size_t)first * kp + i];
    uint64_t *o = out + (size_t)blockIdx.x * kp;
    for (int i = threadIdx.x; i < kp; i += blockDim.x) o[i] = kInvalidCand;
    __syncthreads();
    for (int it = threadIdx.x; it < total; it += blockDim.x) {
        const int w = it / kp;
        const int i = it - w * kp;
        const uint64_t c = lists[it];
        if (c == kInvalidCand) continue;
        int rank = i;
        for (int w2 = 0; w2 < mine; w2++) {
            if (w2 == w) continue;
            rank += lower_count(lists + (size_t)w2 * kp, kp, c);
            if (rank >= kp) break;
        }
        if (rank < kp) o[rank] = c;
    }
}

template <int QBITS, int METRIC>
hipError_t launch_scan_qm(const ScanArgs &a, int grid, int block, size_t lds, hipStream_t stream)
{
    const bool masked = a.live_bits != nullptr || a.allow_bits != nullptr;
    const dim3 g(grid), b(block);
    if (a.collect) {
        if (masked)
            hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, kRing, true, true>), g, b, lds, stream, a);
        else
            hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, kRing, true, false>), g, b, lds, stream, a);
    } else {
        if (masked)
            hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, kRing, false, true>), g, b, lds, stream, a);
        else
            hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, kRing, false, false>), g, b, lds, stream, a);
    }
    return hipGetLastError();
}

template <int QBITS>
hipError_t launch_scan_q(int metric, const ScanArgs &a, int grid, int block, size_t lds,
                         hipStream_t stream)
{
    if (metric == kCosine) return launch_scan_qm<QBITS, kCosine>(a, grid, block, lds, stream);
    return launch_scan_qm<QBITS, kEuclidean>(a, grid, block, lds, stream);
}

}  // namespace

size_t scan_lds_bytes(int qbits, const RowMap &m, int kp, int block)
{
    const size_t e = 128 / qbits;
    const size_t qb = qbits == 64 ? 8 : 4;
    return (size_t)m.r16 * e * qb + (size_t)(block / kWave) * kp * sizeof(uint64_t);
}

hipError_t launch_scan(int qbits, int metric, const ScanArgs &a, int grid, int block,
                       hipStream_t stream)
{
    const size_t lds = scan_lds_bytes(qbits, a.map, a.collect ? 0 : a.kp, block);
    switch (qbits) {
    case 4: return launch_scan_q<4>(metric, a, grid, block, lds, stream);
    case 8: return launch_scan_q<8>(metric, a, grid, block, lds, stream);
    case 16: return launch_scan_q<16>(metric, a, grid, block, lds, stream);
    case 32: return launch_scan_q<32>(metric, a, grid, block, lds, stream);
    case 64: return launch_scan_q<64>(metric, a, grid, block, lds, stream);
    default: return hipErrorInvalidValue;
    }
}

int merge_fan(int kp)
{
    if (kp <= 128) return kWave;                       // tournament: one list per lane
    return kp >= 4096 ? 2 : (8192 / kp < 32 ? 8192 / kp : 32);  // rank merge, <= 64 KiB of LDS
}

hipError_t launch_merge(const uint64_t *in, int n_lists, int kp, int n_queries, uint64_t *out,
                        hipStream_t stream)
{
    const int fan = merge_fan(kp);
    const dim3 grid((n_lists + fan - 1) / fan, n_queries);
    const size_t lds = (size_t)fan * kp * sizeof(uint64_t);
    if (kp <= 128) {
        hipLaunchKernelGGL(merge_heads_kernel, grid, dim3(256), lds, stream, in, n_lists, kp, out);
        return hipGetLastError();
    }
    int block = fan * kp;
    if (block > 1024) block = 1024;
    block = (block + 63) & ~63;
    hipLaunchKernelGGL(merge_kernel, grid, dim3(block), lds, stream, in, n_lists, kp, fan, out);
    return hipGetLastError();
}

}  // namespace szg
