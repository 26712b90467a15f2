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
#include "device_lists.h"

// Built six times: -DSZG_QBITS=4/8/16/32/64 each carry the scan kernels of one
// quantization (80 instantiations in one unit took minutes to compile); the unit
// without it (0) has the list-merge kernels and the dispatcher.
#ifndef SZG_QBITS
#define SZG_QBITS 0
#endif

namespace szg {

namespace {

constexpr int kWave = 64;
#ifndef SZG_MIN_BLOCKS
#define SZG_MIN_BLOCKS 3  // 256-thread blocks per CU the register budget allows: the sweeps run 2-3 per CU
                          // (scan_geometry), and 168 registers keep every variant free of spills (at 128
                          // the deep-ring and row-unrolled kernels spilled 2-59 registers)
#endif
#ifndef SZG_QPF
#define SZG_QPF 0  // integer paths: fetch the query's digit planes of the NEXT piece from LDS before this piece's dots
#endif
[[maybe_unused]] constexpr int kStepList = 256;  // row steps a wave compacts at a time (selective masks)  // 16-byte loads each lane keeps in flight

template <int QBITS>
struct Traits {
    using acc_t = float;
};
template <>
struct Traits<64> {
    using acc_t = double;
};

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// streaming 16-byte load of one piece of a packed row (read once per scan).
// NT (non-temporal): the corpus is far larger than L2 + Infinity Cache and each
// byte is used once per scan -- measured 6.18 vs 5.66 TB/s on the 1M x 768 f32
// scan.  Only when a lane group reads whole 128-byte lines per load: with
// narrower groups (L < 8, 64-byte segments) the neighbouring piece needs the
// same line a moment later and nt evicts it first (4-bit 384-dim rows: 4.6 TB/s
// with nt, 5.5 plain).
template <bool NT>
__device__ __forceinline__ u32x4 load_piece(const uint8_t *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return *reinterpret_cast<const u32x4 *>(p);
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
__device__ __forceinline__ int dpp_xchg(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double v)
{
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// ---- row accumulators ---------------------------------------------------------
// RowAcc<QBITS, METRIC> holds one lane's partial sums for the row its group is
// walking: reset(), piece() for each 16-byte piece, finish() reduces over the
// group's L lanes and returns the ranking key (valid in the group's first lane).

struct Grp {
    int L, lig;
    bool pow2;
};

// sum over the L lanes of a row group; every lane of an aligned power-of-two
// group gets the total, otherwise the group's first lane does
template <typename T>
__device__ __forceinline__ T grp_sum(T v, const Grp &g)
{
    if (g.pow2) {
        if (g.L >= 2) v += dpp_xchg<0xB1>(v);
        if (g.L >= 4) v += dpp_xchg<0x4E>(v);
        if (g.L >= 8) v += dpp_xchg<0x141>(v);
        if (g.L >= 16) v += dpp_xchg<0x140>(v);
        if (g.L >= 32) v += __shfl_xor(v, 16);
        if (g.L >= 64) v += __shfl_xor(v, 32);
    } else {
        for (int w = g.L; w > 1;) {
            const int half = (w + 1) >> 1;
            const T o = __shfl_down(v, half);
            if (g.lig + half < w) v += o;
            w = half;
        }
    }
    return v;
}

__device__ __forceinline__ uint32_t grp_or(uint32_t v, const Grp &g)
{
    for (int w = g.L; w > 1;) {
        const int half = (w + 1) >> 1;
        const uint32_t o = __shfl_down(v, half);
        if (g.lig + half < w) v |= o;
        w = half;
    }
    return v;
}

// Float rows (16/32-bit, float32 sums) and 64-bit rows (float64 sums).
// a0: dot (cosine) or sum of squared differences (euclid); a1: sum of squares of
// the row; nz: OR of the row's magnitude bits (tells a true zero row from an
// underflowed norm).  The query is pre-normalised (cosine: key = -cos) or, for
// 16-bit rows under euclid, pre-scaled by maxInt.
template <int QBITS, int METRIC>
struct RowAcc {
    using acc_t = typename Traits<QBITS>::acc_t;
    static constexpr bool kPrefetch = false;
    acc_t a0, a1;
    uint32_t nz;
    __device__ __forceinline__ void fetch(const uint8_t *, int, int) {}
    __device__ __forceinline__ void piece_pf(const uint4, const uint8_t *, int, int) {}
    __device__ __forceinline__ void reset()
    {
        a0 = 0;
        a1 = 0;
        nz = 0;
    }
    __device__ __forceinline__ void add4(const float4 q, float x0, float x1, float x2, float x3)
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
            const float d0 = q.x - x0, d1 = q.y - x1, d2 = q.z - x2, d3 = q.w - x3;
            a0 = fmaf(d0, d0, a0);
            a0 = fmaf(d1, d1, a0);
            a0 = fmaf(d2, d2, a0);
            a0 = fmaf(d3, d3, a0);
        }
    }
    __device__ __forceinline__ void piece(const uint4 raw, const uint8_t *q, int j, const int r16, const int dim)
    {
        if (QBITS == 32) {  // resident elements are little-endian IEEE floats
            const float4 qv = reinterpret_cast<const float4 *>(q)[j];
            add4(qv, __uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z),
                 __uint_as_float(raw.w));
            if (METRIC == kCosine) nz |= (raw.x | raw.y | raw.z | raw.w) & 0x7FFFFFFFu;
        } else if (QBITS == 64) {  // native float64 arithmetic (FP64 VALU is ample at HBM rate)
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
        } else {  // 16-bit: n = 2v - 65535 (odd, exact in float); dequantize(v) = n / 65535
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            float n[8];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                n[2 * d] = fmaf((float)(w[d] & 0xFFFFu), 2.0f, -65535.0f);
                n[2 * d + 1] = fmaf((float)(w[d] >> 16), 2.0f, -65535.0f);
            }
            const int e0 = j * 8;
            if (e0 + 8 > dim) {  // tail piece: padding decodes to -maxInt, mask it
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (e0 + i >= dim) n[i] = 0.0f;
            }
            const float4 *q4 = reinterpret_cast<const float4 *>(q);
            add4(q4[j], n[0], n[1], n[2], n[3]);
            add4(q4[r16 + j], n[4], n[5], n[6], n[7]);
        }
    }
    __device__ __forceinline__ float finish(const QConst &a, const Grp &g, bool lead)
    {
        a0 = grp_sum(a0, g);
        if (METRIC != kCosine) return (float)a0;
        a1 = grp_sum(a1, g);
        // a zero row is distance 1.0 (collection.go:828-830) == cos -1; an underflowed
        // norm is forced in (key -2) and settled by the float64 rerank -- and so is an
        // overflowed one (float32 rows with elements beyond ~1e19: finite in float64, the
        // reference ranks them; a row with an Inf / NaN element lands here too, its float64
        // distance is NaN and the host drops it)
        const bool zero = a1 == (acc_t)0;
        const bool over = sizeof(acc_t) == 4 && !((float)a1 <= 3.0e38f);
        float key;
        if (sizeof(acc_t) == 4)
            key = -(float)a0 * __frsqrt_rn((float)a1);
        else
            key = (float)(-a0 / sqrt(a1));
        if (__ballot((zero || over) && lead)) {  // rare
            nz = grp_or(nz, g);
            if (zero) key = nz ? -2.0f : 1.0f;
            if (over) key = -2.0f;
        }
        return key;
    }
};

// 8-bit rows, exact integer arithmetic.  With v' = v - 128 (one xor per dword)
// the decoded element is n = 2v - 255 = 2v' + 1.  The query is quantized on the
// host to Q_i = h*16384 + m*128 + l (balanced int8 digits, |Q| < 2^20), so
//     sum Q_i n_i = 2*(16384*H + 128*M + L) + sum Q_i,   H = sum h_i v'_i ...
// and sum n_i^2 = 4*sum v'^2 + 4*sum v' + count, all via v_dot4_i32_i8: five VALU
// ops per four elements, no conversions, and the only error left in the key is
// the query's quantization (bounded on the host, key_eps).
template <int METRIC>
struct RowAcc<8, METRIC> {
    static constexpr bool kPrefetch = SZG_QPF != 0;
    int H, M, L, SQ, SV;
    uint4 ph, pm, pl;  // digit planes of the piece about to be multiplied (dense phase, fetched one piece ahead)
    __device__ __forceinline__ void reset() { H = M = L = SQ = SV = 0; }
    __device__ __forceinline__ void fetch(const uint8_t *q, int j, int r16)
    {
        const uint4 *q4 = reinterpret_cast<const uint4 *>(q);
        ph = q4[j];
        pm = q4[r16 + j];
        pl = q4[2 * r16 + j];
    }
    __device__ __forceinline__ void piece_pf(const uint4 raw, const uint8_t *q, int jnext, const int r16)
    {
        const uint4 qh = ph, qm = pm, ql = pl;
        fetch(q, jnext, r16);
        mul(raw, qh, qm, ql);
    }
    __device__ __forceinline__ void mul(const uint4 raw, const uint4 qh, const uint4 qm, const uint4 ql)
    {
        const uint32_t w[4] = {raw.x ^ 0x80808080u, raw.y ^ 0x80808080u, raw.z ^ 0x80808080u,
                               raw.w ^ 0x80808080u};
        const uint32_t h[4] = {qh.x, qh.y, qh.z, qh.w};
        const uint32_t m[4] = {qm.x, qm.y, qm.z, qm.w};
        const uint32_t l[4] = {ql.x, ql.y, ql.z, ql.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            H = __builtin_amdgcn_sdot4((int)h[d], (int)w[d], H, false);
            M = __builtin_amdgcn_sdot4((int)m[d], (int)w[d], M, false);
            L = __builtin_amdgcn_sdot4((int)l[d], (int)w[d], L, false);
            SQ = __builtin_amdgcn_sdot4((int)w[d], (int)w[d], SQ, false);
            SV = __builtin_amdgcn_sdot4((int)w[d], 0x01010101, SV, false);
        }
    }
    __device__ __forceinline__ void piece(const uint4 raw, const uint8_t *q, int j, const int r16, const int dim)
    {
        const uint4 *q4 = reinterpret_cast<const uint4 *>(q);
        mul(raw, q4[j], q4[r16 + j], q4[2 * r16 + j]);
    }
    __device__ __forceinline__ float finish(const QConst &a, const Grp &g, bool lead)
    {
        // per-lane sums are exact integers below 2^24; combining the planes in float32
        // costs a few roundings (bounded in key_eps) and saves the float64 reduction
        float dot = fmaf(16384.0f, (float)H, fmaf(128.0f, (float)M, (float)L));
        int nrm = 4 * (SQ + SV);
        dot = grp_sum(dot, g);
        nrm = grp_sum(nrm, g);
        const float norm = (float)nrm + a.norm_bias;
        const float d2 = fmaf(2.0f, dot, a.qconst);  // sum Q n
        if (METRIC == kCosine) return -(d2 * a.qscale) * __frsqrt_rn(norm);
        return fmaf(-2.0f * a.qscale, d2, a.qnorm2 + norm);
    }
};

// 4-bit rows: the same with v'' = v - 8 (xor 0x88888888), n = 2v - 15 = 2v'' + 1,
// the query in five balanced int4 digit planes (|Q| < 2^19) and v_dot8_i32_i4:
// seven VALU ops per eight elements, no nibble unpacking.
template <int METRIC>
struct RowAcc<4, METRIC> {
    static constexpr bool kPrefetch = SZG_QPF != 0;
    int D[kPlanes4], SQ, SV;
    uint4 pp[kPlanes4];  // digit planes fetched one piece ahead (dense phase, SZG_QPF)
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int x = 0; x < kPlanes4; x++) D[x] = 0;
        SQ = SV = 0;
    }
    __device__ __forceinline__ void fetch(const uint8_t *q, int j, int r16)
    {
        const uint4 *q4 = reinterpret_cast<const uint4 *>(q);
#pragma unroll
        for (int x = 0; x < kPlanes4; x++) pp[x] = q4[x * r16 + j];
    }
    __device__ __forceinline__ void piece_pf(const uint4 raw, const uint8_t *q, int jnext, const int r16)
    {
        uint4 cur[kPlanes4];
#pragma unroll
        for (int x = 0; x < kPlanes4; x++) cur[x] = pp[x];
        fetch(q, jnext, r16);
        mul(raw, cur);
    }
    __device__ __forceinline__ void mul(const uint4 raw, const uint4 (&p)[kPlanes4])
    {
        const uint32_t w[4] = {raw.x ^ 0x88888888u, raw.y ^ 0x88888888u, raw.z ^ 0x88888888u,
                               raw.w ^ 0x88888888u};
#pragma unroll
        for (int d = 0; d < 4; d++) {
#pragma unroll
            for (int x = 0; x < kPlanes4; x++) {
                const uint32_t qd = d == 0 ? p[x].x : d == 1 ? p[x].y : d == 2 ? p[x].z : p[x].w;
                D[x] = __builtin_amdgcn_sdot8((int)qd, (int)w[d], D[x], false);
            }
            SQ = __builtin_amdgcn_sdot8((int)w[d], (int)w[d], SQ, false);
            SV = __builtin_amdgcn_sdot8((int)w[d], 0x11111111, SV, false);
        }
    }
    __device__ __forceinline__ void piece(const uint4 raw, const uint8_t *q, int j, const int r16, const int dim)
    {
        const uint4 *q4 = reinterpret_cast<const uint4 *>(q);
        uint4 p[kPlanes4];
#pragma unroll
        for (int x = 0; x < kPlanes4; x++) p[x] = q4[x * r16 + j];
        mul(raw, p);
    }
    // INT_OK: a lane's share of the row is short enough (<= 12 pieces) for the weighted plane sum
    // to stay inside int32 (|digit| <= 8, |v''| <= 8, 32 elements per piece): combine the planes
    // exactly with shift-adds and convert once.
    template <bool INT_OK = false>
    __device__ __forceinline__ float finish(const QConst &a, const Grp &g, bool lead)
    {
        float dot;
        if (INT_OK) {
            int t = D[kPlanes4 - 1];
#pragma unroll
            for (int x = kPlanes4 - 2; x >= 0; x--) t = (t << 4) + D[x];
            dot = (float)t;
        } else {
            dot = (float)D[kPlanes4 - 1];
#pragma unroll
            for (int x = kPlanes4 - 2; x >= 0; x--) dot = fmaf(16.0f, dot, (float)D[x]);
        }
        int nrm = 4 * (SQ + SV);
        dot = grp_sum(dot, g);
        nrm = grp_sum(nrm, g);
        const float norm = (float)nrm + a.norm_bias;
        const float d2 = fmaf(2.0f, dot, a.qconst);
        if (METRIC == kCosine) return -(d2 * a.qscale) * __frsqrt_rn(norm);
        return fmaf(-2.0f * a.qscale, d2, a.qnorm2 + norm);
    }
};

// ---- the scan ---------------------------------------------------------------

// LL, PP > 0: the row shape (lanes per row, pieces per lane) is a compile-time constant
// -- the common dims (384, 768, ...) get kernels whose addressing, reductions and query
// offsets are folded; 0, 0 = any shape, from a.map.
template <int QBITS, int METRIC, int D, bool COLLECT, bool MASKED, bool NT, int LL = 0, int PP = 0>
__global__ __launch_bounds__(256, SZG_MIN_BLOCKS) void scan_kernel(const ScanArgs a)
{
    static_assert((LL == 0) == (PP == 0), "shape is fixed as a whole or not at all");
    extern __shared__ __align__(16) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r16 = LL ? LL * PP : a.map.r16;
    const int qbytes = (int)query_lds_bytes(QBITS, r16);  // multiple of 16

    uint64_t *lists = reinterpret_cast<uint64_t *>(smem + qbytes);
    uint64_t *mylist = lists + (size_t)wave * a.kp;
    [[maybe_unused]] uint32_t *steplist =
        reinterpret_cast<uint32_t *>(smem + qbytes + (size_t)nwaves * (COLLECT ? 0 : a.kp) * sizeof(uint64_t)) +
        (size_t)wave * kStepList;
    const int L = LL ? LL : a.map.L, P = PP ? PP : a.map.P, gpw = LL ? kWave / LL : a.map.gpw;
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
    WaveList wl;
    wl.init(mylist, COLLECT ? 0 : a.kp, lane);
    __syncthreads();
    const uint64_t *allow_bits = a.allow_bits ? a.allow_bits + (size_t)qi * a.allow_stride : nullptr;
    const QConst qc{a.qscale[qi], a.qconst[qi], a.qnorm2[qi], (float)a.norm_bias};

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

    // Dense phase under masks (a.mask_dense: most rows pass, so every row is read and the masks
    // only decide at the row finish).  The wave's rows [row0, row0 + gpw) sit in one 64-bit
    // word of each bitmap; row0 is wave-uniform, so the words come through scalar loads and
    // never enter the vector-memory queue the ring is counting.
    auto dense_valid = [&](uint64_t row0) -> bool {
        if (!MASKED) return true;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)row0);
        uint64_t w = ~0ull;
        if (a.live_bits) w &= a.live_bits[r0 >> 6];
        if (allow_bits) w &= allow_bits[r0 >> 6];
        return (w >> ((r0 & 63u) + (uint32_t)grp)) & 1ull;
    };

    // One row is done: reduce the group's L lanes, form the key, select.
    const Grp grp_info{L, lig, LL ? true : a.map.pow2 != 0};
    // The common case after warm-up -- no row of the step can enter the list -- costs one float
    // compare and one wave-uniform branch; clamping, the ordered 64-bit candidate and the exact
    // comparison happen only behind it.
    [[maybe_unused]] const uint32_t thr_ukey = COLLECT ? a.thr_ukeys[qi] : 0u;  // per sweep: radius / escalation threshold
    [[maybe_unused]] const float thr_key = COLLECT ? key_from_ordered(thr_ukey) : 0.0f;
    [[maybe_unused]] uint64_t *const cbuf = COLLECT ? a.collect_buf + (size_t)qi * a.collect_cap : nullptr;
    [[maybe_unused]] uint32_t *const ccount = COLLECT ? a.collect_count + (size_t)qi * kCandCountStride : nullptr;
    auto finish_row = [&](uint64_t row0, bool valid, RowAcc<QBITS, METRIC> &acc) {
        float key;
        if constexpr (QBITS == 4 && LL != 0 && PP <= 12)
            key = acc.template finish<true>(qc, grp_info, valid && lig == 0);
        else
            key = acc.finish(qc, grp_info, valid && lig == 0);
        const bool leader = valid && lig == 0;
        key = fminf(key, 3.0e38f);              // +inf (overflow) and NaN (minNum returns the number): worst finite
        const bool maybe = leader && !(key > (COLLECT ? thr_key : wl.worst_key));
        if (!__ballot(maybe)) return;
        const uint32_t row = (uint32_t)(row0 + grp);
        const uint64_t c = ((uint64_t)ordered_key(key) << 32) | row;

        if (COLLECT) {
            const bool hit = maybe && (uint32_t)(c >> 32) <= thr_ukey;
            const uint64_t m = __ballot(hit);
            if (m) {
                const int first = __ffsll((long long)m) - 1;
                uint32_t base = 0;
                if (lane == first) base = atomicAdd(ccount, (uint32_t)__popcll(m));
                base = __shfl(base, first);
                if (hit) {
                    const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (idx < a.collect_cap) cbuf[idx] = c;
                }
            }
        } else {
            wl.offer(maybe, c, lane);
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
    const uint64_t n_it = row_first < a.n_rows ? (a.n_rows - row_first + stride - 1) / stride : 0;
    RowAcc<QBITS, METRIC> acc;
    acc.reset();

// prologue (fill the ring), steady state (every slot consumed is re-issued), drain
#define SZG_RUN_RING(NPX, ISSUE, CONSUME)                                               \
    {                                                                                   \
        const uint64_t np_ = (NPX);                                                     \
        uint64_t issued_ = 0, consumed_ = 0;                                            \
        _Pragma("unroll") for (int u = 0; u < D; u++)                                   \
        {                                                                               \
            if (issued_ < np_) {                                                        \
                ISSUE(u)                                                                \
                issued_++;                                                              \
            }                                                                           \
        }                                                                               \
        while (consumed_ + 2 * D <= np_) {                                              \
            _Pragma("unroll") for (int u = 0; u < D; u++)                               \
            {                                                                           \
                CONSUME(u)                                                              \
                ISSUE(u)                                                                \
            }                                                                           \
            consumed_ += D;                                                             \
            issued_ += D;                                                               \
        }                                                                               \
        while (consumed_ < np_) {                                                       \
            _Pragma("unroll") for (int u = 0; u < D; u++)                               \
            {                                                                           \
                if (consumed_ < np_) {                                                  \
                    CONSUME(u)                                                          \
                    consumed_++;                                                        \
                    if (issued_ < np_) {                                                \
                        ISSUE(u)                                                        \
                        issued_++;                                                      \
                    }                                                                   \
                }                                                                       \
            }                                                                           \
        }                                                                               \
    }

// The dense phase's ring: at least 2*D pieces, so the prologue is unconditional and the
// compiler can see that slot u's load is always the oldest of D in flight (counted
// s_waitcnt vmcnt(D-1) instead of draining the queue; the sched barriers keep the issue
// order it is counting on).
#define SZG_RUN_RING_DENSE(NPX, ISSUE, CONSUME)                                         \
    {                                                                                   \
        const uint64_t np_ = (NPX);                                                     \
        uint64_t issued_ = D, consumed_ = 0;                                            \
        _Pragma("unroll") for (int u = 0; u < D; u++)                                   \
        {                                                                               \
            ISSUE(u)                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                          \
        }                                                                               \
        while (consumed_ + 2 * D <= np_) {                                              \
            _Pragma("unroll") for (int u = 0; u < D; u++)                               \
            {                                                                           \
                CONSUME(u)                                                              \
                ISSUE(u)                                                                \
                __builtin_amdgcn_sched_barrier(0);                                      \
            }                                                                           \
            consumed_ += D;                                                             \
            issued_ += D;                                                               \
        }                                                                               \
        while (consumed_ < np_) {                                                       \
            _Pragma("unroll") for (int u = 0; u < D; u++)                               \
            {                                                                           \
                if (consumed_ < np_) {                                                  \
                    CONSUME(u)                                                          \
                    consumed_++;                                                        \
                    if (issued_ < np_) {                                                \
                        ISSUE(u)                                                        \
                        issued_++;                                                      \
                    }                                                                   \
                }                                                                       \
            }                                                                           \
        }                                                                               \
    }

// Row-shape kernels whose ring depth divides the pieces per lane: the steady state is unrolled
// over one whole ROW (P pieces, slot = piece % D), so every slot sits at a fixed piece of the
// row -- the row-finish test, the row jump of the issue cursor and the query offsets are
// compile-time facts and no per-piece branch is left.  The rows left over (< D + P pieces
// from the end) drain through the guarded tail loop.
#define SZG_RUN_RING_ROWS(NPX, ISSUE, CONSUME)                                          \
    {                                                                                   \
        const uint64_t np_ = (NPX);                                                     \
        uint64_t issued_ = D, consumed_ = 0;                                            \
        _Pragma("unroll") for (int u = 0; u < D; u++)                                   \
        {                                                                               \
            ISSUE(u)                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                          \
        }                                                                               \
        while (consumed_ + D + PP <= np_) {                                             \
            _Pragma("unroll") for (int p_ = 0; p_ < PP; p_++)                           \
            {                                                                           \
                CONSUME(p_ % D)                                                         \
                ISSUE(p_ % D)                                                           \
                __builtin_amdgcn_sched_barrier(0);                                      \
            }                                                                           \
            consumed_ += PP;                                                            \
            issued_ += PP;                                                              \
        }                                                                               \
        while (consumed_ < np_) {                                                       \
            _Pragma("unroll") for (int u = 0; u < D; u++)                               \
            {                                                                           \
                if (consumed_ < np_) {                                                  \
                    CONSUME(u)                                                          \
                    consumed_++;                                                        \
                    if (issued_ < np_) {                                                \
                        ISSUE(u)                                                        \
                        issued_++;                                                      \
                    }                                                                   \
                }                                                                       \
            }                                                                           \
        }                                                                               \
    }

    // ---- dense phase: no masks, L*P == r16, gpw*L == 64, every group's row in
    // range.  No predicates at all: pointer-increment addressing, unconditional
    // accumulation.  Covers all but (at most) the wave's last row step.
    uint64_t it_dense = 0;
    if ((!MASKED || a.mask_dense) && (LL || a.map.dense) && row_first + (uint64_t)gpw <= a.n_rows)
        it_dense = (a.n_rows - (uint64_t)gpw - row_first) / stride + 1;
    if (it_dense * (uint64_t)P < 2 * D) it_dense = 0;  // too short for the ring: general phase
    uint64_t crow0 = row_first;
    if (it_dense) {
        // linear rows: the group's lanes walk their row L*16 bytes at a time.  Tiled rows (L = 4,
        // 16 rows per wave step): the wave reads one 1 KiB step of its tile per piece.
        const RowLayout lay{a.pitch, a.tiled, a.steps};
        const uint8_t *iptr = a.rows + piece_offset(lay, row_first + grp, (uint32_t)lig);
        const uint64_t piece_step = a.tiled ? 1024u : (uint64_t)L * 16;
        const uint64_t row_jump = (a.tiled ? (stride >> 4) * ((uint64_t)a.steps * 1024) : stride * (uint64_t)a.pitch) -
                                  (uint64_t)(P - 1) * piece_step;
        int ip = 0, cp = 0, jc = lig;
        acc.fetch(smem, lig, r16);
#define SZG_DN_ISSUE(u)                                                                 \
    {                                                                                   \
        ring[u] = load_piece<NT>(iptr);                                                     \
        if (++ip == P) {                                                                \
            ip = 0;                                                                     \
            iptr += row_jump;                                                           \
        } else {                                                                        \
            iptr += piece_step;                                                         \
        }                                                                               \
    }
#define SZG_DN_CONSUME(u)                                                               \
    {                                                                                   \
        const u32x4 v_ = ring[u];                                                       \
        if constexpr (RowAcc<QBITS, METRIC>::kPrefetch)                                 \
            acc.piece_pf(make_uint4(v_.x, v_.y, v_.z, v_.w), smem, cp + 1 == P ? lig : jc + L, r16); \
        else                                                                            \
            acc.piece(make_uint4(v_.x, v_.y, v_.z, v_.w), smem, jc, r16, a.dim);        \
        if (++cp == P) {                                                                \
            finish_row(crow0, dense_valid(crow0), acc);                                 \
            acc.reset();                                                                \
            cp = 0;                                                                     \
            jc = lig;                                                                   \
            crow0 += stride;                                                            \
        } else {                                                                        \
            jc += L;                                                                    \
        }                                                                               \
    }
        if constexpr (LL != 0 && PP % D == 0)
            SZG_RUN_RING_ROWS(it_dense * (uint64_t)P, SZG_DN_ISSUE, SZG_DN_CONSUME)
        else
            SZG_RUN_RING_DENSE(it_dense * (uint64_t)P, SZG_DN_ISSUE, SZG_DN_CONSUME)
#undef SZG_DN_ISSUE
#undef SZG_DN_CONSUME
    }

    // ---- general phase: masks, partial groups, ragged tails.
    // Under selective masks (MASKED && !mask_dense) most row steps of a wave hold no row that
    // passes; the wave first compacts the steps that do into its LDS step list (each lane tests
    // one step: a shift and a mask on the bitmap words), then rings over those only -- the
    // cost follows the rows that pass, not the corpus.
    {
    const RowLayout glay{a.pitch, a.tiled, a.steps};
    // (the test needs a step's gpw rows inside one bitmap word: gpw a power of two)
    const bool compact = MASKED && !a.mask_dense && (gpw & (gpw - 1)) == 0;
    uint64_t it_next = it_dense;  // next row step of this wave still to be looked at
    while (it_next < n_it) {
    uint64_t n_steps;             // steps of this round
    const uint64_t it_base = it_next;
    if (compact) {
        int nl = 0;
        while (it_next < n_it && nl <= kStepList - kWave) {
            const uint64_t it = it_next + lane;
            bool v = false;
            if (it < n_it) {
                const uint64_t r0 = row_first + it * stride;  // gpw rows, inside one bitmap word
                if (r0 < a.n_rows) {
                    const uint32_t rr = (uint32_t)r0;
                    uint64_t w = ~0ull;
                    if (a.live_bits) w &= a.live_bits[rr >> 6];
                    if (allow_bits) w &= allow_bits[rr >> 6];
                    const uint64_t cnt = min((uint64_t)gpw, (uint64_t)a.n_rows - r0);
                    const uint64_t m = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
                    v = ((w >> (rr & 63u)) & m) != 0;
                }
            }
            const uint64_t bal = __ballot(v);
            if (v) steplist[nl + __popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)(it - it_base);
            nl += __popcll(bal);
            it_next += kWave;
        }
        if (it_next > n_it) it_next = n_it;
        n_steps = (uint64_t)nl;
        __builtin_amdgcn_wave_barrier();
    } else {
        n_steps = n_it - it_next;
        it_next = n_it;
    }
    auto step_row0 = [&](uint64_t i) -> uint64_t {  // first row of the round's i-th step
        const uint64_t it = compact ? it_base + (i < n_steps ? steplist[i] : 0u) : it_base + i;
        return row_first + it * stride;
    };
    uint32_t okmask = 0;
    const uint64_t NP = n_steps * (uint64_t)P;  // pieces of this round
    // issue cursor
    uint64_t istep = 0;
    uint64_t irow0 = step_row0(0);
    int ip = 0;
    bool ivalid = n_steps ? row_valid(irow0) : false;
    bool inext = (MASKED && n_steps > 1) ? row_valid(step_row0(1)) : false;
    uint64_t irow = (uint32_t)(irow0 + grp);
    // consume cursor
    uint64_t cstep = 0;
    crow0 = irow0;
    int cp = 0;
    bool cvalid = false;

#define SZG_ISSUE(u)                                                                    \
    {                                                                                   \
        const int j_ = ip * L + lig;                                                    \
        const bool ok_ = ivalid && j_ < r16;                                            \
        ring[u] = load_piece<NT>(ok_ ? a.rows + piece_offset(glay, irow, (uint32_t)j_) : a.rows); \
        okmask = (okmask & ~(1u << (u))) | ((uint32_t)ok_ << (u));                      \
        if (++ip == P) {                                                                \
            ip = 0;                                                                     \
            istep++;                                                                    \
            irow0 = step_row0(istep);                                                   \
            irow = (uint32_t)(irow0 + grp);                                             \
            if (MASKED) {                                                               \
                ivalid = inext;                                                         \
                inext = istep + 1 < n_steps ? row_valid(step_row0(istep + 1)) : false;  \
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
            acc.piece(make_uint4(v_.x, v_.y, v_.z, v_.w), smem, j_, r16, a.dim);                 \
        }                                                                               \
        if (++cp == P) {                                                                \
            finish_row(crow0, cvalid, acc);                                             \
            acc.reset();                                                                \
            cp = 0;                                                                     \
            cstep++;                                                                    \
            crow0 = step_row0(cstep);                                                   \
        }                                                                               \
    }

    if (NP) SZG_RUN_RING(NP, SZG_ISSUE, SZG_CONSUME)
#undef SZG_ISSUE
#undef SZG_CONSUME
    }  // rounds
    }
#undef SZG_RUN_RING
#undef SZG_RUN_RING_DENSE
#undef SZG_RUN_RING_ROWS

    if (COLLECT) continue;

    // block-wide k-select over the waves' lists
    wl.flush(lane);
    __syncthreads();
    block_merge_lists(lists, nwaves, a.kp, a.block_lists + ((size_t)qi * gridDim.x + blockIdx.x) * a.kp,
                      tid, blockDim.x);
    // gfx9 counts loads and stores in one vmcnt and retires them out of order with respect
    // to each other: with the list stores above possibly pending at the top of the next
    // query's sweep, every ring wait would have to be vmcnt(0) -- which also waits for the
    // load just issued.  Drain here, once per query, so the ring gets counted waits.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt, lgkmcnt untouched
    }  // for qi
}

#if SZG_QBITS == 0
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

#endif  // SZG_QBITS == 0

#if SZG_QBITS != 0
// Ring depth.  HBM streams fastest with about 6-9 MB of reads in flight on the chip
// (scripts/readbw; deeper queues only lengthen the DRAM queues), so the ring is kept SHORT:
// 4 loads per lane for the any-shape kernels (8 when the candidate lists live in LDS, kp > 64,
// whose inserts stall longer).  Measured on one box, ring 8 -> 4: 1M x 768 f32 7.04 -> 7.12 TB/s,
// 4M x 768 8-bit 6.75 -> 6.97, 12.5M x 384 4-bit 6.31 -> 6.57.
//
// Row-shape kernels: lanes per row L, pieces per lane P and a ring depth D that DIVIDES P are
// compile-time constants, so after unrolling the ring every slot sits at a fixed position of
// the row: the row-finish test, the query offsets and the row jump are folded, no per-piece
// branch is left (12.5M x 384 4-bit with D = P = 3: 6.31 -> 6.77 TB/s).
#ifdef SZG_RING
constexpr int kRingShort = SZG_RING, kRingDeep = SZG_RING;
#define SZG_SHAPE_RING(d) SZG_RING
#else
constexpr int kRingShort = 4, kRingDeep = 8;
#define SZG_SHAPE_RING(d) d
#endif

template <int QBITS, int METRIC, bool COLLECT, int LL, int PP, int D>
hipError_t launch_shaped(const ScanArgs &a, dim3 g, dim3 b, size_t lds, hipStream_t stream)
{
    if (LL >= 8 || a.tiled)
        hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, D, COLLECT, false, true, LL, PP>), g, b, lds, stream, a);
    else
        hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, D, COLLECT, false, false, LL, PP>), g, b, lds, stream, a);
    return hipGetLastError();
}

template <int QBITS, int METRIC, bool COLLECT, bool MASKED, int D>
hipError_t launch_any_shape(const ScanArgs &a, dim3 g, dim3 b, size_t lds, hipStream_t stream)
{
    if (a.map.L >= 8 || a.tiled)  // whole lines per load instruction: stream past the caches
        hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, D, COLLECT, MASKED, true>), g, b, lds, stream, a);
    else
        hipLaunchKernelGGL((scan_kernel<QBITS, METRIC, D, COLLECT, MASKED, false>), g, b, lds, stream, a);
    return hipGetLastError();
}

template <int QBITS, int METRIC, bool COLLECT, bool MASKED>
hipError_t launch_scan_qmcm(const ScanArgs &a, int grid, int block, size_t lds, hipStream_t stream)
{
    const dim3 g(grid), b(block);
    const bool deep = (!COLLECT && a.kp > 64) || a.ring >= 8;
    if constexpr (!MASKED) {
        if (a.map.dense && a.map.L * a.map.gpw == kWave && !a.no_shape_kernels && !deep) {
            const int L = a.map.L, P = a.map.P;
#define SZG_TRY_SHAPE(l, p, d)                                                                 \
    if (L == l && P == p) return launch_shaped<QBITS, METRIC, COLLECT, l, p, SZG_SHAPE_RING(d)>(a, g, b, lds, stream);
            if constexpr (QBITS == 4) {   // tiled rows walk 4 lanes per row: 384 / 768 dims
                SZG_TRY_SHAPE(4, 3, 3)
                SZG_TRY_SHAPE(4, 6, 6)
            } else if constexpr (QBITS == 8) {
                SZG_TRY_SHAPE(4, 6, 6)
                SZG_TRY_SHAPE(4, 12, 4)
            } else if constexpr (QBITS == 32) {   // linear rows, 8 lanes per row; (8, 24) unrolled over a whole
                SZG_TRY_SHAPE(8, 12, 4)           // row spills registers and measured slower than any-shape
            }
#undef SZG_TRY_SHAPE
        }
    }
    if (deep) return launch_any_shape<QBITS, METRIC, COLLECT, MASKED, kRingDeep>(a, g, b, lds, stream);
    return launch_any_shape<QBITS, METRIC, COLLECT, MASKED, kRingShort>(a, g, b, lds, stream);
}

template <int QBITS, int METRIC>
hipError_t launch_scan_qm(const ScanArgs &a, int grid, int block, size_t lds, hipStream_t stream)
{
    const bool masked = a.live_bits != nullptr || a.allow_bits != nullptr;
    if (a.collect) {
        if (masked) return launch_scan_qmcm<QBITS, METRIC, true, true>(a, grid, block, lds, stream);
        return launch_scan_qmcm<QBITS, METRIC, true, false>(a, grid, block, lds, stream);
    }
    if (masked) return launch_scan_qmcm<QBITS, METRIC, false, true>(a, grid, block, lds, stream);
    return launch_scan_qmcm<QBITS, METRIC, false, false>(a, grid, block, lds, stream);
}

template <int QBITS>
hipError_t launch_scan_q(int metric, const ScanArgs &a, int grid, int block, size_t lds,
                         hipStream_t stream)
{
    if (metric == kCosine) return launch_scan_qm<QBITS, kCosine>(a, grid, block, lds, stream);
    return launch_scan_qm<QBITS, kEuclidean>(a, grid, block, lds, stream);
}

#endif  // SZG_QBITS != 0

}  // namespace

#if SZG_QBITS != 0
// this translation unit carries the scan kernels of ONE quantization
#define SZG_CAT2(a, b) a##b
#define SZG_CAT(a, b) SZG_CAT2(a, b)
// ... and of ONE metric (-DSZG_SCAN_METRIC=0 Euclidean / 1 cosine): ten objects of ~35 s instead of five of ~70 s,
// which is what a parallel build of the library waits for
#ifndef SZG_SCAN_METRIC
#error "kernels_scan.hip with SZG_QBITS needs SZG_SCAN_METRIC (0 or 1)"
#endif
hipError_t SZG_CAT(SZG_CAT(launch_scan_q, SZG_QBITS), SZG_CAT(m, SZG_SCAN_METRIC))(const ScanArgs &a, int grid, int block,
                                                                                 size_t lds, hipStream_t stream)
{
    return launch_scan_qm<SZG_QBITS, SZG_SCAN_METRIC>(a, grid, block, lds, stream);
}
#else
#define SZG_DECL_SCAN(q)                                                          \
    hipError_t launch_scan_q##q##m0(const ScanArgs &, int, int, size_t, hipStream_t); \
    hipError_t launch_scan_q##q##m1(const ScanArgs &, int, int, size_t, hipStream_t);
SZG_DECL_SCAN(4)
SZG_DECL_SCAN(8)
SZG_DECL_SCAN(16)
SZG_DECL_SCAN(32)
SZG_DECL_SCAN(64)
#undef SZG_DECL_SCAN

size_t scan_lds_bytes(int qbits, const RowMap &m, int kp, int block)
{
    // query image + per-wave candidate lists + per-wave lists of row steps that survive the masks
    return query_lds_bytes(qbits, m.r16) + (size_t)(block / kWave) * kp * sizeof(uint64_t) +
           (size_t)(block / kWave) * kStepList * sizeof(uint32_t);
}

hipError_t launch_scan(int qbits, int metric, const ScanArgs &a, int grid, int block,
                       hipStream_t stream)
{
    const size_t lds = scan_lds_bytes(qbits, a.map, a.collect ? 0 : a.kp, block);
    static_assert(kEuclidean == 0 && kCosine == 1, "the scan objects are named by the metric's number");
#define SZG_CASE_SCAN(q)                                                                                            \
    case q: return metric == kCosine ? launch_scan_q##q##m1(a, grid, block, lds, stream) : launch_scan_q##q##m0(a, grid, block, lds, stream);
    switch (qbits) {
    SZG_CASE_SCAN(4)
    SZG_CASE_SCAN(8)
    SZG_CASE_SCAN(16)
    SZG_CASE_SCAN(32)
    SZG_CASE_SCAN(64)
    default: return hipErrorInvalidValue;
    }
#undef SZG_CASE_SCAN
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

#endif  // SZG_QBITS

}  // namespace szg
